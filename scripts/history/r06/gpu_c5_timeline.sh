#!/bin/bash
# round 6: the device timeline of the C5 leg (5 M docs, recall-1000 + 768-d rescore, fused): what one 256-query batch runs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06h}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl5; ( cd /tmp && timeout 500 rocprofv3 --kernel-trace -d /tmp/tl5 -o t --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 8 > $O/${TAG}_bench.log 2>&1 )
f=$(find /tmp/tl5 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/${TAG}_c5_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'hybrid_rescore' in r['Kernel_Name']]
print('hybrid batches', len(idx))
for s in idx[-3:-1]:
    lo = s
    while lo > 0 and int(rows[s]['Start_Timestamp']) - int(rows[lo - 1]['Start_Timestamp']) < 2_000_000: lo -= 1
    t0 = int(rows[lo]['Start_Timestamp']); prev = None
    for r in rows[lo:s + 3]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(r['Kernel_Name'][:50].ljust(50), 'start', round((st - t0) / 1e3, 1), 'us  dur', round((en - st) / 1e3, 1), ' gap', None if prev is None else round((st - prev) / 1e3, 1))
        prev = en
    print()
PY
