#!/bin/bash
# round 6: the device timeline of the exhaustive route (bench.py --no-prune): what surrounds bm25_scan_kernel
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06np}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl6; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/tl6 -o t --output-format csv -- python $ROOT/bench.py --no-prune --steps 8 --warmup 2 --no-cpu-baseline --closed-loop '' --c4-steps 0 --c2-steps 0 --c5-steps 0 > $O/${TAG}_bench.log 2>&1 )
f=$(find /tmp/tl6 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/${TAG}_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'bm25_scan' in r['Kernel_Name']]
print('scan launches', len(ms))
s = ms[-4]
t0 = int(rows[s]['Start_Timestamp']); prev = None
for r in rows[s - 4: s + 14]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(r['Kernel_Name'][:50].ljust(50), 'start', round((st - t0) / 1e3, 1), 'us  dur', round((en - st) / 1e3, 1), ' gap', None if prev is None else round((st - prev) / 1e3, 1), ' wg', r.get('Grid_Size_X', r.get('Grid_Size')))
    prev = en
PY
