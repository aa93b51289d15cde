#!/bin/bash
# round 6: the device timeline of one C4 pass (64 queries): what each launch of the nomination kernel takes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06c}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl4; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/tl4 -o t --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries ${2:-64} --steps 12 --warmup 3 --no-cpu-baseline --no-verify > $O/${TAG}_bench.log 2>&1 )
f=$(find /tmp/tl4 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/${TAG}_c4_timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'knn_panel_fp16' in r['Kernel_Name']]
print('passes', len(idx))
for s in idx[-4:-2]:
    e = idx[idx.index(s) + 1]
    t0 = int(rows[s]['Start_Timestamp']); prev = None
    for r in rows[s - 3:e]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(r['Kernel_Name'][:46].ljust(46), 'start', round((st - t0) / 1e3, 1), 'us  dur', round((en - st) / 1e3, 1), ' gap', None if prev is None else round((st - prev) / 1e3, 1), ' grid', r.get('Grid_Size'), r.get('Workgroup_Size'))
        prev = en
    print()
PY
