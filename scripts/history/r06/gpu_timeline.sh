#!/bin/bash
# round 6: the device's timeline of the driver-form command -- what sits between two bm25_maxscore_kernel launches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06t}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python $ROOT/bench.py --steps 40 --warmup 5 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 > $O/${TAG}_bench.log 2>&1 )
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/${TAG}_timeline.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ms = [r for r in rows if 'bm25_maxscore' in r['Kernel_Name']]
print('launches', len(ms))
gaps = []
for a, b in zip(ms[10:-1], ms[11:]):
    ea, sb = int(a['End_Timestamp']), int(b['Start_Timestamp'])
    between = [r for r in rows if int(r['Start_Timestamp']) >= int(a['Start_Timestamp']) and int(r['Start_Timestamp']) < sb and r is not a]
    gaps.append((sb - ea, int(b['End_Timestamp']) - sb, [(r['Kernel_Name'][:40], int(r['Start_Timestamp']) - ea, int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in between]))
g = sorted(x[0] for x in gaps)
print('gap between consecutive maxscore launches (end -> next start), ns: median', g[len(g)//2], 'mean', sum(g)//len(g), 'p90', g[int(len(g)*.9)], 'min', g[0], 'max', g[-1])
d = sorted(x[1] for x in gaps)
print('maxscore duration ns: median', d[len(d)//2], 'mean', sum(d)//len(d))
for x in gaps[5:9]:
    print('gap', x[0], 'ns; kernels started between the two (name, start - prev end, duration):')
    for k in x[2]: print('    ', k)
PY
