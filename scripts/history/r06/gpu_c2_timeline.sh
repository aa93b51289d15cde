#!/bin/bash
# round 6: the device timeline of C2 (1 M docs, 2-term, top-100, 1024 queries per step, four submitting threads)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06c2t}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl2; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/tl2 -o t --output-format csv -- python $ROOT/bench.py --workload C2 --steps 60 --warmup 5 --host-threads 4 --no-cpu-baseline --closed-loop '' > $O/${TAG}_bench.log 2>&1 )
f=$(find /tmp/tl2 -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee $O/${TAG}_timeline.txt
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'bm25_maxscore' in r['Kernel_Name'] or 'bm25_scan' in r['Kernel_Name']]
print('scorer launches', len(ms))
durs = collections.defaultdict(list)
for r in rows[ms[10]:ms[-5]]:
    durs[r['Kernel_Name'][:44]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])): print(k.ljust(46), len(v), 'avg us', round(sum(v) / len(v), 1), 'total ms', round(sum(v) / 1e3, 2))
span = (int(rows[ms[-5]]['Start_Timestamp']) - int(rows[ms[10]]['Start_Timestamp'])) / 1e3
print('span us', round(span, 1), 'per scorer launch', round(span / (len(ms) - 15), 1))
s = ms[-12]
t0 = int(rows[s]['Start_Timestamp']); prev = None
for r in rows[s - 2: s + 22]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(r['Kernel_Name'][:50].ljust(50), 'start', round((st - t0) / 1e3, 1), 'us  dur', round((en - st) / 1e3, 1), ' gap', None if prev is None else round((st - prev) / 1e3, 1))
    prev = en
PY
