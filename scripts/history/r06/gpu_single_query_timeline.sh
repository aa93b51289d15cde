#!/bin/bash
# round 6: the device timeline of ONE query per call (closed loop, one caller, nrtgpu_search_bm25_coalesced): what a 0.25 ms call runs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06sq}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
rm -rf /tmp/tl1; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl1 -o t --output-format csv -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --closed-loop 1 --closed-loop-ms 300 --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 > $O/${TAG}_bench.log 2>&1 )
tail -1 $O/${TAG}_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('closed loop', d.get('closed_loop'))"
f=$(find /tmp/tl1 -name "*kernel_trace.csv" | head -1); g=$(find /tmp/tl1 -name "*memory_copy_trace.csv" | head -1)
python - "$f" "$g" <<'PY' | tee $O/${TAG}_timeline.txt
import csv, sys
rows = [dict(r, kind='kernel', name=r['Kernel_Name']) for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows += [dict(r, kind='copy', name='COPY ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))) for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e:
    print('no copy trace', e)
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 600 events: single-query calls
tail = rows[-600:]
ms = [i for i, r in enumerate(tail) if 'bm25_maxscore' in r['name'] or 'bm25_scan' in r['name']]
print('events', len(rows), 'scorer launches in the tail', len(ms))
for s in ms[-4:-2]:
    t0 = int(tail[s]['Start_Timestamp']); prev = None
    for r in tail[max(0, s - 6): s + 8]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(r['name'][:56].ljust(56), 'start', round((st - t0) / 1e3, 1), 'us  dur', round((en - st) / 1e3, 1), ' gap', None if prev is None else round((st - prev) / 1e3, 1))
        prev = en
    print()
PY
