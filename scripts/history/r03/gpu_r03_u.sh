#!/bin/bash
# Round 3, call U: rows of a pass's first round (NRTGPU_KNN_FIRST_ROUND: 65536 = the default, 16384, 32768): C4 step at 10 M rows and a 1/8 share.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d['p50_latency_ms'], r['avg_launch_ms'], r['score_launches_per_panel'], r['second_passes'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
for FR in 65536 16384 32768; do
  export NRTGPU_KNN_FIRST_ROUND=$FR
  timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 40 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | show full_$FR
  timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 60 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | show emu8_$FR
done
echo "== done =="
