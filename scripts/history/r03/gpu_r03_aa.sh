#!/bin/bash
# Round 3, call AA: exact kNN over an index of MANY segments (10 M rows in 40 / 160 leaves): what the per-leaf launches cost.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['score_launches_per_panel'], d['config']['segments_per_gpu'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
for SR in 2500000 250000 62500; do
  timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 30 --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" --c4-seg-rows $SR 2>/dev/null | tee $O/bench_aa_$SR.json | show seg_rows_$SR
done
echo "== done =="
