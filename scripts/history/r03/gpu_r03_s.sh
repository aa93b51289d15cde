#!/bin/bash
# Round 3, call S: the one slow step of the 1-query C4 line: gone with the interpreter's cycle collector off during the timed region?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['value'], d['ms_per_step'], d['p50_latency_ms'], d['max_latency_ms'], d['latency_outliers'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
timeout 200 python bench.py --workload C4 --knn-queries 1 --steps 120 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_s_q1.json | show q1_sketch
timeout 200 python bench.py --workload C4 --knn-queries 2 --steps 120 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_s_q2.json | show q2_sketch
echo "== done =="
