#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for k in 1000 500 250 160; do
  timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 --debug-k $k --host-threads 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($k, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['other_scorer_ms_per_step'], d['roofline']['host_plan_ms_per_step'], d['config']['dist_stage_ms'])"
done
timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['p50_latency_ms'], d['max_latency_ms'], d['slowest_step'])"
