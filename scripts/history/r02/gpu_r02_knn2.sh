#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_vectors_gpu.py tests/test_baseline_sizes_gpu.py tests/test_hybrid_gpu.py -q -x -p no:cacheprovider -k "knn or vector or hybrid" > gpurun_out/r02/pytest_knn.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error|Error" gpurun_out/r02/pytest_knn.log | tail -8
for q in 64 48 32; do
  timeout 600 python bench.py --workload C4 --knn-queries $q --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/r02/bench_c4_q$q.err | tee gpurun_out/r02/bench_c4_q$q.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['queries_per_step'], d['value'], d['ms_per_step'], d['p50_latency_ms'], d['roofline'])"
done
