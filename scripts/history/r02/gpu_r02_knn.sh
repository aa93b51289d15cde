#!/bin/bash
# kNN with long rounds (+ overflow fallback tests), the C4 bench line, cold-plan check of the C3 line, library collective
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_baseline_sizes_gpu.py -q -x -p no:cacheprovider -k "knn or vector or exchange or dist" > gpurun_out/r02/pytest_knn.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error|Error" gpurun_out/r02/pytest_knn.log | tail -8
for q in 32 1 64; do
  timeout 600 python bench.py --workload C4 --knn-queries $q --steps 5 --warmup 2 $([ $q != 32 ] && echo --no-cpu-baseline) 2>gpurun_out/r02/bench_c4_q$q.err | tee gpurun_out/r02/bench_c4_q$q.json | cut -c1-1500
done
timeout 300 python bench.py --no-cpu-baseline | tee gpurun_out/r02/bench_c3_b.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['host_plan_ms_per_step'])"
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 2>gpurun_out/r02/bench_emu8_lib.err | tee gpurun_out/r02/bench_emu8_lib.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['host_plan_ms_per_step'], d['config']['sharding'], d['config']['dist_stage_ms'])"
tail -3 gpurun_out/r02/bench_emu8_lib.err
