#!/bin/bash
# full GPU suite (summary captured to a file), planner trace, the three bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r02/pytest_gpu.log | tail -5
NRTGPU_PLAN_TRACE=1 timeout 300 python bench.py --steps 6 --warmup 2 2>gpurun_out/r02/plan_trace.log | tee gpurun_out/r02/bench_c3.json | cut -c1-400
grep -E "plan " gpurun_out/r02/plan_trace.log | tail -4
timeout 300 python bench.py --workload C2 --steps 20 --warmup 3 --no-cpu-baseline | tee gpurun_out/r02/bench_c2.json | cut -c1-300
timeout 300 python bench.py --emulate-world 8 --steps 10 --warmup 3 --no-cpu-baseline | tee gpurun_out/r02/bench_emu8.json | cut -c1-500
