#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_sizes_gpu.py -q -x -p no:cacheprovider > gpurun_out/r02/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error|Error" gpurun_out/r02/pytest_gpu.log | tail -4
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('other_scorer_ms_per_step'), r.get('host_plan_ms_per_step'))" "$1"; }
timeout 300 python bench.py --no-cpu-baseline | show c3
timeout 300 python bench.py --no-cpu-baseline --host-threads 3 | show c3_t3
timeout 300 python bench.py --no-cpu-baseline --workload C2 | show c2
timeout 300 python bench.py --no-cpu-baseline --workload C2 --host-threads 3 | show c2_t3
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | show emu8
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 --host-threads 3 2>/dev/null | show emu8_t3
timeout 300 python bench.py --no-cpu-baseline --no-prune | show c3_noprune
timeout 200 python scripts/gpu_closed_loop.py 2>&1 | grep clients | tail -3
