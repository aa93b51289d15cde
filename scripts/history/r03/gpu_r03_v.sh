#!/bin/bash
# Round 3, call V: vector-search calls take turns on the device (the lock covers the enqueue only): tests, C4 with one / two callers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_v.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_v.log | tail -12 | cut -c1-400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d['p50_latency_ms'], r['avg_launch_ms'], d['config'].get('host_threads'), d['latency_outliers'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
for Q in 32 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 40 --warmup 4 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_v_q${Q}_1.json | show q${Q}_one
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 40 --warmup 4 --no-cpu-baseline --no-verify --c4-callers 2>/dev/null | tee $O/bench_v_q${Q}_2.json | show q${Q}_two
done
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 80 --warmup 4 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_v_emu8_1.json | show emu8_one
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 80 --warmup 4 --no-cpu-baseline --no-verify --c4-callers 2>/dev/null | tee $O/bench_v_emu8_2.json | show emu8_two
echo "== done =="
