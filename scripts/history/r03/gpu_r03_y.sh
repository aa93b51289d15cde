#!/bin/bash
# Round 3, call Y: the query panel converted to fp16 once per panel (not per launch and workgroup): tests, C4 at 10 M rows and a 1/8 share.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_vectors_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_y.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_y.log | tail -8 | cut -c1-400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d['p50_latency_ms'], r['avg_launch_ms'], r['frac'], d.get('closed_loop'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for Q in 32 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 40 --warmup 4 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/bench_y_q${Q}.json | show q${Q}
done
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 80 --warmup 4 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_y_emu8.json | show emu8
echo "== done =="
