#!/bin/bash
# Round 3, call AB: the sketch kernel walks ALL leaves of a search in one launch (leaf table): tests, many-segment C4 again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
NRT_KNN_FUZZ_ROUNDS=80 timeout 900 python -m pytest tests/test_vectors_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_ab.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_ab.log | tail -12 | cut -c1-500
timeout 300 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k knn > $O/pytest_ab2.log 2>&1; echo "sizes rc=$?"; tail -3 $O/pytest_ab2.log | cut -c1-300
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['score_launches_per_panel'], d['config']['segments_per_gpu'], d.get('verify',{}).get('agrees_with_fp64'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for SR in 2500000 250000 62500; do
  timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 30 --warmup 3 --no-cpu-baseline --closed-loop "" --c4-seg-rows $SR 2>/dev/null | tee $O/bench_ab_$SR.json | show seg_rows_$SR
done
timeout 300 python bench.py --workload C4 --knn-queries 64 --steps 30 --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/bench_ab_q64.json | show q64
timeout 300 python bench.py --workload C4 --knn-queries 1 --steps 60 --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/bench_ab_q1.json | show q1
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 80 --warmup 4 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_ab_emu8.json | show emu8
echo "== done =="
