#!/bin/bash
# Round 3, call O: nominations from the fp16 sketch: vector / exchange / hybrid tests, C4 at 10M rows with and without the sketch.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_abi.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_o.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_o.log | tail -25 | cut -c1-400
timeout 300 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k knn > $O/pytest_o2.log 2>&1; echo "pytest sizes rc=$?"; tail -5 $O/pytest_o2.log | cut -c1-400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'q', d['config']['queries_per_step'], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('effective_frac'), r.get('second_passes'), d.get('verify',{}).get('agrees_with_fp64'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for Q in 32 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 10 --warmup 2 --no-cpu-baseline 2>$O/bench_o_c4_q$Q.err | tee $O/bench_o_c4_q$Q.json | show sketch
done
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline --no-sketch 2>$O/bench_o_c4_nosketch.err | tee $O/bench_o_c4_q32_nosketch.json | show fp32
tail -3 $O/bench_o_c4_q32.err | cut -c1-300
echo "== done =="
