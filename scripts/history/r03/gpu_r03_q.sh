#!/bin/bash
# Round 3, call Q: sketch kernel with the hand-driven ring (never drained), scalar norm loads, staging outside the device lock,
# two callers in the C4 bench: tests, C4 at 1 / 32 / 64 queries, one caller for comparison, kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_q.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_q.log | tail -15 | cut -c1-400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], 'q', d['config']['queries_per_step'], d['value'], d['ms_per_step'], d['p50_latency_ms'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('effective_frac'), r.get('second_passes'), d.get('verify',{}).get('agrees_with_fp64'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for Q in 32 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 20 --warmup 2 --no-cpu-baseline 2>$O/bench_q_c4_q$Q.err | tee $O/bench_q_c4_q$Q.json | show sketch
done
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 20 --warmup 2 --no-cpu-baseline --host-threads 1 2>$O/bench_q_c4_1thr.err | tee $O/bench_q_c4_q32_1thr.json | show one-caller
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 20 --warmup 2 --no-cpu-baseline --no-sketch 2>$O/bench_q_c4_nosketch.err | tee $O/bench_q_c4_q32_nosketch.json | show fp32
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_q -o c4 --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $O/prof_q.log 2>&1
cd $ROOT
find $O/prof_q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats_q.csv
echo "== done =="
