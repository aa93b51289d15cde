#!/bin/bash
# Round 3, call N: the rescoring of the nominations with 256 bytes per lane in flight: vector tests, C4 q32 / q1 / q64, kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_vectors_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_n.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_n.log | tail -8 | cut -c1-400
for Q in 32 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $Q --steps 10 --warmup 2 --no-cpu-baseline 2>$O/bench_n_c4.err | tee $O/bench_n_c4_q$Q.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q', d['config']['queries_per_step'], d['value'], d['ms_per_step'], r['frac'], r.get('mfma_frac'), d['config'].get('verify'))"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_n -o c4 -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline --no-verify > $O/prof_n.log 2>&1
cd $ROOT
find $O/prof_n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c4_kernel_stats_n.csv
head -8 $O/c4_kernel_stats_n.csv | cut -c1-200
echo "== done =="
