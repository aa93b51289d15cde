#!/bin/bash
# Round 4, call 12: membership + rank records for sparser terms (NRTGPU_RECORD_DOCS_PER_POSTING: a term gets records when it has a
# posting per that many docs; 128 = rounds 2-4): the binary search of the sparse-clause lookups is 18 % of the walk (call 11).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/m; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'hbm_gb', d.get('hbm_used_gb'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for K in 128 1024 4096 32768 128; do
  NRTGPU_RECORD_DOCS_PER_POSTING=$K timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_K$K.json | show "c3 K=$K"
done
NRTGPU_RECORD_DOCS_PER_POSTING=32768 timeout 300 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_filters_gpu.py -m gpu -q --maxfail=4 --tb=short -p no:cacheprovider -k "not knn" 2>&1 | tail -3
