#!/bin/bash
# Round 4, call 17: speculative thresholds as built (a context starts with margin 5): the new test, the BM25 files, the line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/r; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -${1:-3}; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'spec', d.get('config',{}).get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
timeout 300 python -m pytest tests/test_maxscore_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k speculative 2>&1 | tee $O/spec_test.log | flt 30
BM25="tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fullsize_gpu.py"
timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tee $O/bm25.log | flt 12
for Z in 5 0 5; do
  NRTGPU_MS_SPEC_Z=$Z timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_z$Z.json | show "c3 z=$Z"
done
NRTGPU_MS_SPEC_Z=5 timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --packed 2>/dev/null | tee $O/c3_packed.json | show "c3 packed z=5"
timeout 250 python bench.py --workload C2 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c2.json | show "c2"
