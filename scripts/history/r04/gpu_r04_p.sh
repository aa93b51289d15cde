#!/bin/bash
# Round 4, call 15: the second-accumulator shapes (DisjunctionMaxQuery tie breaker > 0, MUST next to SHOULD clauses: the MaxScore
# kernel's SHAPES == 2 instantiation) against the oracle; the fuzz with them mixed in; the default line (no regression).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/p; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -${1:-3}; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
timeout 300 python -m pytest tests/test_filters_gpu.py -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "tie_breaker or must_next or disjunction_max or must_conjunction" 2>&1 | tee $O/new_shapes.log | flt 40
NRT_FUZZ_ROUNDS=48 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k shapes 2>&1 | tee $O/fuzz.log | flt 30
NRTGPU_PACKED_POSTINGS=1 NRT_FUZZ_ROUNDS=16 timeout 600 python -m pytest tests/test_fuzz_gpu.py tests/test_filters_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "shapes or tie_breaker or must_next" 2>&1 | tee $O/fuzz_packed.log | flt 30

