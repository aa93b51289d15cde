#!/bin/bash
# Round 4, call 14: the BM25 files under the new record policy and under the old threshold (call 13 lost their verdict lines).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/o; mkdir -p $O
BM25="tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fullsize_gpu.py"
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -${1:-3}; }
timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tee $O/bm25_default.log | flt
NRTGPU_RECORD_DOCS_PER_POSTING=128 timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tee $O/bm25_128.log | flt
NRTGPU_PACKED_POSTINGS=1 timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tee $O/bm25_packed.log | flt
timeout 300 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | flt 4
