#!/bin/bash
# Round 3, call AD: the BM25 GPU suite on the compressed-postings layout (NRTGPU_PACKED_POSTINGS=1) and a longer query-shape fuzz, closing build.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
NRTGPU_PACKED_POSTINGS=1 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py tests/test_packed_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_ad_packed.log 2>&1; echo "packed suite rc=$?"; tail -4 $O/pytest_ad_packed.log | cut -c1-300
NRT_FUZZ_ROUNDS=40 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k shapes > $O/pytest_ad_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/pytest_ad_fuzz.log | cut -c1-300
echo "== done =="
