#!/bin/bash
# Round 3, call T: differential fuzz of the exact vector search (bit for bit against the oracle).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
NRT_KNN_FUZZ_ROUNDS=${NRT_KNN_FUZZ_ROUNDS:-24} timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k vector > $O/pytest_t.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_t.log | tail -40 | cut -c1-600
echo "== done =="
