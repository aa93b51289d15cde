#!/bin/bash
# Round 3, call AL: the second pass runs from the sketch (the fp32 rows only when a list overflows): the tests that force second passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 300 python -m pytest tests/test_vectors_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "equal_rows or near_duplicates or sketch" > $O/pytest_al.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_al.log | tail -12 | cut -c1-400
echo "== done =="
