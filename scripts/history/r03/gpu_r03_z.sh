#!/bin/bash
# Round 3, call Z: segments uploaded and sealed while searches run (NRT churn).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_filters_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_z.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_z.log | tail -20 | cut -c1-400
echo "== done =="
