#!/bin/bash
# Round 3, call AM: the whole GPU suite + smoke on the closing build (what the driver runs at round end).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_am.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_am.log | tail -4 | cut -c1-300
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== done =="
