#!/bin/bash
# Round 5, call c: round 4's library against this one under the same lookup structure (records for 2048 terms), the records /
# lookup-cells budget curve, then the whole GPU suite on the new boundary (hooks in the development library).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/c; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
el "round 4's library (records for the 2048 largest terms, no budget)"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_r04dev.so timeout 300 python scripts/gpu_look_policy.py --configs "r04|0;r04|0" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/look_policy_r04lib.log | cut -c1-330
el "this library: records / cells budget curve"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so timeout 600 python scripts/gpu_look_policy.py --configs "bits:2048|100000;bits:2048,cells|100000;bits@384,cells|150;bits@256,cells|150;bits@128,cells|150;bits@512,cells|200;bits@1024,cells|300;bits:2048|100000" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/look_policy_curve.log | cut -c1-330
el "GPU suite"
timeout 600 python -m pytest tests -m gpu -q --maxfail=8 --tb=short --durations=6 -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -16 | cut -c1-220
el "done"
