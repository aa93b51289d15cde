#!/bin/bash
# Round 4, call 26: the fused hybrid under speculative thresholds (host-side change: same code objects): its tests, C5 at full size.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/y; mkdir -p $O
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -${1:-3}; }
timeout 300 python -m pytest tests/test_hybrid_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider 2>&1 | tee $O/hybrid_tests.log | flt 25
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 python scripts/gpu_c5_hybrid.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tee $O/c5_hybrid_50M.log | cut -c1-500
