#!/bin/bash
# Round 4, call 11: where the walk's time goes (scripts/gpu_phase_clocks.py on the -DNRT_MS_PHASE_CLOCKS build).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/l; mkdir -p $O
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_phase.so timeout 300 python scripts/gpu_phase_clocks.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tee $O/phase_clocks.log
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_rounds.so timeout 300 python scripts/gpu_phase_clocks.py --rounds 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tee $O/round_counts.log
