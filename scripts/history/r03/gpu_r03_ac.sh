#!/bin/bash
# Round 3, call AC: full-size parity soak of the closing build: 4096 C3 queries x top-1000 against the oracle's exhaustive scorer
# (plain, 1 % deletes folded, 1 % deletes through the mask), and the same through the compressed-postings layout.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python scripts/gpu_parity_c3.py --queries 4096 2>&1 | grep -v amdgpu.ids | tee $O/parity_c3_full_size.log | cut -c1-200
NRTGPU_PACKED_POSTINGS=1 timeout 600 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v amdgpu.ids | tee $O/parity_c3_full_size_packed.log | cut -c1-200
echo "== done =="
