#!/bin/bash
# Round 3, call K: the cross-GPU bound exchange on the MaxScore route: protocol tests; projection of rank 0's shard with the other
# ranks' publications played by the script (world 2 / 4 / 8).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 300 python -m pytest tests/test_exchange_gpu.py tests/test_maxscore_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_k.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_k.log | tail -8 | cut -c1-300
for W in 8 4 2; do
  timeout 200 python scripts/gpu_exchange_projection.py --world $W 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/exchange_projection_k.log
done
echo "== done =="
