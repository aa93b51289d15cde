#!/bin/bash
# round 5, call d: shard-level speculation -- the two-rank test through the library's collective, the shards played one after
# another, one rank of eight
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05d
timeout 800 python -m pytest tests/test_dist_two_ranks_gpu.py tests/test_shard_speculation_gpu.py tests/test_exchange_gpu.py -m gpu -q -x --tb=short -s -p no:cacheprovider > gpurun_out/r05d/pytest.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" gpurun_out/r05d/pytest.log | grep "shard\|passed\|failed\|Error\|assert" | tail -30 | cut -c1-300
bash scripts/gpu_measure.sh r05d emulate8 trace8
