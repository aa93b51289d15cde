#!/bin/bash
# round 5, call z (the last seconds): bench.py --gpus 2 on one GPU through the library's collective, once, on the tree as committed
cd "${GRAFT_REPO_ROOT:-/root/repo}"
/opt/rocm/bin/hipcc -O1 -fPIC -shared -x hip --offload-arch=gfx950 tests/mockrccl/mockrccl.cpp -o /tmp/librccl_mock.so
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_RCCL_LIB=/tmp/librccl_mock.so NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE=1 NRTGPU_BENCH_COLLECTIVE_TIMEOUT=15 MASTER_ADDR=127.0.0.1 NRTGPU_BENCH_WATCHDOG=15
timeout 18 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --debug-same-gpu --workload C2 --steps 12 --warmup 3 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --exchange-mode alltoall 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n_gpus', d['n_gpus'], 'value', d['value'], 'shard speculation', d['config']['shard_speculation'])"
