#!/bin/bash
# round 7: LDS rtn-atomic microbench, parity after the planner changes, bench pipeline (multi-GPU path on 1 GPU)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== ubench =="
timeout 120 ./scripts/ubench/lds_atomics 2>&1 | tee gpurun_out/ubench_lds.log
echo "== pytest gpu =="
timeout 500 python -m pytest tests -m gpu -q --timeout 120 2>&1 | tail -5
echo "== bench default =="
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/b7_default.json | cut -c1-1500
echo "== bench force-dist (world 1, full index) =="
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --force-dist 2>&1 | tail -1 | tee gpurun_out/b7_dist1.json | cut -c1-1500
echo "== bench force-dist emulate-world 8 =="
timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --force-dist --emulate-world 8 2>&1 | tail -1 | tee gpurun_out/b7_dist8.json | cut -c1-1500
echo "== torchrun world=1 nccl =="
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --force-dist --emulate-world 2 2>&1 | tail -2 | cut -c1-1500
echo "== done =="
