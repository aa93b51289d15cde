#!/bin/bash
# round 5, call y: the two-rank bench (tests/test_bench_two_ranks_gpu.py's command) in a loop, to catch what made it fail once
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05y; mkdir -p $O
/opt/rocm/bin/hipcc -O1 -fPIC -shared -x hip --offload-arch=gfx950 tests/mockrccl/mockrccl.cpp -o /tmp/librccl_mock.so
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_RCCL_LIB=/tmp/librccl_mock.so NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE=1 NRTGPU_BENCH_COLLECTIVE_TIMEOUT=20 MASTER_ADDR=127.0.0.1 NRTGPU_BENCH_WATCHDOG=25
T0=$(date +%s)
for i in 1 2 3 4 5 6 7 8; do
  [ $(( $(date +%s) - T0 )) -gt 55 ] && break
  timeout 30 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 2 --debug-same-gpu --workload C2 --steps 12 --warmup 3 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --exchange-mode allgather > $O/run_$i.out 2> $O/run_$i.err
  rc=$?
  echo "run $i rc=$rc $(( $(date +%s) - T0 )) s: $(grep -c '^{' $O/run_$i.out) line(s)"
  if [ $rc -ne 0 ]; then grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/run_$i.err | grep -n "failed at a step\|Error\|error\|mockrccl\|Traceback" | head -20; fi
done
