#!/bin/bash
# The GPU form of bench.py's N = 2 loop (tests/test_bench_two_ranks_gpu.py: torch.distributed.run, both ranks on the one GPU, the
# library's collective carried by tests/mockrccl) repeated until it fails: VERDICT round 5 item 2 asks for the cause of one
# failure in ~30 runs of round 5, whose exception text was lost.  bench.py now prints what failed at the step (step_failed) and
# leaves at once; a failing repetition's stderr is kept whole.
#     gpurun --timeout 1500 -- 'bash scripts/gpu_two_rank_loop.sh <tag> <repetitions> [workload]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:?tag}; REPS=${2:-20}; WL=${3:-C2}
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
MOCK=/tmp/librccl_mock.so
/opt/rocm/bin/hipcc -O1 -fPIC -shared -x hip --offload-arch=gfx950 tests/mockrccl/mockrccl.cpp -o $MOCK || exit 1
python -c "from nrtsearch_amd import build; build.build_dev()" || exit 1
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_RCCL_LIB=$MOCK NRTGPU_BENCH_DEBUG_LIB_COLLECTIVE=1 NRTGPU_BENCH_COLLECTIVE_TIMEOUT=60
export MASTER_ADDR=127.0.0.1 NRTGPU_BENCH_WATCHDOG=100
bad=0; T0=$(date +%s)
for rep in $(seq 1 $REPS); do
  for mode in allgather alltoall; do
    port=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
    timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --debug-same-gpu \
      --workload $WL --steps 12 --warmup 3 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --exchange-mode $mode \
      > /tmp/two_rank.out 2> /tmp/two_rank.err
    rc=$?
    line=$(grep -c '^{' /tmp/two_rank.out)
    if [ $rc -ne 0 ] || [ "$line" != "1" ]; then
      bad=$((bad + 1))
      echo "repetition $rep $mode: rc=$rc lines=$line ($(( $(date +%s) - T0 )) s)"
      grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" /tmp/two_rank.err > $O/two_rank_failure_${rep}_${mode}.log
      grep -n "exchange stage failed\|Error\|error\|mockrccl" $O/two_rank_failure_${rep}_${mode}.log | head -12 | cut -c1-300
      rm -rf /dev/shm/nrtgpu_mockrccl_* 2>/dev/null
    fi
  done
done
echo "two-rank loop: $REPS repetitions x 2 forms, $bad failed, $(( $(date +%s) - T0 )) s" | tee $O/two_rank_loop_summary.txt
