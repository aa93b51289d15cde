#!/bin/bash
# round 5, call m: the planner's first pass on the GPU box's host CPUs, without the GPU (tests/mockhip preloaded): before / after the
# exponent-from-bits and stack-array changes; 1 and 2 planner threads; one rank of eight's shard
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05m; mkdir -p $O
gcc -O1 -w -fPIC -shared -I/opt/rocm/include tests/mockhip/mockhip.c -o /tmp/libmockhip.so
for rep in 1 2 3; do
  for t in 1 2; do
    echo "old, $t planner thread(s): $(LD_PRELOAD=/tmp/libmockhip.so NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_planner_old.so python scripts/cpu_plan_bench.py 8 $t 2>&1 | tail -1)"
    echo "new, $t planner thread(s): $(LD_PRELOAD=/tmp/libmockhip.so python scripts/cpu_plan_bench.py 8 $t 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/planner_ab.log
