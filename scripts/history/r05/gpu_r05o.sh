#!/bin/bash
# round 5, call o: the planner over the whole C3 index (10 leaves) on the GPU box's host CPUs, without the GPU: 1 / 2 / 4 planner threads
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05o; mkdir -p $O
gcc -O1 -w -fPIC -shared -I/opt/rocm/include tests/mockhip/mockhip.c -o /tmp/libmockhip.so
for rep in 1 2; do
  for t in 1 2 4; do
    echo "10 leaves, $t planner thread(s): $(LD_PRELOAD=/tmp/libmockhip.so python scripts/cpu_plan_bench.py 1 $t 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/planner_threads.log
