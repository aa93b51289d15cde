#!/bin/bash
# Round 3, call AG: where a rank's planning time goes (NRTGPU_PLAN_TRACE): rank 0 (one leaf) and rank 7 (seven) of an 8-GPU C3 job.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
for R in 0 7; do
  NRTGPU_PLAN_TRACE=1 timeout 100 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 --emulate-rank $R --steps 40 --warmup 5 2>$O/plan_trace_r$R.err >/dev/null
  grep "nrtgpu plan" $O/plan_trace_r$R.err | tail -30 | awk '{r+=$6; c+=$11; k+=$13; n+=1} END {printf "rank '$R': resolve %.3f concat %.3f cut+items %.3f ms (mean of %d)\n", r/n, c/n, k/n, n}'
  grep "nrtgpu plan" $O/plan_trace_r$R.err | tail -2
done
echo "== done =="
