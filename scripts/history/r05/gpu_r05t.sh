#!/bin/bash
# round 5, call t: the coalescer's leader leaves when the cohort is back (NRTGPU_CO_COHORT=0/1, development library): closed loop at 1 / 8 / 64 / 512 callers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05t; mkdir -p $O
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for rep in 1 2; do for v in 0 1; do
  NRTGPU_CO_COHORT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop "1,8,64,512" --exhaustive-steps 0 --c4-steps 0 2>/dev/null | tee $O/ab_cohort${v}_$rep.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); cl=d['closed_loop']; print('cohort rule $v rep $rep |', {k: (round(v['qps']), v['p50_ms'], v['p99_ms'], v['mean_batch']) for k, v in cl.items() if k != 'entry'})"
done; done
for v in 0 1; do
  NRTGPU_CO_COHORT=$v timeout 300 python bench.py --workload C2 --steps 20 --warmup 5 --no-cpu-baseline --closed-loop "1,8,64,512" 2>/dev/null | tee $O/ab_c2_cohort${v}.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); cl=d['closed_loop']; print('C2 cohort rule $v |', {k: (round(v['qps']), v['p50_ms'], v['p99_ms'], v['mean_batch']) for k, v in cl.items() if k != 'entry'})"
done
