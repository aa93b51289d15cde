#!/bin/bash
# round 5, call w: from how many (query, leaf) pairs the planner uses its helpers (NRTGPU_PLAN_ALONE_PAIRS 2048 / 8192, development library): closed loop at 512 callers
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05w; mkdir -p $O
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for rep in 1 2 3; do for v in 2048 8192; do
  NRTGPU_PLAN_ALONE_PAIRS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --closed-loop "64,512" --exhaustive-steps 0 --c4-steps 0 2>/dev/null | tee $O/ab_alone${v}_$rep.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); cl=d['closed_loop']; print('helpers from $v pairs, rep $rep |', {k: (round(v['qps']), v['p50_ms'], v['p99_ms'], v['mean_batch']) for k, v in cl.items() if k != 'entry'})"
done; done
