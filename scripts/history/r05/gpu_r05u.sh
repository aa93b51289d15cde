#!/bin/bash
# round 5, call u: the kNN coalescer's leader waits for the last panel's cohort (NRTGPU_KCO_COHORT=0/1, development library): C4 closed loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05u; mkdir -p $O
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for v in 0 1; do
  NRTGPU_KCO_COHORT=$v timeout 500 python bench.py --workload C4 --knn-queries 64 --steps 6 --warmup 2 --no-cpu-baseline --no-verify --closed-loop "1,4,8,16,64,512" 2>/dev/null | tee $O/ab_c4_cohort$v.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); cl=d['closed_loop']; print('C4 knn cohort rule $v |', {k: (round(v['qps']), v['p50_ms'], v['p99_ms'], v.get('mean_panel')) for k, v in cl.items() if k != 'entry'})"
done
unset NRTGPU_LIB_PATH
timeout 400 python -m pytest tests/test_vectors_gpu.py tests/test_maxscore_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "coalesc" 2>&1 | tail -2 | cut -c1-200
