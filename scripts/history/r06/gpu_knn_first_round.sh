#!/bin/bash
# round 6: the exact vector search's pass by the size of its first round (rows that take a slot each before any theta exists;
# NRTGPU_KNN_FIRST_ROUND, development library): C4, 64 queries and 1 query per pass
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06k}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
for fr in ${FIRST_ROUNDS:-65536 32768 16384 8192 4096}; do for q in 64 1; do
  NRTGPU_KNN_FIRST_ROUND=$fr timeout 300 python bench.py --workload C4 --knn-queries $q --steps 20 --warmup 4 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('first_round $fr q$q', d['value'], d['ms_per_step'], 'kernels', r['avg_launch_ms'], 'frac', r['frac'])"
done; done | tee $O/${TAG}_first_round.log
