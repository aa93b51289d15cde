#!/bin/bash
# round 6: the walk asks for the NEXT clause's records together with this clause's codes (-DNRT_MS_PIPE=1, plan.h: kMsPipe) against the
# product order; parity of the variant first (the walk's tests through the variant library), then same box, interleaved bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06p}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
VARIANT=${2:-$ROOT/nrtsearch_amd/libnrtgpu_pipe.so}
if [ "${3:-}" != "notests" ]; then
  NRTGPU_LIB_PATH=$VARIANT timeout 900 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -5 | tee $O/${TAG}_variant_tests.log
fi
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], 'ms | build', r.get('build_id'), '| spec', c.get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for rep in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export NRTGPU_LIB_PATH=$VARIANT; else unset NRTGPU_LIB_PATH; fi
    timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 2>/dev/null | tee $O/${TAG}_${lib}_$rep.json | show "$lib rep $rep"
  done
done | tee $O/${TAG}_pipe_ab.log
unset NRTGPU_LIB_PATH
