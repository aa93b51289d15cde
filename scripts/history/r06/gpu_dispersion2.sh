#!/bin/bash
# round 6, second pass: seeded re-runs (NRTGPU_SEED_RERUNS), fixed speculation settings (NRTGPU_SPEC_NO_VERDICT=1 + NRTGPU_MS_SCATTER)
# and the three-step ladder on the clustered / sorted corpora at C3's size; development library, same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06h}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_BENCH_WATCHDOG=400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| avg launch', r['avg_launch_ms'], 'ms | p50', d['p50_latency_ms'], '| spec', c.get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --c4-steps 0 --c2-steps 0 --c5-steps 0 --exhaustive-steps 0 --no-cpu-baseline --closed-loop "" --corpus-variant $VARIANT ${EXTRA:-} 2>/dev/null | tee $O/${TAG}_${VARIANT}_$name.json | show "$VARIANT $name"; }
{
for VARIANT in clustered sorted; do
  run ladder NRTGPU_X=0
  run ladder_unseeded NRTGPU_SEED_RERUNS=0
  EXTRA="--speculation-margin 0" run off NRTGPU_X=0
  run fixed_scattered_sqrt NRTGPU_SPEC_NO_VERDICT=1 NRTGPU_MS_SCATTER=1
  run fixed_scattered_sqrt_unseeded NRTGPU_SPEC_NO_VERDICT=1 NRTGPU_MS_SCATTER=1 NRTGPU_SEED_RERUNS=0
  run fixed_scattered_dispersion NRTGPU_SPEC_NO_VERDICT=1 NRTGPU_MS_SCATTER=3
  run fixed_docid_sqrt NRTGPU_SPEC_NO_VERDICT=1 NRTGPU_MS_SCATTER=0
done
VARIANT=iid
run ladder NRTGPU_X=0
} | tee $O/${TAG}_dispersion2.log
