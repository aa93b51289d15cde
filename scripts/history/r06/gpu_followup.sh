#!/bin/bash
# round 6: the second pass of a speculative call as a SEEDED FOLLOW-UP launch (NRTGPU_SEED_RERUNS, NRTGPU_FOLLOW_UP) on the
# sorted / clustered corpora at C3's size, the library's own two-step verdict; development library, same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06i}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_BENCH_WATCHDOG=400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| avg launch', r['avg_launch_ms'], 'ms | p50', d['p50_latency_ms'], 'max', d['max_latency_ms'], '| spec', c.get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --c4-steps 0 --c2-steps 0 --c5-steps 0 --exhaustive-steps 0 --no-cpu-baseline --closed-loop "" --corpus-variant $VARIANT ${EXTRA:-} 2>/dev/null | tee $O/${TAG}_${VARIANT}_$name.json | show "$VARIANT $name"; }
{
for VARIANT in sorted clustered; do
  run seeded_follow_up NRTGPU_X=0
  run seeded_own_turn NRTGPU_FOLLOW_UP=0
  run unseeded_follow_up NRTGPU_SEED_RERUNS=0
  run unseeded_own_turn NRTGPU_SEED_RERUNS=0 NRTGPU_FOLLOW_UP=0
  EXTRA="--speculation-margin 0" run off NRTGPU_X=0
  EXTRA="--host-threads 3" run seeded_follow_up_3_threads NRTGPU_X=0
done
VARIANT=iid
run default NRTGPU_X=0
} | tee $O/${TAG}_followup.log
