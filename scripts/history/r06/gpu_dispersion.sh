#!/bin/bash
# round 6: speculation on docid-ordered corpora -- the measured-dispersion margin (MsArgs.scatter bit 1) and the follow-up launch of
# the re-run queries (NRTGPU_FOLLOW_UP), development library, C3 size, same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06g}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_BENCH_WATCHDOG=400
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], 'ms | p50', d['p50_latency_ms'], '| spec', c.get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" python bench.py --steps 100 --warmup 10 --c4-steps 0 --c2-steps 0 --c5-steps 0 --exhaustive-steps 0 --no-cpu-baseline --closed-loop "" --corpus-variant $VARIANT 2>/dev/null | tee $O/${TAG}_${VARIANT}_$name.json | show "$VARIANT $name"; }
{
VARIANT=clustered
run ladder_default NRTGPU_X=0
run docid_only NRTGPU_MS_SCATTER=0
run scattered_sqrt NRTGPU_MS_SCATTER=1
run docid_dispersion NRTGPU_MS_SCATTER=2
run scattered_dispersion NRTGPU_MS_SCATTER=3
VARIANT=sorted
run ladder_default NRTGPU_X=0
run ladder_no_follow_up NRTGPU_FOLLOW_UP=0
run scattered_dispersion NRTGPU_MS_SCATTER=3
run scattered_sqrt NRTGPU_MS_SCATTER=1
VARIANT=iid
run ladder_default NRTGPU_X=0
run scattered_dispersion NRTGPU_MS_SCATTER=3
} | tee $O/${TAG}_dispersion.log
