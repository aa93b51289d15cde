#!/bin/bash
# round 5, call l: finer doc windows for every query of a small shard (development library: NRTGPU_MS_FINE_ITEMS=1, NRTGPU_MS_FINE_SHIFT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05l; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| plan', r['host_plan_ms_per_step'], '| stages', c.get('dist_stage_ms'), '| shard spec', c.get('shard_speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dist --emulate-world $W --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 $EXTRA 2>/dev/null | tee $O/ab_$name.json | show "$name"; }
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
W=8
EXTRA=""
run spec8_base X=1
run spec8_fine_s1 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=1
run spec8_fine_s2 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=2
run spec8_fine_s3 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=3
EXTRA="--emulate-peers final"
run final8_base X=1
run final8_fine_s1 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=1
run final8_fine_s2 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=2
run final8_fine_s3 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=3
W=4
EXTRA=""
run spec4_base X=1
run spec4_fine_s1 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=1
run spec4_fine_s2 NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=2
