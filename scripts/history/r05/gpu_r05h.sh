#!/bin/bash
# round 5, call h: one rank of eight -- the exchange stage on streams of the highest priority
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05h; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| plan', r['host_plan_ms_per_step'], '| stages', c.get('dist_stage_ms'), '| shard spec', c.get('shard_speculation'), '| p50/p99', d['p50_latency_ms'], d['p99_latency_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 $EXTRA 2>/dev/null | tee $O/ab_$name.json | show "$name"; }
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for rep in 1 2; do
EXTRA="--emulate-peers final"
run final_$rep X=1
EXTRA="--emulate-peers final --exchange-priority high"
run final_xhigh_$rep X=1
run final_xhigh_mergehigh_$rep NRTGPU_HI_PRIORITY=1
EXTRA=""
run spec_$rep X=1
EXTRA="--exchange-priority high"
run spec_xhigh_mergehigh_$rep NRTGPU_HI_PRIORITY=1
done
