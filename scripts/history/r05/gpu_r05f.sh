#!/bin/bash
# round 5, call f: one rank of eight -- hardware queues, planner threads, when the shard-level estimates are due
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05f; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| plan', r['host_plan_ms_per_step'], '| stages', c.get('dist_stage_ms'), '| shard spec', c.get('shard_speculation'), '| p50/p99', d['p50_latency_ms'], d['p99_latency_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 $EXTRA 2>/dev/null | tee $O/ab_$name.json | show "$name"; }
EXTRA="--emulate-peers final"
run final X=1
run final_q8 GPU_MAX_HW_QUEUES=8
run final_q16 GPU_MAX_HW_QUEUES=16
run final_q8_p4 GPU_MAX_HW_QUEUES=8 EXTRA2=1
EXTRA="--emulate-peers final --planner-threads 4"
run final_p4 X=1
run final_p4_q8 GPU_MAX_HW_QUEUES=8
EXTRA=""
run spec X=1
run spec_q8 GPU_MAX_HW_QUEUES=8
EXTRA="--speculation-margin 4"
run spec_z4 X=1
EXTRA="--speculation-margin 3"
run spec_z3 X=1
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
EXTRA=""
run spec_dev X=1
run spec_dev_first12 NRTGPU_MS_SPEC_FIRST=12
run spec_dev_first12_grow20 NRTGPU_MS_SPEC_FIRST=12 NRTGPU_MS_SPEC_GROW=20
run spec_dev_grow20 NRTGPU_MS_SPEC_GROW=20
run spec_dev_first12_grow24 NRTGPU_MS_SPEC_FIRST=12 NRTGPU_MS_SPEC_GROW=24
