#!/bin/bash
# round 5, call e: the host side of one rank of eight -- launcher thread, planner chunks, staged peers' lists; A/B through the
# development build's knobs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_dist_two_ranks_gpu.py tests/test_shard_speculation_gpu.py tests/test_exchange_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
bash scripts/gpu_measure.sh r05e emulate8
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| plan', r['host_plan_ms_per_step'], '| stages', c.get('dist_stage_ms'), '| shard spec', c.get('shard_speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
E8="--force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop  --exhaustive-steps 0 --c4-steps 0 --submitters 1"
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 $EXTRA 2>/dev/null | tee $O/ab_$name.json | show "$name"; }
EXTRA="--emulate-peers final"
run final_dev X=1
run final_nolauncher NRTGPU_LAUNCHER=0
run final_spare16 NRTGPU_MS_SPARE_CUS=16
run final_spare32 NRTGPU_MS_SPARE_CUS=32
run final_planner4 X=1 ; 
EXTRA="--emulate-peers final --planner-threads 4"
run final_planner4 X=1
EXTRA=""
run spec_dev X=1
run spec_spare16 NRTGPU_MS_SPARE_CUS=16
run spec_nolauncher NRTGPU_LAUNCHER=0
unset NRTGPU_LIB_PATH
bash scripts/gpu_measure.sh r05e bench
