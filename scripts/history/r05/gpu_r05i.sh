#!/bin/bash
# round 5, call i: one host wait for exchange + merge; spare CUs of the persistent launch on a 1/8 shard (development library, 3 repetitions)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05i; mkdir -p $O
timeout 600 python -m pytest tests/test_dist_two_ranks_gpu.py tests/test_exchange_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200
bash scripts/gpu_measure.sh r05i emulate8
bash scripts/gpu_measure.sh r05i_b emulate8
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| plan', r['host_plan_ms_per_step'], '| stages', c.get('dist_stage_ms'), '| p50/p99', d['p50_latency_ms'], d['p99_latency_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { name=$1; shift; env "$@" timeout 200 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 $EXTRA 2>/dev/null | tee $O/ab_$name.json | show "$name"; }
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for rep in 1 2 3; do
EXTRA="--emulate-peers final"
run final_spare8_$rep NRTGPU_MS_SPARE_CUS=8
run final_spare24_$rep NRTGPU_MS_SPARE_CUS=24
run final_spare48_$rep NRTGPU_MS_SPARE_CUS=48
EXTRA=""
run spec_spare8_$rep NRTGPU_MS_SPARE_CUS=8
run spec_spare24_$rep NRTGPU_MS_SPARE_CUS=24
done
