#!/bin/bash
# Round 3, call AH: replica groups instead of 8 shards: 8 GPUs as 2 groups x 4 shards (each group answers half of every 1024-query
# batch over quarter shards) or 4 groups x 2 shards: one rank's share on one GPU (peers' bounds played).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], 'batch', c['batch_queries'], 'ms_per_step', d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'plan', r.get('host_plan_ms_per_step'), 'segs', c['segments_per_gpu'], 'cpus', c.get('host_cpus_busy'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 150 python bench.py --no-cpu-baseline --force-dist "$@" 2>$O/bench_ah_$n.err | tee $O/bench_ah_$n.json | show $n; }
run g2x4_r0 --emulate-world 4 --emulate-rank 0 --batch 512 --emulate-peers final --planner-threads 2
run g2x4_r3 --emulate-world 4 --emulate-rank 3 --batch 512 --emulate-peers final --planner-threads 2
run g4x2_r0 --emulate-world 2 --emulate-rank 0 --batch 256 --emulate-peers final --planner-threads 2
run g4x2_r1 --emulate-world 2 --emulate-rank 1 --batch 256 --emulate-peers final --planner-threads 2
echo "== done =="
