#!/bin/bash
# Round 3, call AF: one rank's share of 8 GPUs with 4 planner threads per rank (a node with >= 32 CPUs; this pool's boxes grant 16,
# hence 2 per rank in every other emulation): index layout, ranks 0 and 7, the peers' bounds played.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'plan', r.get('host_plan_ms_per_step'), 'segs', c['segments_per_gpu'], 'pl_thr', c['planner_threads'], 'cpus', c.get('host_cpus_busy'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 150 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 "$@" 2>$O/bench_af_$n.err | tee $O/bench_af_$n.json | show $n; }
run r0_peers_p4 --emulate-rank 0 --emulate-peers final --planner-threads 4
run r7_peers_p4 --emulate-rank 7 --emulate-peers final --planner-threads 4
run r7_p4 --emulate-rank 7 --planner-threads 4
echo "== done =="
