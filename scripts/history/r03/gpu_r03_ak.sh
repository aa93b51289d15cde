#!/bin/bash
# Round 3, call AK: two submitting threads per rank (begin / wait): one rank's share of 8 GPUs, ranks 0 and 7, peers' bounds played.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'plan', r.get('host_plan_ms_per_step'), 'segs', c['segments_per_gpu'], 'cpus', c.get('host_cpus_busy'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 150 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 "$@" 2>$O/bench_ak_$n.err | tee $O/bench_ak_$n.json | show $n; }
run r7_peers_s2 --emulate-rank 7 --emulate-peers final --submitters 2
run r0_peers_s2 --emulate-rank 0 --emulate-peers final --submitters 2
run r7_s2 --emulate-rank 7 --submitters 2
echo "== done =="
