#!/bin/bash
# Round 3, call AE: one rank's share of an 8-GPU C3 job with the BALANCED shard layout (every rank 2-3 leaves): ranks 0, 4 (three
# leaves), 7, shards on their own and with the peers' bounds played (the index layout's last rank, 7 small segments, beside them).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'plan', r.get('host_plan_ms_per_step'), 'segs', c['segments_per_gpu'], 'cpus', c.get('host_cpus_busy'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 150 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 "$@" 2>$O/bench_ae_$n.err | tee $O/bench_ae_$n.json | show $n; }
run bal_r0_peers --shard-layout balanced --emulate-rank 0 --emulate-peers final
run bal_r4_peers --shard-layout balanced --emulate-rank 4 --emulate-peers final
run bal_r7_peers --shard-layout balanced --emulate-rank 7 --emulate-peers final
run bal_r4 --shard-layout balanced --emulate-rank 4
run idx_r7_peers --emulate-rank 7 --emulate-peers final
echo "== done =="
