#!/bin/bash
# Round 3, call L: one rank's share of 8 / 4 / 2 GPUs with the bound exchange open and the other ranks' rows played by the bench
# (--emulate-peers final: the optimistic end); a 2-rank job on one GPU with the exchange on by default (the real table, two processes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('other_scorer_ms_per_step'), c.get('host_cpus_busy'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 120 python bench.py --no-cpu-baseline --force-dist "$@" 2>$O/bench_l_$n.err | tee $O/bench_l_$n.json | show $n; }
run emu8_r0_peers --emulate-world 8 --emulate-rank 0 --emulate-peers final
run emu8_r7_peers --emulate-world 8 --emulate-rank 7 --emulate-peers final
run emu4_r0_peers --emulate-world 4 --emulate-rank 0 --emulate-peers final
run emu2_r0_peers --emulate-world 2 --emulate-rank 0 --emulate-peers final
timeout 200 python bench.py --no-cpu-baseline --gpus 2 --debug-same-gpu --steps 40 --warmup 4 2>$O/bench_l_same2.err | tee $O/bench_l_same2.json | show same2
tail -3 $O/bench_l_same2.err | cut -c1-300
echo "== done =="
