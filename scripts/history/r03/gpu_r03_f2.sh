#!/bin/bash
# Round 3, call F2: one rank's share of an 8-GPU job with the exchange stage emulated (all-to-all vs all-gather, submitting / planner threads).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=50
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), c.get('host_cpus_busy_by_thread_kind'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
E="--no-cpu-baseline --force-dist --emulate-world 8"
run() { n=$1; shift; timeout 60 python bench.py $E "$@" 2>/dev/null | tee $O/bench_f_$n.json | show $n; }
run emu8_r0_a2a --emulate-rank 0
run emu8_r7_a2a --emulate-rank 7
run emu8_r0_ag --emulate-rank 0 --exchange-mode allgather
run emu8_r0_a2a_h3 --emulate-rank 0 --host-threads 3
run emu8_r0_a2a_p2 --emulate-rank 0 --planner-threads 2
run emu8_r0_a2a_p2h3 --emulate-rank 0 --planner-threads 2 --host-threads 3
run emu8_r0_a2a_p4h3 --emulate-rank 0 --planner-threads 4 --host-threads 3
timeout 120 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $O/bench_f_c4_emu8.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4_emu8', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_frac'])"
echo "== done =="
