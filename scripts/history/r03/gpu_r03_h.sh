#!/bin/bash
# Round 3, call H: begin / wait submission, deadlines + diagnostics; one rank's share of 2 / 4 / 8 GPUs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=50
timeout 300 python -m pytest tests/test_exchange_gpu.py tests/test_parity_gpu.py tests/test_abi.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_h.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_h.log | cut -c1-300
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), c.get('host_cpus_busy_by_thread_kind'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 60 python bench.py --no-cpu-baseline --force-dist "$@" 2>/dev/null | tee $O/bench_h_$n.json | show $n; }
run emu8_r0 --emulate-world 8 --emulate-rank 0
run emu8_r7 --emulate-world 8 --emulate-rank 7
run emu8_r0_sync --emulate-world 8 --emulate-rank 0 --sync-submit
run emu8_r0_p1 --emulate-world 8 --emulate-rank 0 --planner-threads 1
run emu8_r0_p4 --emulate-world 8 --emulate-rank 0 --planner-threads 4
run emu4_r0 --emulate-world 4 --emulate-rank 0
run emu4_r3 --emulate-world 4 --emulate-rank 3
run emu2_r0 --emulate-world 2 --emulate-rank 0
run emu2_r1 --emulate-world 2 --emulate-rank 1
run emu8_r0_ag --emulate-world 8 --emulate-rank 0 --exchange-mode allgather
echo "== done =="
