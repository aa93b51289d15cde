#!/bin/bash
# Round 3, call F: exchange modes / dist kNN (world of one), the new default bench line (200 steps, 10k queries, closed loop), one rank's
# share of an 8-GPU job with the exchange stage emulated (all-to-all vs all-gather, submitting / planner threads).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['kernel'], r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), c.get('host_cpus_busy_by_thread_kind'), c.get('dist_stage_ms'), d.get('closed_loop'))" "$1"; }
timeout 600 python -m pytest tests/test_exchange_gpu.py tests/test_vectors_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_f.log
timeout 600 python bench.py 2>$O/bench_f.err | tee $O/bench_f.json | show c3
E="--no-cpu-baseline --force-dist --emulate-world 8"
timeout 300 python bench.py $E --emulate-rank 0 2>/dev/null | tee $O/bench_f_emu8_r0_a2a.json | show emu8_r0_alltoall
timeout 300 python bench.py $E --emulate-rank 7 2>/dev/null | tee $O/bench_f_emu8_r7_a2a.json | show emu8_r7_alltoall
timeout 300 python bench.py $E --emulate-rank 0 --exchange-mode allgather 2>/dev/null | tee $O/bench_f_emu8_r0_ag.json | show emu8_r0_allgather
timeout 300 python bench.py $E --emulate-rank 0 --host-threads 3 2>/dev/null | tee $O/bench_f_emu8_r0_a2a_h3.json | show emu8_r0_alltoall_h3
timeout 300 python bench.py $E --emulate-rank 0 --planner-threads 1 2>/dev/null | tee $O/bench_f_emu8_r0_a2a_p1.json | show emu8_r0_alltoall_p1
timeout 300 python bench.py $E --emulate-rank 0 --planner-threads 2 --host-threads 3 2>/dev/null | tee $O/bench_f_emu8_r0_a2a_p2h3.json | show emu8_r0_alltoall_p2h3
timeout 300 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tee $O/bench_f_c4_emu8.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4_emu8', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_frac'])"
echo "== done =="
