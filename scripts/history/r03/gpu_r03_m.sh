#!/bin/bash
# Round 3, call M: exact vector search returns the oracle's bits (nominate on the matrix cores, rescore in the oracle's order,
# certify): vector tests, the exchange / hybrid tests that sit on it, C4 at 10M rows; rank 7's share of 8 GPUs again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=100
timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_m.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_m.log | tail -25 | cut -c1-400
timeout 300 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k knn > $O/pytest_m2.log 2>&1; echo "pytest sizes rc=$?"; tail -5 $O/pytest_m2.log | cut -c1-400
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline 2>$O/bench_m_c4.err | tee $O/bench_m_c4_q32.json | cut -c1-900
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy_by_thread_kind'), c.get('dist_stage_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { n=$1; shift; timeout 120 python bench.py --no-cpu-baseline --force-dist "$@" 2>$O/bench_m_$n.err | tee $O/bench_m_$n.json | show $n; }
run emu8_r7_peers --emulate-world 8 --emulate-rank 7 --emulate-peers final
run emu8_r7_plain --emulate-world 8 --emulate-rank 7
run emu8_r7_peers2 --emulate-world 8 --emulate-rank 7 --emulate-peers final
echo "== done =="
