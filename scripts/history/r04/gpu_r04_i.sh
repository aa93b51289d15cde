#!/bin/bash
# Round 4, call 8: the two lifetime tests again (live-handle counter; the pipeline test with its state dump), the suite with the
# turn ending behind the scorers as the default, the driver-form line, C5 at 50 M docs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/i; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'frac', r.get('frac'), 'p50', d['p50_latency_ms'], 'max', d.get('max_latency_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
el "lifetime tests"
timeout 300 python -m pytest tests/test_exchange_gpu.py::test_segment_release_while_searches_are_in_flight tests/test_exchange_gpu.py::test_begin_wait_pipeline_against_a_writer_of_the_same_segments -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_new.log | tail -30 | cut -c1-400
el "suite"
timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -12 | cut -c1-300
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "bench"
for rep in 1 2 3; do timeout 250 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20 --warmup 5"; done
timeout 250 python bench.py 2>/dev/null | tee $O/c3_line.json | show "c3 default (200 steps)"
el "C5: 50 M docs"
timeout 500 python scripts/gpu_c5_hybrid.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tee $O/c5_hybrid_50M.log | cut -c1-500
el "done"
