#!/bin/bash
# Round 4, call 7: lifetime tests (release under searches, begin / wait next to a writer, vector relation); the turn ending
# before the merge (A/B); C5 at 50 M docs on one GPU.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/h; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'p50', d['p50_latency_ms'], 'p99', d.get('p99_latency_ms'), 'max', d.get('max_latency_ms'), 'merge', r.get('merge_ms_per_step'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
bc3() { env "$@" timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0; }
el "new tests"
timeout 400 python -m pytest tests/test_exchange_gpu.py::test_segment_release_while_searches_are_in_flight tests/test_exchange_gpu.py::test_begin_wait_pipeline_against_a_writer_of_the_same_segments tests/test_vectors_gpu.py::test_exact_vector_query_relation_is_the_reference_collectors -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_new.log | tail -30 | cut -c1-220
el "turn before merge A/B"
for rep in 1 2; do
  bc3 2>/dev/null | tee $O/c3_default_$rep.json | show "c3 default"
  bc3 NRTGPU_TURN_BEFORE_MERGE=1 2>/dev/null | tee $O/c3_tbm_$rep.json | show "c3 TURN_BEFORE_MERGE=1"
done
bc3 NRTGPU_TURN_BEFORE_MERGE=1 NRTGPU_MS_SPARE_CUS=12 2>/dev/null | tee $O/c3_tbm_spare12.json | show "c3 TURN_BEFORE_MERGE=1 SPARE=12"
bc3 NRTGPU_TURN_BEFORE_MERGE=1 NRTGPU_MS_SPARE_CUS=16 2>/dev/null | tee $O/c3_tbm_spare16.json | show "c3 TURN_BEFORE_MERGE=1 SPARE=16"
for rep in 1 2 3; do
  timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 2>/dev/null | tee $O/c3_s20_default_$rep.json | show "c3 --steps 20 default"
  NRTGPU_TURN_BEFORE_MERGE=1 timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 2>/dev/null | tee $O/c3_s20_tbm_$rep.json | show "c3 --steps 20 TURN_BEFORE_MERGE=1"
done
NRTGPU_TURN_BEFORE_MERGE=1 timeout 150 python bench.py --no-cpu-baseline --exhaustive-steps 0 --steps 50 2>/dev/null | tee $O/c3_closed_tbm.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('closed loop TBM', d.get('closed_loop'))"
timeout 150 python bench.py --no-cpu-baseline --exhaustive-steps 0 --steps 50 2>/dev/null | tee $O/c3_closed_default.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('closed loop default', d.get('closed_loop'))"
NRTGPU_TURN_BEFORE_MERGE=1 timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_hybrid_gpu.py tests/test_exchange_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tail -3
el "C5: 50 M docs"
timeout 500 python scripts/gpu_c5_hybrid.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee $O/c5_hybrid_50M.log | cut -c1-400
el "done"
