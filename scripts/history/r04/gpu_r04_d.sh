#!/bin/bash
# Round 4, call 3: persistent workgroups on the MaxScore route (one per CU, each choosing work until none is left) against one
# workgroup per item; per-CU gaps between pieces from the instrumented kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/d; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
bc3() { env "$@" timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0; }
el "suite (BM25 files)"
timeout 420 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fullsize_gpu.py -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_bm25.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_bm25.log | tail -8 | cut -c1-300
el "parity soak"
timeout 400 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee $O/parity_c3_2048.log
el "makespan"
for cfg in "NRTGPU_MS_PERSISTENT=0 NRTGPU_MS_HELPERS=0" "NRTGPU_MS_PERSISTENT=0" "NRTGPU_MS_PERSISTENT=1" "NRTGPU_MS_PERSISTENT=1 NRTGPU_MS_HELP_ALPHA=0"; do
  echo "-- $cfg"
  env $cfg timeout 200 python scripts/gpu_makespan.py --batches 2 2>/dev/null | grep -v last_batch | tee -a $O/makespan.log | cut -c1-1500
done
el "bench A/B"
for rep in 1 2; do
  bc3 2>/dev/null | tee $O/c3_persistent_$rep.json | show "c3 persistent (default)"
  bc3 NRTGPU_MS_PERSISTENT=0 2>/dev/null | tee $O/c3_per_item_$rep.json | show "c3 PERSISTENT=0"
done
bc3 NRTGPU_MS_HELP_ALPHA=0 2>/dev/null | tee $O/c3_persistent_alpha0.json | show "c3 persistent ALPHA=0"
bc3 NRTGPU_MS_HELP_ALPHA=24 2>/dev/null | tee $O/c3_persistent_alpha24.json | show "c3 persistent ALPHA=24"
bc3 NRTGPU_MS_HELPERS=0 2>/dev/null | tee $O/c3_persistent_helpers0.json | show "c3 persistent HELPERS=0"
bc3 NRTGPU_MS_HELP_MIN=8 2>/dev/null | tee $O/c3_persistent_min8.json | show "c3 persistent HELP_MIN=8"
for rep in 1 2; do timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20"; done
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --workload C2 2>/dev/null | tee $O/c2.json | show "c2"
NRTGPU_MS_PERSISTENT=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --workload C2 2>/dev/null | tee $O/c2_per_item.json | show "c2 PERSISTENT=0"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8.json | show "emu8"
NRTGPU_MS_PERSISTENT=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_per_item.json | show "emu8 PERSISTENT=0"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --packed 2>/dev/null | tee $O/c3_packed.json | show "c3 packed"
el "full bench line (new fields)"
timeout 200 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_full_line.json | cut -c1-3000
el "done"
