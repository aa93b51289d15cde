#!/bin/bash
# Round 4, call 2: helpers v2 -- the items are a queue, a workgroup helps an item on the launch's critical path while items are
# still queued (NRTGPU_MS_HELP_ALPHA x 16), any unfinished item once the queue is empty.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/c; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
bc3() { env "$@" timeout 100 python bench.py --no-cpu-baseline --closed-loop ''; }
el "suite"
timeout 420 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_suite.log | tail -8 | cut -c1-300
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "parity soak"
timeout 400 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee $O/parity_c3_2048.log
el "makespan"
for a in 0 16 10; do
  echo "-- NRTGPU_MS_HELP_ALPHA=$a"
  NRTGPU_MS_HELP_ALPHA=$a timeout 200 python scripts/gpu_makespan.py 2>/dev/null | tee $O/makespan_alpha$a.log | cut -c1-1000
done
el "bench A/B"
for rep in 1 2; do
  bc3 2>/dev/null | tee $O/c3_default_$rep.json | show "c3 default (alpha 16)"
  bc3 NRTGPU_MS_HELP_ALPHA=0 2>/dev/null | tee $O/c3_alpha0_$rep.json | show "c3 ALPHA=0 (tail helpers only)"
done
for a in 6 10 13 20 28; do bc3 NRTGPU_MS_HELP_ALPHA=$a 2>/dev/null | tee $O/c3_alpha$a.json | show "c3 ALPHA=$a"; done
bc3 NRTGPU_MS_HELPERS=0 2>/dev/null | tee $O/c3_helpers0.json | show "c3 HELPERS=0"
bc3 NRTGPU_MS_LPT=1 2>/dev/null | tee $O/c3_lpt1.json | show "c3 LPT=1"
bc3 NRTGPU_MS_HELP_GREEDY=1 2>/dev/null | tee $O/c3_greedy.json | show "c3 GREEDY=1"
bc3 NRTGPU_MS_HELP_MIN=8 2>/dev/null | tee $O/c3_min8.json | show "c3 HELP_MIN=8"
bc3 NRTGPU_MS_HELP_MIN=32 2>/dev/null | tee $O/c3_min32.json | show "c3 HELP_MIN=32"
for ti in 512 2048; do timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --target-items $ti 2>/dev/null | tee $O/c3_target_items_$ti.json | show "c3 --target-items $ti"; done
for rep in 1 2 3; do timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20"; done
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --workload C2 2>/dev/null | tee $O/c2_default.json | show "c2 default"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_default.json | show "emu8 default"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --packed 2>/dev/null | tee $O/c3_packed.json | show "c3 packed"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --no-prune 2>/dev/null | tee $O/c3_noprune.json | show "c3 no-prune"
el "done"
