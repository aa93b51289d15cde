#!/bin/bash
# Round 4, call 1: the helper workgroups of the MaxScore route (plan.h: DHelp).
#   1. the whole -m gpu suite on the default build (helpers on, the two-clause launch order);
#   2. full-size parity soak (2048 C3 queries x top-1000, three delete configurations) -- batches larger than the CU count: helpers engage;
#   3. makespan against the balanced load from the instrumented kernel, helpers off / on;
#   4. C3 bench A/B: helpers x launch order x thresholds; --steps 20 as the driver runs it;
#   5. exact kNN: the scalar-norms build (knn.hip: -DNRT_KNN_SCALAR_NORMS), parity + C4 at 32 / 64 queries;
#   6. kernel trace of the default bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/b; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
el "suite"
timeout 420 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_suite.log | tail -8 | cut -c1-300
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "parity soak"
timeout 400 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" | tee $O/parity_c3_2048.log
el "makespan"
NRTGPU_MS_HELPERS=0 timeout 200 python scripts/gpu_makespan.py 2>/dev/null | tee $O/makespan_helpers0.log | cut -c1-900
timeout 200 python scripts/gpu_makespan.py 2>/dev/null | tee $O/makespan_default.log | cut -c1-900
NRTGPU_MS_HELP_GREEDY=1 timeout 200 python scripts/gpu_makespan.py 2>/dev/null | tee $O/makespan_greedy.log | cut -c1-900
el "bench A/B"
bc3() { env "$@" timeout 100 python bench.py --no-cpu-baseline --closed-loop ''; }   # bc3 [VAR=value ...]
for rep in 1 2; do
  bc3 2>/dev/null | tee $O/c3_default_$rep.json | show "c3 default"
  bc3 NRTGPU_MS_HELPERS=0 2>/dev/null | tee $O/c3_helpers0_$rep.json | show "c3 HELPERS=0"
done
bc3 NRTGPU_MS_LPT=0 2>/dev/null | tee $O/c3_lpt0.json | show "c3 LPT=0"
bc3 NRTGPU_MS_LPT=0 NRTGPU_MS_HELPERS=0 2>/dev/null | tee $O/c3_lpt0_helpers0.json | show "c3 LPT=0 HELPERS=0 (round 3)"
bc3 NRTGPU_MS_HELP_MIN=6 2>/dev/null | tee $O/c3_min6.json | show "c3 HELP_MIN=6"
bc3 NRTGPU_MS_HELP_MIN=40 2>/dev/null | tee $O/c3_min40.json | show "c3 HELP_MIN=40"
bc3 NRTGPU_MS_HELP_GREEDY=1 2>/dev/null | tee $O/c3_greedy.json | show "c3 HELP_GREEDY=1"
bc3 NRTGPU_MS_HELPERS=256 2>/dev/null | tee $O/c3_h256.json | show "c3 HELPERS=256"
bc3 NRTGPU_MS_HELPERS=4096 2>/dev/null | tee $O/c3_h4096.json | show "c3 HELPERS=4096"
bc3 NRTGPU_OVERLAP_SCORERS=1 2>/dev/null | tee $O/c3_overlap.json | show "c3 OVERLAP_SCORERS=1"
for rep in 1 2 3; do timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20"; done
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --workload C2 2>/dev/null | tee $O/c2_default.json | show "c2 default"
NRTGPU_MS_HELPERS=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --workload C2 2>/dev/null | tee $O/c2_helpers0.json | show "c2 HELPERS=0"
timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_default.json | show "emu8 default"
NRTGPU_MS_HELPERS=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_helpers0.json | show "emu8 HELPERS=0"
timeout 150 python bench.py --no-cpu-baseline --closed-loop '1,8,64,512' 2>/dev/null | tee $O/c3_closed_loop.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('closed loop', d.get('closed_loop'))"
el "kernel trace"
cd /tmp; rm -rf /tmp/prof_c3
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' > /tmp/prof_c3.log 2>&1
find /tmp/prof_c3 -name "*kernel_stats*" -exec cp {} $O/c3_kernel_stats.csv \;
head -6 $O/c3_kernel_stats.csv | cut -c1-60,140-330
cd $ROOT
el "kNN scalar norms A/B"
AB=/tmp/libnrtgpu_scalar_norms.so
timeout 300 python - <<'PY' 2>&1 | tail -1
from nrtsearch_amd import build
print(build.build(force=True, extra=["-DNRT_KNN_SCALAR_NORMS"], out="/tmp/libnrtgpu_scalar_norms.so"))
PY
if [ -f $AB ]; then
  python scripts/kernel_resources.py $AB | grep "sketch_kernel" | tee $O/ab_kernel_resources.txt
  NRTGPU_LIB_PATH=$AB NRT_KNN_FUZZ_ROUNDS=64 timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_fuzz_gpu.py::test_fuzz_exact_vector_search tests/test_baseline_sizes_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_ab.log 2>&1
  echo "pytest (A/B) rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_ab.log | tail -4 | cut -c1-300
  for q in 32 64; do
    timeout 200 python bench.py --workload C4 --knn-queries $q --steps 40 --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/c4_q${q}_shipped.json | show "q$q shipped"
    NRTGPU_LIB_PATH=$AB timeout 200 python bench.py --workload C4 --knn-queries $q --steps 40 --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/c4_q${q}_ab.json | show "q$q a/b"
  done
fi
el "done"
