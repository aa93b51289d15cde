#!/bin/bash
# Round 4, FIRST call (prepared at the end of round 3, when the GPU minutes were spent): the sketch kernel with the tile norms back
# in the scalar cache (profiles/r03_knn_sketch_isa_note.txt; knn.hip: -DNRT_KNN_SCALAR_NORMS).  The A/B library is built ON THE BOX
# (same image, same hipcc) next to the shipped one; NRTGPU_LIB_PATH selects it.
#   1. the vector parity tests + the kNN fuzz against the A/B library (bit-exact or it does not go in);
#   2. C4 at 1 / 32 / 64 queries and the 1/8 share with both libraries, same box, interleaved;
#   3. kernel trace of both (the sketch kernel's average launch).
#   4. the MaxScore items' launch order (the default two-clause key against NRTGPU_MS_LPT=0, round 3's measured order): parity suites
#      under the old order once more, C3 / 1-of-8 / C2 bench lines A/B.
# -> gpurun_out/r04/a.  If 1 is green and 2 is not slower: make the cast the only code path (drop the macro), rebuild, re-run
# scripts/kernel_resources.py > profiles/r04_kernel_resources.txt and drop `knn_sketch_kernel<4, 8>` from
# tests/test_kernel_resources.py: KNOWN_SCRATCH.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/a; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
AB=/tmp/libnrtgpu_scalar_norms.so
timeout 300 python - <<'PY' 2>&1 | tail -2
from nrtsearch_amd import build
print(build.build(force=True, extra=["-DNRT_KNN_SCALAR_NORMS"], out="/tmp/libnrtgpu_scalar_norms.so"))
PY
[ -f $AB ] || { echo "A/B build failed"; exit 1; }
python scripts/kernel_resources.py $AB | grep "sketch_kernel" | tee $O/ab_kernel_resources.txt
echo "== parity on the A/B library ($(( $(date +%s) - T0 )) s)"
NRTGPU_LIB_PATH=$AB NRT_KNN_FUZZ_ROUNDS=64 timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_fuzz_gpu.py::test_fuzz_exact_vector_search tests/test_baseline_sizes_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_ab.log 2>&1
echo "pytest (A/B) rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_ab.log | tail -6 | cut -c1-300
NRTGPU_LIB_PATH=$AB timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
echo "== C4, shipped vs A/B, interleaved ($(( $(date +%s) - T0 )) s)"
for rep in 1 2; do
  for q in 1 32 64; do
    st=$([ $q = 1 ] && echo 120 || echo 40)
    timeout 200 python bench.py --workload C4 --knn-queries $q --steps $st --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/c4_q${q}_shipped_$rep.json | show "q$q shipped"
    NRTGPU_LIB_PATH=$AB timeout 200 python bench.py --workload C4 --knn-queries $q --steps $st --warmup 3 --no-cpu-baseline --no-verify --closed-loop "" 2>/dev/null | tee $O/c4_q${q}_ab_$rep.json | show "q$q a/b"
  done
done
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 40 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/c4_emu8_shipped.json | show "emu8 shipped"
NRTGPU_LIB_PATH=$AB timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 40 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/c4_emu8_ab.json | show "emu8 a/b"
echo "== launch order of the MaxScore items: the default (two heaviest clauses) against NRTGPU_MS_LPT=0 (all postings: round 3's measured order); scripts/cpu_launch_order_sim.py predicts 1.56 -> 1.33 x the balanced load ($(( $(date +%s) - T0 )) s)"
NRTGPU_MS_LPT=0 timeout 600 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_sizes_gpu.py tests/test_exchange_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_ms_lpt.log 2>&1
echo "pytest (NRTGPU_MS_LPT=0; the default order runs in the round's own suite) rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_ms_lpt.log | tail -4 | cut -c1-300
for rep in 1 2; do
  timeout 100 python bench.py --no-cpu-baseline --closed-loop "" 2>/dev/null | tee $O/c3_default_$rep.json | show "c3 default (two-clause key)"
  NRTGPU_MS_LPT=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop "" 2>/dev/null | tee $O/c3_ms_lpt0_$rep.json | show "c3 NRTGPU_MS_LPT=0 (round 3 order)"
done
timeout 100 python bench.py --no-cpu-baseline --closed-loop "" --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_default.json | show "emu8 default (two-clause key)"
NRTGPU_MS_LPT=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop "" --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_ms_lpt0.json | show "emu8 NRTGPU_MS_LPT=0"
for ti in 512 2048; do timeout 100 python bench.py --no-cpu-baseline --closed-loop "" --target-items $ti 2>/dev/null | tee $O/c3_target_items_$ti.json | show "c3 --target-items $ti (queries cut by all postings: smaller items, more cold starts)"; done
NRTGPU_OVERLAP_SCORERS=1 timeout 100 python bench.py --no-cpu-baseline --closed-loop "64,512" 2>/dev/null | tee $O/c3_overlap_scorers.json | show "c3 NRTGPU_OVERLAP_SCORERS=1"
python -c "import json; d=json.loads(open('$O/c3_overlap_scorers.json').read().strip().splitlines()[-1]); print('   closed loop:', d.get('closed_loop'), 'p50', d.get('p50_latency_ms'))" 2>/dev/null
NRTGPU_OVERLAP_SCORERS=1 timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_maxscore_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2
timeout 100 python bench.py --no-cpu-baseline --closed-loop "" --workload C2 2>/dev/null | tee $O/c2_default.json | show "c2 default (two-clause key)"
NRTGPU_MS_LPT=0 timeout 100 python bench.py --no-cpu-baseline --closed-loop "" --workload C2 2>/dev/null | tee $O/c2_ms_lpt0.json | show "c2 NRTGPU_MS_LPT=0"
echo "== kernel traces ($(( $(date +%s) - T0 )) s)"
cd /tmp
for v in shipped ab; do
  rm -rf /tmp/prof_$v
  [ $v = ab ] && export NRTGPU_LIB_PATH=$AB || unset NRTGPU_LIB_PATH
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o c4 --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 64 --steps 10 --warmup 2 --no-cpu-baseline --no-verify --closed-loop "" > /tmp/prof_$v.log 2>&1
  find /tmp/prof_$v -name "*kernel_stats*" -exec cp {} $O/c4_q64_kernel_stats_$v.csv \;
  echo "$v:"; grep "knn_" $O/c4_q64_kernel_stats_$v.csv | cut -c1-50,150-330
done
unset NRTGPU_LIB_PATH
cd $ROOT
echo "== done ($(( $(date +%s) - T0 )) s) =="
