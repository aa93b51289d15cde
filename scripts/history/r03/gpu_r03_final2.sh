#!/bin/bash
# Round 3, closing measurement set on the final build -> gpurun_out/r03/final2 (copied into profiles/r03_* afterwards): the whole GPU
# suite + smoke; bench lines (C3 default with closed loop + CPU baseline, exhaustive A/B, packed, C2; C4 at 1 / 32 / 64 queries with the
# sketch, 32 without, its 1/8 share); rocprofv3 kernel stats of the default bench and of C4; FETCH_SIZE of the sketch kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03/final2; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - T0 )) s)"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_gpu.log | tail -14 | cut -c1-300
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['kernel'], r['avg_launch_ms'], r['frac'], r.get('effective_frac'), r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), (d.get('cpu_baseline') or {}).get('value'), d.get('closed_loop'), d.get('latency_outliers'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
timeout 150 python bench.py 2>$O/bench.err | tee $O/bench_line.json | show c3
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --no-prune --steps 60 2>/dev/null | tee $O/bench_line_no_prune.json | show c3_noprune
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --packed 2>/dev/null | tee $O/bench_c3_packed.json | show c3_packed
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --workload C2 2>/dev/null | tee $O/bench_c2.json | show c2
timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 40 --warmup 3 --closed-loop "1,8,64,512" 2>$O/bench_c4.err | tee $O/bench_c4_q32.json | show c4_q32
for q in 1 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $q --steps $([ $q = 1 ] && echo 120 || echo 40) --warmup 3 --no-cpu-baseline --closed-loop "" 2>/dev/null | tee $O/bench_c4_q$q.json | show c4_q$q
done
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 20 --warmup 3 --no-cpu-baseline --closed-loop "" --no-sketch 2>/dev/null | tee $O/bench_c4_q32_no_sketch.json | show c4_q32_fp32
timeout 100 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 40 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_c4_emulate8.json | show c4_emu8
for q in 1 32 64; do timeout 200 python bench.py --workload C4 --knn-queries $q --steps 40 --warmup 4 --no-cpu-baseline --no-verify --closed-loop "" --c4-callers 2>/dev/null | tee $O/bench_c4_q${q}_two_callers.json | show c4_q${q}_two; done
cd /tmp
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r03 --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --warmup 2 --steps 20 > /tmp/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} $O/r03_kernel_stats.csv \;
head -4 $O/r03_kernel_stats.csv | cut -c1-60,200-420
rm -rf /tmp/prof4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o c4 --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 2 --no-cpu-baseline --no-verify --closed-loop "" > /tmp/prof_c4.log 2>&1
find /tmp/prof4 -name "*kernel_stats*" -exec cp {} $O/r03_c4_kernel_stats.csv \;
grep "knn_" $O/r03_c4_kernel_stats.csv | cut -c1-50,150-330
rm -rf /tmp/pmc4; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/pmc4 -o p --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 3 --warmup 1 --no-cpu-baseline --no-verify --closed-loop "" > /tmp/pmc4.log 2>&1
f=$(find /tmp/pmc4 -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee $O/r03_c4_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'knn_sketch_kernel' in k or 'knn_select' in k:
        print(k, {c: (len(v), round(sum(v), 1)) for c, v in d.items()}, '(launches, SUM over launches; 4 passes of 3 launches: FETCH_SIZE in KB)')
PY
cd $ROOT
echo "== done ($(( $(date +%s) - T0 )) s) =="
