#!/bin/bash
# Round 4, call 25 (as call 21, after the fine windows): the closing build after the second half of the round (records for the 2048 largest terms, second accumulator
# shapes, speculative thresholds): the suite twice, bench lines, makespan, kernel trace, PMC passes (their own runs), query shapes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/x; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], 'frac', r.get('frac'), 'eff', r.get('effective_frac'), 'exh', r.get('exhaustive', {}).get('frac') if isinstance(r.get('exhaustive'), dict) else None, 'spec', d.get('config',{}).get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
python -c "from nrtsearch_amd import build; print('build_id', build.build_id())"
for rep in 1; do
  el "suite $rep"
  timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --tb=short --durations=8 -p no:cacheprovider > $O/pytest_suite_$rep.log 2>&1
  echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite_$rep.log | tail -14 | cut -c1-200
done
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "bench lines"
for rep in 1 2; do timeout 250 python bench.py 2>/dev/null | tee $O/c3_line_$rep.json | show "c3 default (200 steps, full line)"; done
for rep in 1 2 3; do timeout 250 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20 --warmup 5 (driver's form)"; done
timeout 150 python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | tee $O/c2_line.json | show "c2"
timeout 150 python bench.py --packed --no-cpu-baseline --closed-loop '' 2>/dev/null | tee $O/c3_packed.json | show "c3 packed"
timeout 150 python bench.py --no-prune --no-cpu-baseline --closed-loop '' 2>/dev/null | tee $O/c3_noprune.json | show "c3 no-prune"
el "makespan (instrumented)"
timeout 200 python scripts/gpu_makespan.py --batches 2 --cus 248 2>/dev/null | tee $O/makespan.log | cut -c1-1500
el "kernel trace"
cd /tmp; rm -rf /tmp/prof
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r04 --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --steps 20 --warmup 5 > /tmp/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} $O/r04_kernel_stats.csv \;
head -6 $O/r04_kernel_stats.csv | cut -c1-60,200-420
el "PMC passes"
pmc() {  # name, counters..., then "--" and extra bench flags
  n=$1; shift; cs=""; while [ "$1" != "--" ]; do cs="$cs $1"; shift; done; shift
  rm -rf /tmp/pmc1; timeout 200 rocprofv3 --kernel-trace --pmc $cs -d /tmp/pmc1 -o p --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --warmup 1 --steps 4 --host-threads 1 "$@" > /tmp/pmc1.log 2>&1
  f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$n" <<'PY' | tee -a $O/r04_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'bm25' in k:
        print(sys.argv[2], k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()}, '(launches, mean per launch)')
PY
}
rm -f $O/r04_pmc.txt
pmc fetch_default FETCH_SIZE GRBM_GUI_ACTIVE --
pmc fetch_noprune FETCH_SIZE GRBM_GUI_ACTIVE -- --no-prune
pmc fetch_packed FETCH_SIZE GRBM_GUI_ACTIVE -- --packed
pmc sq1_default SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --
pmc sq2_default SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --
cd $ROOT
el "done"
