#!/bin/bash
# End-of-round refresh of profiles/r02_*: bench lines (C3 default / exhaustive / packed / C2 / per-rank share of 8 and 2 GPUs /
# C4), rocprofv3 kernel stats + FETCH_SIZE of the default bench, kNN kernels under rocprofv3 (stats, FETCH_SIZE, MFMA
# counters), the closed loop.  Counters in passes of their own, kernel trace only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r02; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['kernel'], r['avg_launch_ms'], r['achieved'], r['frac'], r.get('host_plan_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))" "$1"; }
timeout 600 python bench.py 2>$O/bench.err | tee $O/bench_line.json | show c3
timeout 300 python bench.py --no-cpu-baseline --no-prune | tee $O/bench_line_no_prune.json | show c3_noprune
timeout 300 python bench.py --no-cpu-baseline --packed | tee $O/bench_c3_packed.json | show c3_packed
timeout 300 python bench.py --no-cpu-baseline --workload C2 | tee $O/bench_c2.json | show c2
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | tee $O/bench_emulate8.json | show emu8
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 2 2>/dev/null | tee $O/bench_emulate2.json | show emu2
for q in 1 32 64; do
  timeout 600 python bench.py --workload C4 --knn-queries $q --steps 10 --warmup 5 $([ $q != 32 ] && echo --no-cpu-baseline) 2>/dev/null | tee $O/bench_c4_q$q.json | show c4_q$q
done
cd /tmp
rm -rf /tmp/prof; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r02 --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} $O/r02_kernel_stats.csv \;
head -4 $O/r02_kernel_stats.csv | cut -c1-60,240-420
rm -rf /tmp/pmcb; timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/pmcb -o b --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > /tmp/pmcb.log 2>&1
f=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee $O/r02_pmc_fetch.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
PY
# kNN at C4 (bench.py --workload C4: 10M x 768, 32 queries per step)
rm -rf /tmp/knnp; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/knnp -o knn --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 8 --warmup 2 --no-cpu-baseline > /tmp/knnp.log 2>&1
find /tmp/knnp -name "*kernel_stats*" -exec cp {} $O/r02_knn_kernel_stats.csv \;
head -4 $O/r02_knn_kernel_stats.csv | cut -c1-50,200-330
rm -f $O/r02_knn_pmc.txt
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA SQ_WAVES"; do
  rm -rf /tmp/knn1; timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/knn1 -o p --output-format csv -- python $ROOT/bench.py --workload C4 --knn-queries 32 --steps 4 --warmup 1 --no-cpu-baseline > /tmp/knn1.log 2>&1
  f=$(find /tmp/knn1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/r02_knn_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'knn' in k:
        print(k, {c: (len(v), round(sum(v), 1), round(max(v), 1)) for c, v in d.items()}, '(launches, sum over launches, max)')
PY
done
cd $ROOT
timeout 300 python scripts/gpu_closed_loop.py 2>&1 | grep -v amdgpu.ids | tee $O/closed_loop_c3.log | tail -12
echo "== done =="
