#!/bin/bash
# Round 2, first GPU call: per-sub-tile cost of the scan on sparse-only queries; kNN kernels under rocprofv3
# (kernel trace + two PMC passes) at config C4's size.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r02
echo "== sweep: plain C3 / sparse-only (all ranks >= 100) / instrumented =="
timeout 300 python scripts/gpu_sweep.py --queries 1024 --steps 4 --oracle-queries 0 --variants "0:0:1024,0:1792:1024" 2>&1 | tee gpurun_out/r02/sweep_plain.log | cut -c1-900
timeout 300 python scripts/gpu_sweep.py --queries 1024 --steps 4 --oracle-queries 0 --min-rank 100 --variants "0:0:1024,0:1792:1024" 2>&1 | tee gpurun_out/r02/sweep_sparse100.log | cut -c1-900
timeout 300 python scripts/gpu_sweep.py --queries 1024 --steps 4 --oracle-queries 0 --min-rank 32 --variants "0:0:1024" 2>&1 | tee gpurun_out/r02/sweep_sparse32.log | cut -c1-600
cd /tmp
echo "== knn C4: rocprofv3 --kernel-trace --stats =="
KNN_N=10000000 timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/knnp -o knn --output-format csv -- python $ROOT/scripts/gpu_knn_bench.py > $ROOT/gpurun_out/r02/knn_trace.log 2>&1
tail -5 $ROOT/gpurun_out/r02/knn_trace.log
find /tmp/knnp -name "*kernel_stats*" -exec cp {} $ROOT/gpurun_out/r02/r02_knn_kernel_stats.csv \;
head -6 $ROOT/gpurun_out/r02/r02_knn_kernel_stats.csv
echo "== knn C4: PMC pass 1 (FETCH_SIZE) =="
KNN_N=10000000 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/knn1 -o p --output-format csv -- python $ROOT/scripts/gpu_knn_bench.py > /tmp/knn1.log 2>&1
tail -2 /tmp/knn1.log
echo "== knn C4: PMC pass 2 (SQ: MFMA / VALU / waves) =="
KNN_N=10000000 timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_F32 SQ_WAVES -d /tmp/knn2 -o p --output-format csv -- python $ROOT/scripts/gpu_knn_bench.py > /tmp/knn2.log 2>&1
tail -2 /tmp/knn2.log
for d in /tmp/knn1 /tmp/knn2; do
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $ROOT/gpurun_out/r02/r02_knn_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'knn' in k:
        print(k, {c: (len(v), round(sum(v) / len(v), 1), round(max(v), 1)) for c, v in d.items()})
PY
done
echo "== done =="
