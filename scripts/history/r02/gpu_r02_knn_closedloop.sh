#!/bin/bash
# Round 2: closed loop (C concurrent clients, coalesced single-query calls) on the MaxScore route; kNN kernels at
# config C4's size under rocprofv3 (kernel trace, then two PMC passes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r02
echo "== closed loop C3 =="
timeout 300 python scripts/gpu_closed_loop.py 2>&1 | tee gpurun_out/r02/closed_loop_c3.log
cd /tmp
echo "== knn C4: rocprofv3 --kernel-trace --stats =="
KNN_N=10000000 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/knnp -o knn --output-format csv -- python $ROOT/scripts/gpu_knn_bench.py > $ROOT/gpurun_out/r02/knn_trace.log 2>&1
tail -5 $ROOT/gpurun_out/r02/knn_trace.log
find /tmp/knnp -name "*kernel_stats*" -exec cp {} $ROOT/gpurun_out/r02/r02_knn_kernel_stats.csv \;
head -5 $ROOT/gpurun_out/r02/r02_knn_kernel_stats.csv | cut -c1-50,200-330
rm -f $ROOT/gpurun_out/r02/r02_knn_pmc.txt
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_MFMA SQ_WAVES"; do
  KNN_N=10000000 timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/knn1 -o p --output-format csv -- python $ROOT/scripts/gpu_knn_bench.py > /tmp/knn1.log 2>&1
  tail -1 /tmp/knn1.log | cut -c1-200
  f=$(find /tmp/knn1 -name "*counter_collection.csv" | head -1)
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
  rm -rf /tmp/knn1
done
echo "== done =="
