#!/bin/bash
# Round 2: kernel trace + PMC passes of the MaxScore kernel at C3 (1024-query batches).  Separate passes, kernel trace only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r02
V="${1:-0:0:1024}"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python $ROOT/scripts/gpu_sweep.py --queries 2048 --steps 8 --oracle-queries 0 --variants "$V" > /tmp/kt.log 2>&1
find /tmp/kt -name "*kernel_stats*" -exec cp {} $ROOT/gpurun_out/r02/maxscore_kernel_stats.csv \;
head -8 $ROOT/gpurun_out/r02/maxscore_kernel_stats.csv | cut -c1-200
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcx -o p --output-format csv -- python $ROOT/scripts/gpu_sweep.py --queries 1024 --steps 2 --oracle-queries 0 --variants "$V" > /tmp/pmcx.log 2>&1
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a $ROOT/gpurun_out/r02/maxscore_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'maxscore' in k or 'scan' in k: print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
PY
  else tail -3 /tmp/pmcx.log; fi
  rm -rf /tmp/pmcx
done
