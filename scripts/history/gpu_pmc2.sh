#!/bin/bash
# SQ counters of the scan kernel for the library in $NRTGPU_LIB_PATH (default: in-tree), variant $1
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
export TMPDIR=/tmp
V="${1:-0:0:1024}"
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR"; do
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcx -o p --output-format csv -- python $ROOT/scripts/gpu_sweep.py --docs 10000000 --queries 1024 --steps 2 --oracle-queries 0 --variants "$V" > /tmp/pmcx.log 2>&1
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r['Kernel_Name'][:30]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'scan' in k: print(k, dict(d))
PY
  rm -rf /tmp/pmcx
done
