#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 400 python -m pytest tests -m gpu -q --timeout 90 2>&1 | tail -5
echo "== sweep =="
timeout 600 python scripts/gpu_sweep.py --docs 10000000 --queries 2048 --steps 4 --oracle-queries 2 --variants "0:0:1024,0:1792:1024,0:0:2048,0:0:256,0:0:64" 2>&1 | cut -c1-900
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU_TRANS"; do
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcx -o p --output-format csv -- python $ROOT/scripts/gpu_sweep.py --docs 10000000 --queries 1024 --steps 2 --oracle-queries 0 --variants "0:0:1024" > /tmp/pmcx.log 2>&1
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
echo "== done =="
