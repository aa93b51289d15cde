#!/bin/bash
# more SQ counters of the scan kernel: LDS / VMEM latency levels, instruction fetch, lane utilisation
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
export TMPDIR=/tmp
V="${1:-0:0:1024}"
cd /tmp
for set in "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES"; do
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
