#!/bin/bash
# PMC passes (counters only, with --kernel-trace) over the sweep driver; results -> gpurun_out/pmc_*
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|LDSBankConflict|MemUnitBusy|VALUBusy|SALUBusy|L2CacheHit|MeanOccupancy[A-Za-z]*)" | sort -u > $ROOT/gpurun_out/pmc/counters.txt
wc -l $ROOT/gpurun_out/pmc/counters.txt
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc$i -o p$i --output-format csv -- python $ROOT/scripts/gpu_sweep.py --docs 10000000 --queries 1024 --steps 2 --oracle-queries 0 --variants "0:0:1024" > /tmp/pmc$i.log 2>&1
  tail -2 /tmp/pmc$i.log | cut -c1-300
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  echo "pass $i: $set -> $f"
  if [ -n "$f" ]; then
    python - "$f" <<'PY' | tee $ROOT/gpurun_out/pmc/pass$i.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r.get('Kernel_Name', '')[:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'scan' in k or 'merge' in k:
        print(k, {c: v for c, v in d.items()})
PY
  fi
done
