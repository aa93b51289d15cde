#!/bin/bash
# quick loop: MaxScore parity tests, C3 sweep (pruned, instrumented, small batches), one SQ counter pass
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_maxscore_gpu.py -x -q 2>&1 | tail -15
timeout 400 python scripts/gpu_sweep.py --queries 2048 --steps 8 --oracle-queries 8 --variants "${VARIANTS:-0:0:1024,0:1792:1024,0:0:256,0:0:64,0:0:8,0:0:1}" 2>&1 | tee gpurun_out/r02/sweep_quick.log | cut -c1-700
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/pmcx -o p --output-format csv -- python $ROOT/scripts/gpu_sweep.py --queries 1024 --steps 2 --oracle-queries 0 --variants "0:0:1024" > /tmp/pmcx.log 2>&1
f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee $ROOT/gpurun_out/r02/quick_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'maxscore' in k: print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
PY
