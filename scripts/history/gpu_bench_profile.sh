#!/bin/bash
# Official bench line + rocprofv3 kernel stats + PMC traffic for the same command.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
echo "== bench.py (default) =="
timeout 900 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench_r01.json | cut -c1-1500
tail -3 gpurun_out/bench.err
echo "== bench.py --host-threads 1 =="
timeout 600 python bench.py --host-threads 1 --no-cpu-baseline --steps 10 2>/dev/null | tee gpurun_out/bench_r01_1thread.json | cut -c1-400
cd /tmp
echo "== rocprofv3 --kernel-trace --stats =="
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r01 --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_bench.log 2>&1
tail -1 /tmp/prof_bench.log | cut -c1-300
find /tmp/prof -name "*kernel_stats*" -exec cp {} $ROOT/gpurun_out/prof/r01_kernel_stats.csv \;
find /tmp/prof -name "*domain_stats*" -exec cp {} $ROOT/gpurun_out/prof/r01_domain_stats.csv \;
head -8 $ROOT/gpurun_out/prof/r01_kernel_stats.csv
echo "== rocprofv3 --pmc FETCH_SIZE =="
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d /tmp/pmcb -o b --output-format csv -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --host-threads 1 > /tmp/pmcb.log 2>&1
tail -1 /tmp/pmcb.log | cut -c1-200
f=$(find /tmp/pmcb -name "*counter_collection.csv" | head -1)
python - "$f" $ROOT/gpurun_out/prof/pmc_traffic.json <<'PY' | tee $ROOT/gpurun_out/prof/r01_pmc_fetch.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
import json
for k, d in agg.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
    if 'bm25_scan' in k and 'FETCH_SIZE' in d:
        v = d['FETCH_SIZE']
        per_launch_kb = sum(v) / len(v)
        json.dump({"workload": "C3", "batch": 1024, "kernel": "bm25_scan_kernel", "launches": len(v),
                   "FETCH_SIZE_KB_per_launch": per_launch_kb,
                   "hbm_bytes_per_launch": per_launch_kb * 1024 * 2,
                   "note": "FETCH_SIZE (KB) x 1024 x 2: gfx950 rocprofv3 reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM)"},
                  open(sys.argv[2], "w"))
PY
echo "== knn bench =="
cd $ROOT && timeout 200 python scripts/gpu_knn_bench.py 2>&1 | tee gpurun_out/knn_bench.log
echo "== done =="
