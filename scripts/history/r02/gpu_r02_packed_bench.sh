#!/bin/bash
# packed postings as a separately reported configuration: C3 on both routes, C2; FETCH_SIZE of the packed MaxScore / scan kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out/r02
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['achieved'], r['frac'], r['other_scorer_ms_per_step'], r['host_plan_ms_per_step'], d['config'].get('device_bytes_per_gpu'))" "$1"; }
timeout 300 python bench.py --no-cpu-baseline --packed | tee gpurun_out/r02/bench_c3_packed.json | show c3_packed
timeout 300 python bench.py --no-cpu-baseline | tee gpurun_out/r02/bench_c3_plain.json | show c3_plain
timeout 300 python bench.py --no-cpu-baseline --packed --no-prune | tee gpurun_out/r02/bench_c3_packed_noprune.json | show c3_packed_noprune
timeout 300 python bench.py --no-cpu-baseline --no-prune | tee gpurun_out/r02/bench_c3_plain_noprune.json | show c3_plain_noprune
timeout 300 python bench.py --no-cpu-baseline --workload C2 --packed | tee gpurun_out/r02/bench_c2_packed.json | show c2_packed
cd /tmp
for mode in "--packed" "--packed --no-prune"; do
  rm -rf /tmp/pmcf
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcf -o p --output-format csv -- python $ROOT/bench.py --no-cpu-baseline $mode --steps 5 --warmup 2 > /tmp/pmcf.log 2>&1
  f=$(find /tmp/pmcf -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$mode" <<'PY' | tee -a $ROOT/gpurun_out/r02/packed_pmc_fetch.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r['Counter_Name'] == 'FETCH_SIZE': agg[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'maxscore_kernel' in k or 'scan_kernel' in k: print(sys.argv[2], k, len(v), round(sum(v[-5:]) / len(v[-5:]), 1), 'KB per launch (last 5)')
PY
done
