#!/bin/bash
# round 6: freq nibbles behind the records (plan.h: kLookBits) against the code column, same box, development library
# (NRTGPU_LOOK_NIBS=0/1), interleaved; then the fabric-side counters of both.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06b}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_BENCH_WATCHDOG=400
bench() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], 'ms | GB', round(c.get('device_bytes_per_gpu', 0) / 1e9, 3), '| spec', c.get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for rep in 1 2 3; do
  for nibs in 0 1; do
    NRTGPU_LOOK_NIBS=$nibs bench 2>/dev/null | tee $O/${TAG}_nibs${nibs}_$rep.json | show "nibs=$nibs rep $rep"
  done
done | tee $O/${TAG}_nibs_ab.log
[ "${2:-}" = "nopmc" ] && exit 0
for nibs in 0 1; do
  rm -rf /tmp/pmcn; ( cd /tmp && NRTGPU_LOOK_NIBS=$nibs timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d /tmp/pmcn -o p --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --warmup 1 --steps 4 --host-threads 1 > /tmp/pmcn.log 2>&1 )
  f=$(find /tmp/pmcn -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "nibs=$nibs" <<'PY' | tee -a $O/${TAG}_nibs_ab.log
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'bm25_maxscore' in k:
        print(sys.argv[2], k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
PY
done
