#!/bin/bash
# round 6: one rank of eight (bench.py --force-dist --emulate-world 8, shard-level speculation) by the size of the doc windows and the
# window count of a workgroup's first estimate -- an eighth of C3 is 19 windows of 65536 docs per query: the first estimate (default: at
# 24 windows begun) is never due.  Development library (NRTGPU_MS_FINE_ITEMS=1: every query; NRTGPU_MS_FINE_SHIFT; NRTGPU_MS_SPEC_FIRST).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06w}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], 'ms | shard spec', c.get('shard_speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_$name.json | show "$name"
}
for rep in 1 2; do
  run default_$rep X=1
  run fine2_$rep NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=2
  run fine2_first12_$rep NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=2 NRTGPU_MS_SPEC_FIRST=12
  run fine1_first12_$rep NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=1 NRTGPU_MS_SPEC_FIRST=12
  run fine3_first24_$rep NRTGPU_MS_FINE_ITEMS=1 NRTGPU_MS_FINE_SHIFT=3 NRTGPU_MS_SPEC_FIRST=24
  run first14_$rep NRTGPU_MS_SPEC_FIRST=14
done | tee $O/${TAG}_d8_windows.log
