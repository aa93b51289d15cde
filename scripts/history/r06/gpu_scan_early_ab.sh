#!/bin/bash
# round 6: the exhaustive scan (bm25_scan_kernel) with the next sub-tile's posting columns requested BEFORE the current sub-tile's
# LDS adds and swaps (-DNRT_SCAN_EARLY_LOAD) against the product order; same box, interleaved; bench.py --no-prune
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:-r06e}; O=$ROOT/gpurun_out/$TAG; mkdir -p $O
VARIANT=${2:-$ROOT/nrtsearch_amd/libnrtgpu_early.so}
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '|', r['kernel'], r['avg_launch_ms'], 'ms | frac(9B)', r.get('frac'), '| at 8 B', r.get('frac_at_8B_per_posting'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for rep in 1 2 3; do
  for lib in product variant; do
    if [ $lib = variant ]; then export NRTGPU_LIB_PATH=$VARIANT; else unset NRTGPU_LIB_PATH; fi
    timeout 300 python bench.py --no-prune --steps 12 --warmup 3 --no-cpu-baseline --closed-loop "" --c4-steps 0 2>/dev/null | tee $O/${TAG}_${lib}_$rep.json | show "$lib rep $rep"
  done
done | tee $O/${TAG}_scan_ab.log
unset NRTGPU_LIB_PATH
