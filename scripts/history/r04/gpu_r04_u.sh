#!/bin/bash
# Round 4, call 20: four postings per lane and instruction under speculative thresholds (later-clause rounds run at 30 % lane-slot
# use): 16 waves x 4 x 32-tile windows (128 VGPRs) and 12 waves x 4 x 64, against the product's 12 x 8 x 64.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/u; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
for v in "" _s4w16 _s4w12 ""; do
  L=$ROOT/nrtsearch_amd/libnrtgpu$v.so
  NRTGPU_LIB_PATH=$L timeout 200 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py -m gpu -q --maxfail=4 --tb=line -p no:cacheprovider -k "not knn and not speculative" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -1
  NRTGPU_LIB_PATH=$L timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3$v.json | show "c3$v"
done
