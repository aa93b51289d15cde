#!/bin/bash
# Round 4, call 16: speculative thresholds (NRTGPU_MS_SPEC_Z): the C3 line at several margins, the re-run counters, parity.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/q; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
flt() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -${1:-3}; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'spec', d.get('config',{}).get('speculation'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for Z in 0 6 4 3 0; do
  NRTGPU_MS_SPEC_Z=$Z timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_z$Z.json | show "c3 z=$Z"
done
NRTGPU_MS_SPEC_Z=4 timeout 300 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | flt 4
NRTGPU_MS_SPEC_Z=4 NRT_FUZZ_ROUNDS=24 timeout 600 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "not knn" 2>&1 | flt 12
