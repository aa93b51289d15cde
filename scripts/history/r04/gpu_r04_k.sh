#!/bin/bash
# Round 4, call 10: one doc per lane once an instruction's survivors fit a row (-DNRT_MS_COLLAPSE=1; round 3: did not pay), measured
# again under the round-4 launch, and the same with every later clause's record / code requested at once (NRT_MS_ROWS_SPEC).
# A/B libraries built beside the product's (nrtsearch_amd/libnrtgpu_collapse*.so, NRTGPU_LIB_PATH); parity of each first.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/k; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'build', r.get('build_id'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
tailpy() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" "$1" | tail -${2:-4} | cut -c1-300; }
for v in "" _collapse _collapse_spec; do
  L=$ROOT/nrtsearch_amd/libnrtgpu$v.so
  el "parity libnrtgpu$v.so"
  NRTGPU_LIB_PATH=$L timeout 300 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q --maxfail=4 --tb=short -p no:cacheprovider -k "not knn" > $O/pytest$v.log 2>&1
  echo "pytest rc=$?"; tailpy $O/pytest$v.log 3
  el "bench libnrtgpu$v.so"
  NRTGPU_LIB_PATH=$L timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3$v.json | show "c3$v"
done
el "done"
