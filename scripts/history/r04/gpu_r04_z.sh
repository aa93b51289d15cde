#!/bin/bash
# Round 4, call 27: a part's clause records built once per workgroup (MsSmem.pc_*) against every wave building its own
# (-DNRT_MS_PART_CACHE=0, libnrtgpu_nopc.so): parity of both, the C3 line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/z; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'reruns', d.get('config',{}).get('speculation',{}).get('reruns'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for v in "" _nopc "" _nopc; do
  L=$ROOT/nrtsearch_amd/libnrtgpu$v.so
  NRTGPU_LIB_PATH=$L timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3$v.json | show "c3$v"
done
NRT_FUZZ_ROUNDS=32 timeout 400 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_packed_gpu.py tests/test_hybrid_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "not knn" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -4
timeout 300 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -3
