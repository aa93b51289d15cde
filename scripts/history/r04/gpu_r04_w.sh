#!/bin/bash
# Round 4, call 24: quarter-size doc windows for the launch's slowest queries, chosen by the postings of their two heaviest clauses
# (NRTGPU_MS_FINE_ITEMS: one MaxScore query in that many; 0: none.  NRTGPU_MS_FINE_SHIFT: 64 >> that many sub-tiles per window).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/w; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'reruns', d.get('config',{}).get('speculation',{}).get('reruns'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
b() { n=$1; shift; env "$@" timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/$n.json | show "$n"; }
b fine_off NRTGPU_MS_FINE_ITEMS=0
b fine_64_s2 NRTGPU_MS_FINE_ITEMS=64
b fine_32_s2 NRTGPU_MS_FINE_ITEMS=32
b fine_16_s2 NRTGPU_MS_FINE_ITEMS=16
b fine_64_s3 NRTGPU_MS_FINE_ITEMS=64 NRTGPU_MS_FINE_SHIFT=3
b fine_32_s3 NRTGPU_MS_FINE_ITEMS=32 NRTGPU_MS_FINE_SHIFT=3
b fine_128_s2 NRTGPU_MS_FINE_ITEMS=128
b fine_off2 NRTGPU_MS_FINE_ITEMS=0
timeout 200 python scripts/gpu_makespan.py --batches 2 --cus 248 2>/dev/null | tee $O/makespan.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'batch' in d: print({k:d[k] for k in ['span_us','balanced_us','makespan_over_balanced','helper_sessions','owner_busy_us','owner_finish_us_deciles','running_at_tenths']})
"
timeout 300 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_packed_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "not knn" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -3
timeout 300 python scripts/gpu_parity_c3.py --queries 2048 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -3
