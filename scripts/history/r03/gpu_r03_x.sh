#!/bin/bash
# Round 3, call X: nrtgpu_knn_exact_coalesced: tests, C4 line with the closed loop (64 / 512 single-query callers).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_vectors_gpu.py tests/test_abi.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_x.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_x.log | tail -12 | cut -c1-400
timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 40 --warmup 3 --closed-loop "1,8,64,512" 2>$O/bench_x.err | tee $O/bench_x_c4_q32.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c4', d['value'], d['ms_per_step'], r['frac'], d.get('verify',{}).get('agrees_with_fp64'), d.get('closed_loop'))"
tail -3 $O/bench_x.err | cut -c1-300
echo "== done =="
