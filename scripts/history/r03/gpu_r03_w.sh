#!/bin/bash
# Round 3, call W: lazy sketch build: vector / fuzz / exchange / hybrid / sizes tests, C4 line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_vectors_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_sizes_gpu.py tests/test_abi.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_w.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_w.log | tail -12 | cut -c1-400
timeout 200 python bench.py --workload C4 --knn-queries 32 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c4', d['value'], d['ms_per_step'], r['kernel'], r['frac'], d.get('verify',{}).get('agrees_with_fp64'), d['config']['corpus_build_s'])"
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== done =="
