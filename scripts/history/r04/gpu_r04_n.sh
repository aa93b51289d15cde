#!/bin/bash
# Round 4, call 13: the record policy as built (a posting per 4096 docs, the 2048 largest terms): bench, parity under the old
# and the new thresholds, the fuzz at 40 rounds (it varies the policy per round), HBM held by the C3 index.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/n; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'index_gb', round(d.get('config',{}).get('device_bytes_per_gpu',0)/1e9,2))" "$1" 2>/dev/null || echo "$1 FAILED"; }
BM25="tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fullsize_gpu.py"
timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_default.json | show "c3 default policy"
NRTGPU_RECORD_DOCS_PER_POSTING=128 timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_128.json | show "c3 a posting per 128 docs"
timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --packed 2>/dev/null | tee $O/c3_packed.json | show "c3 packed default policy"
timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tail -2
NRTGPU_RECORD_DOCS_PER_POSTING=128 timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" 2>&1 | tail -2
NRT_FUZZ_ROUNDS=40 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k shapes 2>&1 | tail -2
NRTGPU_PACKED_POSTINGS=1 NRT_FUZZ_ROUNDS=16 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k shapes 2>&1 | tail -2
