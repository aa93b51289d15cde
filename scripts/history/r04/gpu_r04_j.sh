#!/bin/bash
# Round 4, call 9: the validation matrix of the closing build -- the suite as the driver runs it (twice); the BM25 files on packed
# postings and with one workgroup per item (the A/B paths must stay correct); query shapes at C3 size; the driver-form line.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/j; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'frac', r.get('frac'), 'traffic', r.get('traffic') is not None, 'exh', (r.get('exhaustive') or {}).get('frac'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
tailpy() { grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" "$1" | tail -${2:-4} | cut -c1-300; }
BM25="tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py tests/test_hybrid_gpu.py tests/test_fullsize_gpu.py"
for rep in 1 2; do
  el "suite $rep"
  timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite_$rep.log 2>&1
  echo "pytest rc=$?"; tailpy $O/pytest_suite_$rep.log 6
done
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "BM25 files, packed postings"
NRTGPU_PACKED_POSTINGS=1 timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" > $O/pytest_packed.log 2>&1
echo "pytest rc=$?"; tailpy $O/pytest_packed.log 6
el "BM25 files, one workgroup per item (NRTGPU_MS_PERSISTENT=0), merge inside the turn"
NRTGPU_MS_PERSISTENT=0 NRTGPU_TURN_BEFORE_MERGE=0 timeout 600 python -m pytest $BM25 -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider -k "not knn and not c4 and not hybrid_c5" > $O/pytest_per_item.log 2>&1
echo "pytest rc=$?"; tailpy $O/pytest_per_item.log 6
el "BM25 files, no helpers"
NRTGPU_MS_HELPERS=0 timeout 600 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_helpers0.log 2>&1
echo "pytest rc=$?"; tailpy $O/pytest_helpers0.log 4
el "query shapes"
timeout 400 python scripts/gpu_query_shapes.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tee $O/query_shapes.log | cut -c1-400
el "bench"
for rep in 1 2; do timeout 250 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20 --warmup 5"; done
el "done"
