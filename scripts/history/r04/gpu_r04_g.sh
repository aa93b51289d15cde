#!/bin/bash
# Round 4, call 6: the suite with the new shapes (several masks per query, vector dimensions that are no multiple of 16), twice;
# a driver-form bench line (traffic from the build-stamped PMC record).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/g; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'exh', (r.get('exhaustive') or {}).get('frac'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
el "new tests"
timeout 300 python -m pytest "tests/test_filters_gpu.py::test_several_filter_and_must_not_clauses" "tests/test_vectors_gpu.py::test_knn_dimensions_that_are_no_multiple_of_16" tests/test_vectors_gpu.py tests/test_filters_gpu.py tests/test_hybrid_gpu.py -m gpu -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest_new.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_new.log | tail -30 | cut -c1-220
for rep in 1 2; do
  el "suite $rep"
  timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite_$rep.log 2>&1
  echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_suite_$rep.log | tail -4 | cut -c1-200
done
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "bench (driver's form)"
for rep in 1 2; do timeout 250 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_steps20_$rep.json | show "c3 --steps 20 --warmup 5"; done
el "done"
