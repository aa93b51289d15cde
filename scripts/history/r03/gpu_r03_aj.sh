#!/bin/bash
# Round 3, call AJ: sanity of the bench entry points after the last refactor (small sizes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 200 python bench.py --workload C4 --docs 400000 --knn-queries 32 --steps 5 --warmup 2 --closed-loop "8" 2>$O/aj_c4.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c4 small', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['effective_frac'], r['streamed_bytes_per_launch'], d['verify']['agrees_with_fp64'], d['cpu_baseline']['agrees_with_device'], d['closed_loop']['8'])" || tail -5 $O/aj_c4.err
timeout 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --closed-loop "" 2>$O/aj_c3.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c3', d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['effective_frac'])" || tail -5 $O/aj_c3.err
echo "== done =="
