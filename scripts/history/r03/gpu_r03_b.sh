#!/bin/bash
# Round 3, call B: msm stream cut parity + timing; cycle profile of the MaxScore items on a 1/8 shard and on the whole index.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_maxscore_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_b.log
timeout 600 python scripts/gpu_query_shapes.py --skip-hybrid 2>&1 | grep -v amdgpu.ids | grep "min_should\|filter" | tee $O/query_shapes_b.log
for W in 8 1; do
  timeout 300 python scripts/gpu_sweep.py --world $W --oracle-queries 0 --variants 0:0:1024,0:1792:1024 2>&1 | grep -v amdgpu.ids | tee $O/sweep_prof_w$W.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if d.get('event') != 'variant': continue
    print('W', $W, 'flags', d['flags'], 'ms', d['maxscore_ms'], 'step', d['ms_per_step'])
    p = d.get('maxscore_profile_per_query')
    if p: print(json.dumps(p))
"
done
echo "== done =="
