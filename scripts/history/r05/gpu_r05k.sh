#!/bin/bash
# round 5, call k: consecutive batches' scorers without a turn between them (NRTGPU_OVERLAP_SCORERS, development library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r05k; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; cl=d.get('closed_loop') or {}; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| merge', r['merge_ms_per_step'], '| p50/p99', d['p50_latency_ms'], d['p99_latency_ms'], '| closed loop', {k: (v.get('queries_per_s'), v.get('p50_ms'), v.get('p99_ms')) for k, v in cl.items()} if isinstance(cl, dict) else cl)" "$1" 2>/dev/null || echo "$1 FAILED"; }
export NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_dev.so
for rep in 1 2; do
  for v in 0 1; do
    NRTGPU_OVERLAP_SCORERS=$v timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --closed-loop "64,512" --exhaustive-steps 0 --c4-steps 0 2>/dev/null | tee $O/ab_overlap${v}_$rep.json | show "overlap=$v rep $rep"
  done
done
NRTGPU_OVERLAP_SCORERS=1 timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_maxscore_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
