#!/bin/bash
# Round 4, call 29: queries per step (bench.py --batch): 512 / 1024 / 2048 / 4096 at C3.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/batch; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'p50', d['p50_latency_ms'], 'reruns', d.get('config',{}).get('speculation',{}).get('reruns'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for B in 1024 2048 4096 512 1024; do
  timeout 250 python bench.py --batch $B --steps $((102400 / B)) --warmup 5 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_b$B.json | show "c3 batch $B"
done
