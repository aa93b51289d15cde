#!/bin/bash
# Round 4, call 18: when a workgroup's speculative estimates are due (first after F windows begun, then every time their number has
# grown by G / 16).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/s; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'], 'reruns', d.get('config',{}).get('speculation',{}).get('reruns'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for FG in "24 32" "12 32" "13 24" "24 24" "36 32" "24 48" "13 20" "24 32"; do
  set -- $FG
  NRTGPU_MS_SPEC_FIRST=$1 NRTGPU_MS_SPEC_GROW=$2 timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/c3_f$1_g$2.json | show "c3 first=$1 grow=$2/16"
done
