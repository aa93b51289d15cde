#!/bin/bash
# Round 4, call 19: the launch parameters again under speculative thresholds (items are ~15 % shorter): spare CUs, the critical-path
# bar, helper slots.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/t; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], 'kernel', r['avg_launch_ms'])" "$1" 2>/dev/null || echo "$1 FAILED"; }
b() { n=$1; shift; env "$@" timeout 250 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 2>/dev/null | tee $O/$n.json | show "$n"; }
b default NRTGPU_X=0
b spare4 NRTGPU_MS_SPARE_CUS=4
b spare12 NRTGPU_MS_SPARE_CUS=12
b spare16 NRTGPU_MS_SPARE_CUS=16
b alpha16 NRTGPU_MS_HELP_ALPHA=16
b alpha32 NRTGPU_MS_HELP_ALPHA=32
b helpmin8 NRTGPU_MS_HELP_MIN=8
b helpmin32 NRTGPU_MS_HELP_MIN=32
b threads3 NRTGPU_X=0 
b default2 NRTGPU_X=0
