#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('host_plan_ms_per_step'), d['config']['planner_threads'])" "$1"; }
for t in 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --host-threads $t | show c3_t$t
  timeout 300 python bench.py --no-cpu-baseline --workload C2 --host-threads $t | show c2_t$t
done
NRTGPU_PLAN_TRACE=1 timeout 300 python bench.py --no-cpu-baseline --workload C2 --steps 6 --warmup 5 2>&1 >/dev/null | grep -E "nrtgpu (plan|call)" | tail -6
