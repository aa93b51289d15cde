#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['avg_launch_ms'], r.get('other_scorer_ms_per_step'), r.get('host_plan_ms_per_step'), d['config']['scan_items_per_step'])" "$1"; }
timeout 300 python bench.py --no-cpu-baseline | show c3
timeout 300 python bench.py --no-cpu-baseline --workload C2 | show c2
timeout 300 python bench.py --no-cpu-baseline --workload C2 --packed | show c2_packed
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | show emu8
timeout 300 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 --host-threads 3 2>/dev/null | show emu8_t3
NRTGPU_PACKED_POSTINGS=1 timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_vectors_gpu.py 2>&1 | grep -E "passed|failed" | tail -2
