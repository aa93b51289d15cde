#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu =="
timeout 500 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -3
for extra in "" "--host-threads 3" "--force-dist --emulate-world 8 --steps 40 --warmup 5" "--force-dist --emulate-world 8 --steps 40 --warmup 5 --host-threads 3" "--force-dist --emulate-world 2" "--force-dist --emulate-world 4"; do
  echo "== bench $extra =="
  timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $extra 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'ms/step', d['ms_per_step'], 'p50', d['p50_latency_ms'], 'scan', r['avg_launch_ms'], 'frac', r['frac'], 'plan', r['host_plan_ms_per_step'], r.get('accumulators'))"
done
