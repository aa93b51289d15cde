#!/bin/bash
# Round 5, call g: the GPU suite on the current build; the default bench line in the driver's form (with the C4 leg); the clustered /
# sorted corpora through bench.py.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/g; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=400
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], 'frac', r.get('frac'), 'exh', (r.get('exhaustive') or {}).get('frac'), 'c4', {k: (r.get('c4') or {}).get(k) for k in ('frac','mfma_frac','queries_per_s','avg_launch_ms')}, 'spec', d.get('config',{}).get('speculation'), 'dev_GB', round(d.get('config',{}).get('device_bytes_per_gpu',0)/1e9,2))" "$1" 2>/dev/null || echo "$1 FAILED"; }
python -c "from nrtsearch_amd import build; print('build_id', build.build_id())"
el "GPU suite"
timeout 600 python -m pytest tests -m gpu -q --maxfail=8 --tb=short --durations=5 -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -12 | cut -c1-220
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
el "bench line, driver's form"
( time timeout 400 python bench.py --steps 20 --warmup 5 2>$O/bench_steps20.err | tee $O/bench_steps20.json | show "c3 --steps 20 --warmup 5" ) 2>&1 | grep -v "^$\|user\|sys"
el "bench line, default"
( time timeout 400 python bench.py --c4-steps 0 2>/dev/null | tee $O/bench_default.json | show "c3 default (200 steps)" ) 2>&1 | grep -v "^$\|user\|sys"
for v in clustered sorted; do
  el "bench $v"
  timeout 300 python bench.py --corpus-variant $v --steps 60 --warmup 10 --c4-steps 0 --exhaustive-steps 0 --no-cpu-baseline --closed-loop "64" 2>/dev/null | tee $O/bench_$v.json | show "c3 $v"
done
el "done"
