#!/bin/bash
# Round 4, last call: the tree as committed -- the driver's suite, smoke, the line in the driver's form and in full (the PMC record of
# this build is in profiles/pmc_traffic.json now: roofline.traffic / frac come from it).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/final; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=150
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], 'frac', r.get('frac'), 'traffic', r.get('traffic'), 'eff', r.get('effective_frac'), 'exh', (r.get('exhaustive') or {}).get('frac'), 'build', r.get('build_id'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
python -c "from nrtsearch_amd import build; print('build_id', build.build_id())"
timeout 600 python -m pytest tests -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -4 | cut -c1-200
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 250 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/c3_steps20.json | show "c3 --steps 20 --warmup 5"
timeout 250 python bench.py 2>/dev/null | tee $O/c3_line.json | show "c3 default"
