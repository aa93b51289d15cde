#!/bin/bash
# Round 3, call D: MaxScore kernel shapes A/B -- a0: 12 waves x 8 postings per lane x 64-tile windows (the round-2 shape), a1: + collapse
# to one doc per lane; b0: 16 waves x 4 postings x 32-tile windows (128 VGPRs), b1: + collapse.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'))" "$1"; }
for v in a0 a1 b0; do
  export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_$v.so
  timeout 600 python -m pytest tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_packed_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_d_$v.log 2>&1; echo "$v pytest rc=$?"; tail -2 $O/pytest_d_$v.log
  timeout 300 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tee $O/bench_d_$v.json | show c3_$v
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | tee $O/bench_d_emu8_$v.json | show emu8_$v
done
echo "== done =="
