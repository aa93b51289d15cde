#!/bin/bash
# Round 3, call C: lane-mask refactor of the MaxScore kernel, with and without the collapse to one doc per lane.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'))" "$1"; }
for v in default nocollapse; do
  if [ $v = nocollapse ]; then export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_nocollapse.so; else unset NRTGPU_LIB_PATH; fi
  timeout 900 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_c_$v.log 2>&1; echo "$v pytest rc=$?"; tail -4 $O/pytest_c_$v.log
  timeout 300 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tee $O/bench_c_$v.json | show c3_$v
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | tee $O/bench_c_emu8_$v.json | show emu8_$v
done
echo "== done =="
