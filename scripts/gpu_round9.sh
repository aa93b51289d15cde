#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
bash scripts/gpu_round8.sh "0:0:1024,0:0:2048"
for lib in "" "$PWD/nrtsearch_amd/libnrtgpu_16x768.so"; do
  echo "== bench emulate-world 8, lib=[$lib] =="
  NRTGPU_LIB_PATH=$lib timeout 400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --force-dist --emulate-world 8 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['host_plan_ms_per_step'])"
  echo "== bench default, lib=[$lib] =="
  NRTGPU_LIB_PATH=$lib timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['host_plan_ms_per_step'])"
done
