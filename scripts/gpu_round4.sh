#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 600 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== sweep 10M $1 =="
timeout 900 python scripts/gpu_sweep.py --docs 10000000 --queries 2048 --steps 4 --oracle-queries ${2:-4} --variants "$1" 2>&1 | tee gpurun_out/sweep_${3:-x}.log | tail -30
echo "== done =="
