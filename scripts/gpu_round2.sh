#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 600 python -m pytest tests -m gpu -q --timeout 200 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
echo "== sweep 1M =="
timeout 300 python scripts/gpu_sweep.py --docs 1000000 --queries 1024 --steps 3 --variants "0:0:256,0:1:256" 2>&1 | tee gpurun_out/sweep_1m.log | tail -12
echo "== sweep 10M =="
timeout 600 python scripts/gpu_sweep.py --docs 10000000 --queries 2048 --steps 4 --variants "$1" 2>&1 | tee gpurun_out/sweep_10m.log | tail -20
echo "== done =="
