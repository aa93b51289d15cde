#!/bin/bash
# round 8: scan kernel v9 (owner-lane collect, parts without barriers) in two workgroup shapes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
V="${1:-0:0:1024,0:1:1024,0:1792:1024}"
echo "== pytest gpu (default 12x1024) =="
timeout 500 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -4
echo "== sweep default 12x1024 =="
timeout 600 python scripts/gpu_sweep.py --docs 10000000 --queries 2048 --steps 10 --oracle-queries 3 --variants "$V" 2>&1 | cut -c1-1200
echo "== pytest gpu (alt 16x768) =="
NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_16x768.so timeout 500 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -4
echo "== sweep alt 16x768 =="
NRTGPU_LIB_PATH=$PWD/nrtsearch_amd/libnrtgpu_16x768.so timeout 600 python scripts/gpu_sweep.py --docs 10000000 --queries 2048 --steps 10 --oracle-queries 3 --variants "$V" 2>&1 | cut -c1-1200
echo "== done =="
