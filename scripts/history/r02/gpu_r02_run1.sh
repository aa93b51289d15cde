#!/bin/bash
# Round 2: first run of the MaxScore route -- parity tests, then C3 timing pruned / exhaustive / instrumented.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
for i in 1 2; do timeout 600 python -m pytest tests/test_maxscore_gpu.py -x -q 2>&1 | tail -25 | tee gpurun_out/r02/maxscore_tests_$i.log; done
echo "== sweep =="
timeout 400 python scripts/gpu_sweep.py --queries 2048 --steps 6 --oracle-queries 8 --variants "0:0:1024,0:16:1024,0:1792:1024,0:0:64,0:16:64,0:0:1,0:16:1" 2>&1 | tee gpurun_out/r02/sweep_maxscore.log | cut -c1-1500
