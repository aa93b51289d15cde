#!/bin/bash
# the driver's GPU tier (pytest -m gpu, smoke) + the BM25 suite on the packed layout
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r02/pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error|Error|mean hits" gpurun_out/r02/pytest_gpu.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
NRTGPU_PACKED_POSTINGS=1 timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_vectors_gpu.py > gpurun_out/r02/pytest_suite_packed.log 2>&1
echo "pytest suite (packed) rc=$?"; grep -E "passed|failed|error|Error" gpurun_out/r02/pytest_suite_packed.log | tail -4
