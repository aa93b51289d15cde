#!/bin/bash
# packed postings: dedicated tests, then the BM25 part of the GPU suite on the packed layout
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_packed_gpu.py tests/test_fuzz_gpu.py -q -x -p no:cacheprovider > gpurun_out/r02/pytest_packed.log 2>&1
echo "pytest packed rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/r02/pytest_packed.log | tail -12
NRTGPU_PACKED_POSTINGS=1 timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_vectors_gpu.py > gpurun_out/r02/pytest_suite_packed.log 2>&1
echo "pytest suite (packed) rc=$?"; grep -E "passed|failed|error|Error|assert" gpurun_out/r02/pytest_suite_packed.log | tail -12
