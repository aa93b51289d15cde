#!/bin/bash
# Round 5, call d: the survivor queue of the MaxScore walk -- parity with it (product library) and without (development library,
# NRTGPU_MS_QUEUE=0), then kernel time with / without / by flush policy on C3, same box, next to round 4's library.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/d; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
CORE="tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py"
el "core suites with the queue (product library)"
timeout 400 python -m pytest $CORE -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider > $O/pytest_queue.log 2>&1; echo "rc=$?"; tail -6 $O/pytest_queue.log | cut -c1-300
el "BASELINE sizes with the queue"
timeout 400 python -m pytest tests/test_baseline_sizes_gpu.py -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider -k "bm25 or hybrid" > $O/pytest_sizes.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_sizes.log | cut -c1-300
el "core suites without the queue (development library)"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so NRTGPU_MS_QUEUE=0 timeout 400 python -m pytest $CORE -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider > $O/pytest_noqueue.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_noqueue.log | cut -c1-300
el "timing: round 4's library"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_r04dev.so timeout 300 python scripts/gpu_look_policy.py --configs "r04|0" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/timing_r04lib.log | cut -c1-330
el "timing: this library"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so timeout 600 python scripts/gpu_look_policy.py --configs "bits:2048|100000|NRTGPU_MS_QUEUE=0;bits:2048|100000|NRTGPU_MS_QUEUE=1;bits@256,cells|150|NRTGPU_MS_QUEUE=0;bits@256,cells|150|NRTGPU_MS_QUEUE=1;bits@256,cells|150|NRTGPU_MS_QUEUE=1,NRTGPU_MS_QUEUE_FLUSH_WINS=0;bits@256,cells|150|NRTGPU_MS_QUEUE=1,NRTGPU_MS_QUEUE_FLUSH_WINS=2;bits@256,cells|150|NRTGPU_MS_QUEUE=1,NRTGPU_MS_QUEUE_FLUSH_WINS=4;bits@256,cells|150|NRTGPU_MS_QUEUE=1,NRTGPU_MS_QUEUE_FLUSH_WINS=1000000;bits@256,cells|150|NRTGPU_MS_QUEUE=0" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/timing_queue.log | cut -c1-380
el "done"
