#!/bin/bash
# Round 5, call e: survivor queue (with the streaming waves polling for meetings) x scattered window order on C3; then the
# clustered and the length-sorted corpus: re-runs and kernel time with / without scattering, with / without speculation.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/e; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
CORE="tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_fullsize_gpu.py tests/test_hybrid_gpu.py"
el "core suites (product library: queue + scattered windows)"
timeout 400 python -m pytest $CORE -m gpu -q --maxfail=5 --tb=short -p no:cacheprovider > $O/pytest_core.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_core.log | cut -c1-300
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
P="bits@256,cells|150"
el "C3 iid: queue x scatter"
timeout 600 python scripts/gpu_look_policy.py --configs "$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=0;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=0;$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=1;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=1;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=1,NRTGPU_MS_QUEUE_FLUSH_WINS=0;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=1,NRTGPU_MS_QUEUE_FLUSH_WINS=1000000;$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=0" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/timing_iid.log | cut -c1-400
for v in clustered sorted; do
  el "C3 $v: scatter x speculation (queue off)"
  timeout 600 python scripts/gpu_look_policy.py --variant $v --oracle-queries 6 --configs "$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=0,SPEC=5;$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=1,SPEC=5;$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=0,SPEC=0;$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=1,SPEC=0;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=1,SPEC=5" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/timing_$v.log | cut -c1-400
done
el "done"
