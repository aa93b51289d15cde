#!/bin/bash
# Round 5, call f: why the survivor queue was slower (instrumented kernels, event counts per query: the queued build of call e
# against the plain walk); item-wide scattered window order on the iid / clustered / sorted corpora; the GPU suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/f; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
P="bits@256,cells|150"
el "event counts: the queued build (call e's development library)"
NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_queue_dev.so timeout 300 python scripts/gpu_look_policy.py --profile --steps 8 --oracle-queries 0 --configs "$P|NRTGPU_MS_QUEUE=0,NRTGPU_MS_SCATTER=0;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=0;$P|NRTGPU_MS_QUEUE=1,NRTGPU_MS_SCATTER=0,NRTGPU_MS_QUEUE_FLUSH_WINS=1000000" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/queue_event_counts.log | cut -c1-900
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
for v in iid clustered sorted; do
  el "C3 $v: scatter x speculation"
  timeout 600 python scripts/gpu_look_policy.py --variant $v --oracle-queries 6 --configs "$P|NRTGPU_MS_SCATTER=0,SPEC=5;$P|NRTGPU_MS_SCATTER=1,SPEC=5;$P|NRTGPU_MS_SCATTER=1,SPEC=0;$P|NRTGPU_MS_SCATTER=1,SPEC=5" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/scatter_$v.log | cut -c1-420
done
unset NRTGPU_LIB_PATH
el "GPU suite"
timeout 600 python -m pytest tests -m gpu -q --maxfail=8 --tb=short --durations=6 -p no:cacheprovider > $O/pytest_suite.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -16 | cut -c1-220
el "done"
