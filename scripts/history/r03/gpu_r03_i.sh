#!/bin/bash
# Round 3, call I: the whole GPU suite on the current build; C4 at the BASELINE size with the fp64 verification.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=200
timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_i.log 2>&1; echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_i.log | tail -12 | cut -c1-400
timeout 300 python bench.py --workload C4 --knn-queries 32 --steps 10 --warmup 3 2>$O/bench_i_c4.err | tee $O/bench_i_c4_q32.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4_q32', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_frac'], d.get('verify'), (d.get('cpu_baseline') or {}).get('agrees_with_device'), (d.get('cpu_baseline') or {}).get('device_hits_checked'))" || tail -5 $O/bench_i_c4.err
echo "== done =="
