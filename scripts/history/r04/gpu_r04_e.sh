#!/bin/bash
# Round 4, call 4: the persistent launch with the launch record read from memory (libnrtgpu.so) against kernel arguments
# (libnrtgpu_kargs.so: the build call 3 measured), CUs left to the next batch's plan expansion (NRTGPU_MS_SPARE_CUS), alpha.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04/e; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(sys.argv[1], d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('effective_frac'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
bc3() { env "$@" timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0; }
KA=$ROOT/nrtsearch_amd/libnrtgpu_kargs.so
el "parity (BM25 files, launch record build)"
timeout 420 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_filters_gpu.py tests/test_baseline_sizes_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py tests/test_exchange_gpu.py -m gpu -q --maxfail=6 --tb=short -p no:cacheprovider > $O/pytest_bm25.log 2>&1
echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr" $O/pytest_bm25.log | tail -6 | cut -c1-300
el "bench A/B: builds x spare CUs"
for sp in 0 4 8 16; do
  bc3 NRTGPU_MS_SPARE_CUS=$sp 2>/dev/null | tee $O/c3_rec_spare$sp.json | show "c3 record  SPARE=$sp"
  bc3 NRTGPU_MS_SPARE_CUS=$sp NRTGPU_LIB_PATH=$KA 2>/dev/null | tee $O/c3_kargs_spare$sp.json | show "c3 kargs   SPARE=$sp"
done
bc3 NRTGPU_MS_SPARE_CUS=2 2>/dev/null | tee $O/c3_rec_spare2.json | show "c3 record  SPARE=2"
bc3 NRTGPU_MS_SPARE_CUS=32 2>/dev/null | tee $O/c3_rec_spare32.json | show "c3 record  SPARE=32"
el "alpha at spare 8"
for a in 0 16 24 32 48; do bc3 NRTGPU_MS_SPARE_CUS=8 NRTGPU_MS_HELP_ALPHA=$a 2>/dev/null | tee $O/c3_rec_spare8_alpha$a.json | show "c3 record SPARE=8 ALPHA=$a"; done
bc3 NRTGPU_MS_SPARE_CUS=8 NRTGPU_MS_PERSISTENT=0 2>/dev/null | tee $O/c3_rec_per_item.json | show "c3 record PERSISTENT=0"
el "--steps 20 (driver's form), spare 0 / 8"
for sp in 0 8; do for rep in 1 2; do NRTGPU_MS_SPARE_CUS=$sp timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 2>/dev/null | tee $O/c3_steps20_spare${sp}_$rep.json | show "c3 --steps 20 SPARE=$sp"; done; done
el "other workloads, spare 0 / 8"
for sp in 0 8; do
  NRTGPU_MS_SPARE_CUS=$sp timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --workload C2 2>/dev/null | tee $O/c2_spare$sp.json | show "c2 SPARE=$sp"
  NRTGPU_MS_SPARE_CUS=$sp timeout 100 python bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --force-dist --emulate-world 8 --emulate-peers final 2>/dev/null | tee $O/emu8_spare$sp.json | show "emu8 SPARE=$sp"
  NRTGPU_MS_SPARE_CUS=$sp timeout 150 python bench.py --no-cpu-baseline --exhaustive-steps 0 --steps 50 2>/dev/null | tee $O/c3_closed_spare$sp.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('closed loop SPARE=$sp', d.get('closed_loop'))"
done
el "makespan (record build)"
NRTGPU_MS_SPARE_CUS=8 timeout 200 python scripts/gpu_makespan.py --batches 2 --cus 248 2>/dev/null | grep -v last_batch | tee $O/makespan_spare8.log | cut -c1-1500
el "done"
