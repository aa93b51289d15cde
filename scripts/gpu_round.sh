#!/bin/bash
# One gpurun call: parity tests, smoke, bench variants, rocprof stats.  Everything logs to gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" ; rocminfo | grep -E "gfx|Compute Unit" | head -4
echo "== pytest gpu =="
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench default =="
timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json
tail -5 gpurun_out/bench_default.err
for variant in "--no-prefetch" "--target-items 1024" "--target-items 8192" "--batch 256" "--batch 64"; do
  name=$(echo "$variant" | tr -d ' -')
  echo "== bench $variant =="
  timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $variant 2>gpurun_out/bench_$name.err | tee gpurun_out/bench_$name.json
  tail -3 gpurun_out/bench_$name.err
done
echo "== rocprofv3 kernel stats =="
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r01 -- python "${GRAFT_REPO_ROOT:-/root/repo}/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof_bench.log 2>&1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
tail -2 /tmp/prof_bench.log
mkdir -p gpurun_out/prof && find /tmp/prof -name "*stats*" -exec cp {} gpurun_out/prof/ \; 2>/dev/null
ls -la gpurun_out/prof | head
for f in gpurun_out/prof/*kernel_stats*; do echo "--- $f"; head -12 "$f"; done 2>/dev/null
echo "== done =="
