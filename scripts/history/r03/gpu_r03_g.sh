#!/bin/bash
# debug: where does the emulated-rank bench hang?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=25
E="--no-cpu-baseline --force-dist --emulate-world 8 --steps 8 --warmup 2"
for v in "" "--exchange-mode allgather"; do
  echo "== [$v]"
  timeout 40 python bench.py $E $v > $O/dbg_g.out 2> $O/dbg_g.err; echo "rc=$?"; tail -c 300 $O/dbg_g.out; grep -v amdgpu.ids $O/dbg_g.err | grep -A6 "Thread\|File" | head -60
done
