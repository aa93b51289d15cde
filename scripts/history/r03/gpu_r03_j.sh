#!/bin/bash
# Round 3, call J: postings per cell of the sparse terms' lookup tables (binary-search steps per lookup) vs kernel time.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=50
for c in 8 4 2 1; do
  export NRTGPU_CELL_POSTINGS=$c
  timeout 60 python bench.py --steps 60 --no-cpu-baseline --closed-loop "" 2>/dev/null | tee $O/bench_j_cell$c.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cell', $c, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['device_bytes_per_gpu'])"
  timeout 60 python bench.py --steps 60 --no-cpu-baseline --force-dist --emulate-world 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  emu8 cell', $c, d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  timeout 60 python bench.py --steps 40 --no-cpu-baseline --no-prune --closed-loop "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  noprune cell', $c, d['roofline']['avg_launch_ms'])"
done
NRTGPU_CELL_POSTINGS=2 timeout 200 python -m pytest tests/test_maxscore_gpu.py tests/test_parity_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -2
echo "== done =="
