#!/bin/bash
# Round 3, call AI: the N-rank glue of bench.py on the closing build: 2 ranks on one GPU (gloo, collectives staged through the host), the
# bound exchange on by default (the real shared-memory table, two processes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
timeout 200 python bench.py --no-cpu-baseline --gpus 2 --debug-same-gpu --steps 40 --warmup 4 2>$O/bench_ai_same2.err | tee $O/bench_ai_same2.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('same2', d['n_gpus'], d['value'], d['ms_per_step'], d['config']['sharding'][:160])"
tail -2 $O/bench_ai_same2.err | cut -c1-200
echo "== done =="
