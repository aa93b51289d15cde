#!/bin/bash
# Round 3, call A: parity suite on the count-then-prune / shapes-on-MaxScore build, bench line, query shapes, per-rank shares.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['kernel'], r['avg_launch_ms'], r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), c.get('dist_stage_ms'), c.get('segments_per_gpu'))" "$1"; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_a.log
timeout 300 python bench.py --steps 60 --warmup 5 2>$O/bench_a.err | tee $O/bench_a.json | show c3
timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-prune 2>/dev/null | tee $O/bench_a_noprune.json | show c3_noprune
timeout 600 python scripts/gpu_query_shapes.py --skip-hybrid 2>&1 | grep -v amdgpu.ids | tee $O/query_shapes_a.log
for r in 0 7; do
  timeout 300 python bench.py --steps 60 --no-cpu-baseline --force-dist --emulate-world 8 --emulate-rank $r 2>/dev/null | tee $O/bench_a_emu8_r$r.json | show emu8_r$r
done
timeout 300 python bench.py --steps 60 --no-cpu-baseline --force-dist --emulate-world 8 --shard-layout per_shard 2>/dev/null | tee $O/bench_a_emu8_pershard.json | show emu8_pershard
echo "== done =="
