#!/bin/bash
# Round 3, the round's measurement set -> gpurun_out/r03/final (copied into profiles/r03_* afterwards): bench lines (C3 default with closed
# loop + CPU baseline, exhaustive A/B, packed, C2, C4 at 1 / 32 / 64 queries, one rank's share of 2 / 4 / 8 GPUs), query shapes, rocprofv3
# kernel stats of the default bench, FETCH_SIZE and the SQ counters of the scorers in passes of their own (kernel trace only).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03/final; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=110
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], d['value'], d['ms_per_step'], d.get('p50_latency_ms'), r['kernel'], r['avg_launch_ms'], r['frac'], r.get('effective_frac'), r.get('host_plan_ms_per_step'), c.get('host_cpus_busy'), c.get('dist_stage_ms'), (d.get('cpu_baseline') or {}).get('value'), d.get('closed_loop'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
timeout 120 python bench.py 2>$O/bench.err | tee $O/bench_line.json | show c3
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --no-prune --steps 60 2>/dev/null | tee $O/bench_line_no_prune.json | show c3_noprune
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --packed 2>/dev/null | tee $O/bench_c3_packed.json | show c3_packed
timeout 60 python bench.py --no-cpu-baseline --closed-loop "" --workload C2 2>/dev/null | tee $O/bench_c2.json | show c2
for w in "8 0" "8 7" "4 0" "4 3" "2 0" "2 1"; do set -- $w
  timeout 60 python bench.py --no-cpu-baseline --force-dist --emulate-world $1 --emulate-rank $2 2>/dev/null | tee $O/bench_emu$1_r$2.json | show emu$1_r$2
done
timeout 60 python bench.py --no-cpu-baseline --force-dist --emulate-world 8 --exchange-mode allgather 2>/dev/null | tee $O/bench_emu8_r0_allgather.json | show emu8_r0_allgather
for q in 1 32 64; do
  timeout 200 python bench.py --workload C4 --knn-queries $q --steps 10 --warmup 3 $([ $q != 32 ] && echo "--no-cpu-baseline --no-verify") 2>/dev/null | tee $O/bench_c4_q$q.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4_q$q', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_frac'], d.get('verify'))" || echo "c4_q$q FAILED"
done
timeout 60 python bench.py --workload C4 --emulate-world 8 --knn-queries 32 --steps 20 --warmup 3 --no-cpu-baseline --no-verify 2>/dev/null | tee $O/bench_c4_emu8.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4_emu8', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['mfma_frac'])"
timeout 200 python scripts/gpu_query_shapes.py 2>&1 | grep -v amdgpu.ids | tee $O/query_shapes.log | cut -c1-260
timeout 100 python scripts/gpu_sweep.py --oracle-queries 4 --variants 0:1792:1024 2>&1 | grep -v amdgpu.ids > $O/maxscore_sweep.log; grep -c MISMATCH $O/maxscore_sweep.log
cd /tmp
B="python $ROOT/bench.py --no-cpu-baseline --closed-loop '' --warmup 2"
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r03 --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --warmup 2 --steps 20 > /tmp/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} $O/r03_kernel_stats.csv \;
head -5 $O/r03_kernel_stats.csv | cut -c1-60,200-420
pmc() {  # name, counters..., then "--" and extra bench flags
  n=$1; shift; cs=""; while [ "$1" != "--" ]; do cs="$cs $1"; shift; done; shift
  rm -rf /tmp/pmc1; timeout 200 rocprofv3 --kernel-trace --pmc $cs -d /tmp/pmc1 -o p --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --warmup 1 --steps 4 --host-threads 1 "$@" > /tmp/pmc1.log 2>&1
  f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$n" <<'PY' | tee -a $O/r03_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'bm25' in k:
        print(sys.argv[2], k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()}, '(launches, mean per launch)')
PY
}
rm -f $O/r03_pmc.txt
pmc fetch_default FETCH_SIZE GRBM_GUI_ACTIVE --
pmc fetch_noprune FETCH_SIZE GRBM_GUI_ACTIVE -- --no-prune
pmc sq1_default SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --
pmc sq2_default SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --
cd $ROOT
echo "== done =="
