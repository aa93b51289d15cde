#!/bin/bash
# The standard measurements of a round on one MI355X box, one script instead of a script per call (scripts/history/ keeps the
# one-shot ones of rounds 2-5).  Usage, through gpurun:
#     gpurun --timeout 1500 -- 'bash scripts/gpu_measure.sh <tag> <step> [<step> ...]'
# writes under gpurun_out/<tag>/ (copy what is to be judged into profiles/).  Steps:
#   suite        the driver's GPU suite (pytest -m gpu) + smoke
#   bench        the default bench line twice in the driver's form (--steps 20 --warmup 5) and once at 200 steps
#   variants     bench lines: C2, packed postings, exhaustive route, clustered / sorted corpora
#   trace        rocprofv3 --kernel-trace --stats of the driver's command -> <tag>_kernel_stats.csv
#   pmc          rocprofv3 --pmc passes (each a run of its own): FETCH_SIZE of the default / exhaustive / packed lines, the SQ
#                counters of the MaxScore kernel, FETCH_SIZE + matrix-core counters of the C4 sketch kernel -> <tag>_pmc.txt
#   shapes       scripts/gpu_query_shapes.py (deletes, FILTER, MUST_NOT, minimumNumberShouldMatch, DisjunctionMax, hybrid tail)
#   emulate8     one rank's share of an 8-GPU C3 job (bench.py --force-dist --emulate-world 8): peers' bounds present / shard-level
#                speculation / the shard on its own;  trace8: rocprofv3 --kernel-trace --stats of the speculation line;
#                emulate24: the same for one rank of two / of four (peers' bounds present, shard-level speculation)
#   c4           the full C4 lines (1 / 32 / 64 queries per pass)
#   gather       scripts/ubench/gather_fetch under rocprofv3 --pmc FETCH_SIZE (what the counter tallies per access pattern)
#   cache        (round 6) the cache path of the MaxScore kernel, one --pmc pass per counter block: L2 hits / misses / requests,
#                fabric-side read requests by size and target, what the vector L1s ask the L2 for and how long it takes, texture
#                addresser busy / stalled cycles, L1 TLB -> <tag>_pmc_cache.txt
#   gatherruns   scripts/ubench/gather_fetch runs: what a gather instruction costs by the distinct lines its lanes ask for (16 KiB / 2 MiB / 64 MiB)
#   gathersweep  scripts/ubench/gather_fetch sweep: G lines/s of random 8-byte gathers by footprint (2 MiB ... 4 GiB), alone and
#                under --pmc (L2 hits / misses, fabric-side requests per launch, launches in the printed order)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export NRTGPU_BENCH_WATCHDOG=400
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d.get('config',{}); print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '|', r['kernel'], r['avg_launch_ms'], 'ms | frac', r.get('frac'), 'eff', r.get('effective_frac'), '| exh', (r.get('exhaustive') or {}).get('frac'), '| c4', {k: (r.get('c4') or {}).get(k) for k in ('frac', 'mfma_frac', 'queries_per_s')}, '| spec', c.get('speculation'), '| GB', round(c.get('device_bytes_per_gpu', 0) / 1e9, 2))" "$1" 2>/dev/null || echo "$1 FAILED"; }
pmc() {  # name, kernel substring, counters..., then "--" and the command
  n=$1; k=$2; shift 2; cs=""; while [ "$1" != "--" ]; do cs="$cs $1"; shift; done; shift
  rm -rf /tmp/pmc1; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $cs -d /tmp/pmc1 -o p --output-format csv -- "$@" > /tmp/pmc1.log 2>&1 )
  f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$n" "$k" <<'PY' | tee -a $O/${TAG}_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if sys.argv[3] in k:
        print(sys.argv[2], k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()}, '(launches, mean per launch)')
PY
}
python -c "from nrtsearch_amd import build; print('build_id', build.build_id())"
for step in "$@"; do case $step in
suite)
  el "GPU suite"
  timeout 700 python -m pytest tests -m gpu -q --maxfail=8 --tb=short --durations=6 -p no:cacheprovider > $O/pytest_suite.log 2>&1
  echo "pytest rc=$?"; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" $O/pytest_suite.log | tail -14 | cut -c1-220
  timeout 120 python __graft_entry__.py smoke 2>&1 | tail -1 ;;
bench)
  el "bench lines"
  for rep in 1 2; do timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/${TAG}_bench_steps20_$rep.json | show "c3 --steps 20 --warmup 5 (driver's form)"; done
  timeout 400 python bench.py --c4-steps 0 2>/dev/null | tee $O/${TAG}_bench_line.json | show "c3 default (200 steps)" ;;
variants)
  el "bench variants"
  timeout 200 python bench.py --workload C2 --no-cpu-baseline 2>/dev/null | tee $O/${TAG}_bench_c2.json | show "c2"
  timeout 200 python bench.py --packed --no-cpu-baseline --closed-loop '' --c4-steps 0 2>/dev/null | tee $O/${TAG}_bench_c3_packed.json | show "c3 packed"
  timeout 200 python bench.py --no-prune --no-cpu-baseline --closed-loop '' --c4-steps 0 2>/dev/null | tee $O/${TAG}_bench_line_no_prune.json | show "c3 no-prune"
  for v in clustered sorted; do
    timeout 300 python bench.py --corpus-variant $v --steps 100 --warmup 10 --c4-steps 0 --exhaustive-steps 0 --no-cpu-baseline --closed-loop "64" 2>/dev/null | tee $O/${TAG}_bench_c3_$v.json | show "c3 $v"
  done ;;
trace)
  el "kernel trace"
  rm -rf /tmp/prof; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --steps 20 --warmup 5 > /tmp/prof_bench.log 2>&1 )
  find /tmp/prof -name "*kernel_stats*" -exec cp {} $O/${TAG}_kernel_stats.csv \;
  head -6 $O/${TAG}_kernel_stats.csv | cut -c1-60,200-420 ;;
pmc)
  el "PMC passes"
  rm -f $O/${TAG}_pmc.txt
  pmc fetch_default bm25 FETCH_SIZE GRBM_GUI_ACTIVE -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1
  pmc fetch_noprune bm25 FETCH_SIZE GRBM_GUI_ACTIVE -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1 --no-prune
  pmc fetch_packed bm25 FETCH_SIZE GRBM_GUI_ACTIVE -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1 --packed
  pmc sq1_default bm25 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1
  pmc c4_fetch knn_sketch FETCH_SIZE GRBM_GUI_ACTIVE -- python $ROOT/bench.py --workload C4 --knn-queries 64 --no-cpu-baseline --no-verify --closed-loop "" --warmup 1 --steps 4
  pmc c4_mfma knn_sketch SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES -- python $ROOT/bench.py --workload C4 --knn-queries 64 --no-cpu-baseline --no-verify --closed-loop "" --warmup 1 --steps 4 ;;
shapes)
  el "query shapes"
  timeout 400 python scripts/gpu_query_shapes.py --steps 6 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/${TAG}_query_shapes.log | cut -c1-260 ;;
emulate8)
  el "one rank of eight"
  timeout 300 python bench.py --force-dist --emulate-world 8 --emulate-peers final --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_bench_emulate8.json | show "1 of 8, peers' bounds present"
  timeout 300 python bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_bench_emulate8_speculation.json | show "1 of 8, shard-level speculation (nothing travels)"
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('   shard speculation', d['config'].get('shard_speculation'), 'stages', d['config'].get('dist_stage_ms'))" $O/${TAG}_bench_emulate8_speculation.json
  timeout 300 python bench.py --force-dist --emulate-world 8 --shard-bounds local --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_bench_emulate8_silent.json | show "1 of 8, peers silent, the shard's own speculation only" ;;
emulate24)
  el "one rank of two / of four"
  for W in 2 4; do
    timeout 400 python bench.py --force-dist --emulate-world $W --emulate-peers final --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_bench_emulate${W}.json | show "1 of $W, peers' bounds present"
    timeout 400 python bench.py --force-dist --emulate-world $W --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 2>/dev/null | tee $O/${TAG}_bench_emulate${W}_speculation.json | show "1 of $W, shard-level speculation"
  done ;;
trace8)
  el "kernel trace, one rank of eight"
  rm -rf /tmp/prof8; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof8 -o t --output-format csv -- python $ROOT/bench.py --force-dist --emulate-world 8 --steps 100 --warmup 10 --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --submitters 1 > /tmp/prof8_bench.log 2>&1 )
  find /tmp/prof8 -name "*kernel_stats*" -exec cp {} $O/${TAG}_emulate8_kernel_stats.csv \;
  python -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])): print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us', r['Percentage'])" $O/${TAG}_emulate8_kernel_stats.csv | head -14 ;;
c4)
  el "C4 lines"
  for q in 1 32 64; do timeout 400 python bench.py --workload C4 --knn-queries $q --steps 20 --warmup 4 2>/dev/null | tee $O/${TAG}_bench_c4_q$q.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('c4 q$q', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], 'frac', r['frac'], 'mfma', r['mfma_frac'], 'traffic', r['traffic'], 'verify', (d.get('verify') or {}).get('agrees_with_fp64'))"; done ;;
cache)
  el "cache-path PMC passes"
  rm -f $O/${TAG}_pmc_cache.txt
  cpmc() { n=$1; shift; pmc $n bm25_maxscore "$@" -- python $ROOT/bench.py --no-cpu-baseline --closed-loop "" --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1 | tee -a $O/${TAG}_pmc_cache.txt; }
  cpmc tcc_hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum GRBM_GUI_ACTIVE
  cpmc tcc_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum
  cpmc tcc_stall TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum
  cpmc tcp_req TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
  cpmc tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
  cpmc tcp_stall2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum
  cpmc ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
  cpmc ta2 TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUSY_max TA_BUSY_min
  cpmc tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
  cpmc sq_mem SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY
  ;;
gatherruns)
  el "gather cost by distinct lines per instruction"
  ( cd /tmp && timeout 120 $ROOT/scripts/ubench/gather_fetch runs ) 2>&1 | grep "^runs" | tee $O/${TAG}_gather_runs.txt
  ;;
gathersweep)
  el "gather ceiling by footprint"
  ( cd /tmp && timeout 200 $ROOT/scripts/ubench/gather_fetch sweep ) 2>&1 | grep "^sweep" | tee $O/${TAG}_gather_sweep.txt
  for cs in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmcs; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $cs -d /tmp/pmcs -o g --output-format csv -- $ROOT/scripts/ubench/gather_fetch sweep > /tmp/pmcs.log 2>&1 )
    f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/${TAG}_gather_sweep.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gather_mod' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
for i, (d, c) in enumerate(sorted(by.items(), key=lambda kv: int(kv[0]))):
    print('sweep_pmc launch', i, '(footprint', i // 3, 'rep', i % 3, ')', {k: round(v) for k, v in c.items()})
PY
  done ;;
gather)
  el "gather calibration"
  rm -rf /tmp/pmcg; ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcg -o g --output-format csv -- $ROOT/scripts/ubench/gather_fetch 4 > $O/${TAG}_gather_fetch.txt 2>&1 )
  grep "requested_bytes" $O/${TAG}_gather_fetch.txt | tail -5 ;;
*) echo "unknown step $step" ;;
esac; done
el "done"
