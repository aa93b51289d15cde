cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "maxscore or parity or filters or fuzz or baseline or hybrid" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids" | tail -8
bash scripts/history/r06/gpu_dispersion.sh r06g
O=$ROOT/gpurun_out/r06g
rm -rf /tmp/pmcx; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY -d /tmp/pmcx -o p --output-format csv -- python $ROOT/bench.py --no-prune --no-cpu-baseline --closed-loop "" --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1 > /tmp/pmcx.log 2>&1 )
for cs in "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS"; do
  rm -rf /tmp/pmcy; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $cs -d /tmp/pmcy -o p --output-format csv -- python $ROOT/bench.py --no-prune --no-cpu-baseline --closed-loop "" --c4-steps 0 --c2-steps 0 --c5-steps 0 --warmup 1 --steps 4 --host-threads 1 > /tmp/pmcy.log 2>&1 )
done
for d in /tmp/pmcx /tmp/pmcy; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" <<'PY' | tee -a $O/r06g_scan_sq.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'bm25_scan' in k:
        print('scan_sq', k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tee $O/r06g_bench_default.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('default line q/s', d['value'], 'ms', d['ms_per_step'], 'kernel', r['avg_launch_ms'])
print(' exhaustive', {k: r['exhaustive'].get(k) for k in ('frac','physical_frac','traffic_frac','avg_launch_ms')})
print(' c2', r.get('c2'))
print(' c5', r.get('c5'))
print(' c4', {k: (r.get('c4') or {}).get(k) for k in ('frac','queries_per_s','ms_per_pass_call')})"
