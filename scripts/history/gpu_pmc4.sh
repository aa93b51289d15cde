set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for V in "0:0:1024" "0:2:1024"; do
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pmcx -o p --output-format csv -- python $ROOT/scripts/gpu_sweep.py --docs 10000000 --queries 1024 --steps 2 --oracle-queries 0 --variants "$V" > /tmp/pmcx.log 2>&1
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  python - "$f" "$V" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    agg[r['Kernel_Name'][:30]][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'scan' in k: print(sys.argv[2], dict(d))
PY
  rm -rf /tmp/pmcx
done
