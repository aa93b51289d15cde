#!/bin/bash
# Round 5, call b: the lookup policy x budget matrix on C3 (scripts/gpu_look_policy.py), FETCH_SIZE calibration of the gathers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/b; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
el "policy matrix, C3"
timeout 600 python scripts/gpu_look_policy.py --configs "bits:2048|100000;nib:2048|100000;map:2048|100000;map@32,cells|150;cells|100000;|-1;map@32,nib:2048|100000;map@32,bits:2048|100000;nib:210,cells|150;nib:512,cells|100000;map@8,nib:600,cells|300" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/look_policy_c3.log | cut -c1-400
el "gather calibration"
cd /tmp; rm -rf /tmp/pmcg
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmcg -o g --output-format csv -- $ROOT/scripts/ubench/gather_fetch 4 > $O/gather_fetch_stdout.txt 2>&1
cat $O/gather_fetch_stdout.txt | grep -v "^W\|^E\|rocprof" | head -12
f=$(find /tmp/pmcg -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY' | tee $O/gather_fetch_pmc.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if r['Counter_Name'] == 'FETCH_SIZE':
        agg[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, 'FETCH_SIZE per launch (as counted, KB):', [round(x, 1) for x in v])
PY
cd $ROOT
el "done"
