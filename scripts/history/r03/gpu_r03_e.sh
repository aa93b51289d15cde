#!/bin/bash
# Round 3, call E: event counts of the MaxScore walk (experiment build): dense / sparse lookup rounds, search steps, test-and-set rounds.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r03; mkdir -p $O
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_cnt.so
for W in 1 8; do
timeout 300 python scripts/gpu_sweep.py --world $W --oracle-queries 0 --variants 0:1792:1024 2>&1 | grep -v amdgpu.ids | tee $O/sweep_cnt_w$W.log | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    if d.get('event') != 'variant': continue
    p = d.get('maxscore_profile_per_query')
    print('W', $W, 'ms', d['maxscore_ms'])
    print(json.dumps({'windows': p['windows'], 'chunks': p['chunks'], 'streamed': p['postings_streamed'], 'surviving': p['postings_surviving'], 'lookups': p['lookups'], 'cands': p['candidates'], 'compactions': p['compactions'],
      'tas_rounds': p['waves_idle_cycles'], 'dense_rounds': p['waves_part_prologue_cycles'], 'sparse_rounds': p['waves_meeting_cycles'], 'search_steps': p['waves_walk_cycles'], 'cand_rounds': p['prologue_cycles']}))
"
done
