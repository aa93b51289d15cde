#!/bin/bash
# Round 5, call a: the lookup structures of the MaxScore walk's later clauses (plan.h: kLook*) -- parity under every kind, then
# the policy x budget matrix on C3 (scripts/gpu_look_policy.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd); export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05/a; mkdir -p $O
T0=$(date +%s)
el() { echo "== $1 ($(( $(date +%s) - T0 )) s)"; }
python -c "from nrtsearch_amd import build; print('build_id', build.build_id())"
CORE="tests/test_parity_gpu.py tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_packed_gpu.py tests/test_fuzz_gpu.py"
el "core suites, product library (default policy)"
timeout 400 python -m pytest $CORE -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_default.log 2>&1; echo "rc=$?"; tail -4 $O/pytest_default.log | cut -c1-300
export NRTGPU_LIB_PATH=$ROOT/nrtsearch_amd/libnrtgpu_dev.so
for pol in nib bits cells none; do
  el "suites under policy $pol (development library, unlimited budget)"
  P=$pol; [ $pol = none ] && P=""
  NRTGPU_LOOK_POLICY="$P" NRTGPU_TEST_LOOKUP_BUDGET_PCT=100000 timeout 300 python -m pytest tests/test_maxscore_gpu.py tests/test_filters_gpu.py tests/test_fuzz_gpu.py tests/test_packed_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider > $O/pytest_$pol.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_$pol.log | cut -c1-300
done
el "policy matrix, C3"
timeout 600 python scripts/gpu_look_policy.py --configs "bits:2048|100000;nib:2048|100000;map:2048|100000;map@32,cells|150;cells|100000;|-1;map@32,nib:2048|100000;map@32,bits:2048|100000;nib:210,cells|150;nib:512,cells|100000;map@8,nib:600,cells|300" 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tee $O/look_policy_c3.log | cut -c1-400
el "done"
