#!/bin/bash
# round 5, first session after GPU access comes back: what was committed unverified, then the
# walk experiments, A/B
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r05f; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_mgpu_identity.py tests/test_gpu_mgpu.py tests/test_gpu_cabi.py tests/test_bench_contract.py tests/test_gpu_debug.py -q > $OUT/pytest_new.log 2>&1
echo "pytest new rc=$?"; grep -E "passed|failed|Error" $OUT/pytest_new.log | tail -5
for two in 0 1; do
  echo "== BT_WALK_TWO_PASS=$two"
  bash tools/stage_times.sh "c4 c4 c3 c5" BT_WALK_TWO_PASS=$two 2>&1 | cut -c1-420
done
BT_WALK_TWO_PASS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q > $OUT/pytest_two_pass.log 2>&1
echo "pytest two-pass rc=$?"; grep -E "passed|failed" $OUT/pytest_two_pass.log | tail -2
