#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_debug.py tests/test_gpu_fmm.py tests/test_gpu_distributed_fmm.py tests/test_gpu_cost.py -q > $OUT/pytest_a.log 2>&1
echo "pytest a rc=$?"; grep -n "Error\|passed\|failed" $OUT/pytest_a.log | head -20
timeout 2400 python -m pytest tests/test_gpu_c5.py -q --durations=10 > $OUT/pytest_c5.log 2>&1
echo "pytest c5 rc=$?"; tail -22 $OUT/pytest_c5.log | cut -c1-250
