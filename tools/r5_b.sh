#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_mgpu_identity.py -q > $OUT/pytest_identity.log 2>&1
echo "pytest identity rc=$?"; grep -n "AssertionError\|passed\|failed" $OUT/pytest_identity.log | head -20
