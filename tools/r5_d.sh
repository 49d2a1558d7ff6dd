#!/bin/bash
# round 5: the N = 8 dress rehearsal on one GPU; golden counts regain the particle-order checksum
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
timeout 1500 python tools/c5_full.py --worlds 1 2 4 8 --write --out $OUT/c5_full.json > /dev/null 2> $OUT/c5_full.err
echo "c5 full rc=$?"; tail -3 $OUT/c5_full.err | cut -c1-400
cp tests/golden/c5_global_counts.json $OUT/
timeout 2400 python -m pytest tests/test_gpu_c5.py -x -q --durations=10 > $OUT/pytest_c5.log 2>&1
echo "pytest c5 rc=$?"; tail -25 $OUT/pytest_c5.log
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
