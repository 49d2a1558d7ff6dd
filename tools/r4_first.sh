#!/bin/bash
# round 4, first GPU session: the new multi-GPU tests, the full c5 tree on one GPU, baselines
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mgpu.py -x -q > $OUT/pytest_mgpu.log 2>&1
echo "pytest mgpu rc=$?"; tail -15 $OUT/pytest_mgpu.log
timeout 300 python tools/c5_full.py --worlds 1 2 --n 2000000 --out $OUT/c5_small.json > /dev/null 2> $OUT/c5_small.err
echo "c5 small rc=$?"; tail -5 $OUT/c5_small.err
timeout 1500 python tools/c5_full.py --worlds 1 2 4 8 --write --out $OUT/c5_full.json > /dev/null 2> $OUT/c5_full.err
echo "c5 full rc=$?"; tail -8 $OUT/c5_full.err
cp tests/golden/c5_global_counts.json $OUT/ 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
echo "bench c3 rc=$?"; tail -c 600 $OUT/bench_c3.json
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_c3_forcedist.json 2> $OUT/bench_c3_forcedist.err
echo "bench c3 forcedist rc=$?"; tail -c 600 $OUT/bench_c3_forcedist.json
BT_MGPU_SELF_LOOPBACK=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_c3_forcedist_loopback.json 2> $OUT/bench_c3_forcedist_loopback.err
echo "bench c3 forcedist loopback rc=$?"; tail -c 900 $OUT/bench_c3_forcedist_loopback.json
