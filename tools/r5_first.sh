#!/bin/bash
# round 5, first GPU session: particle identity (bt_mgpu_route), the refactored partition ranking
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_mgpu_identity.py -x -q > $OUT/pytest_identity.log 2>&1
echo "pytest identity rc=$?"; tail -30 $OUT/pytest_identity.log
timeout 1500 python -m pytest tests/test_gpu_mgpu.py tests/test_gpu_mgpu_extents.py -x -q > $OUT/pytest_mgpu.log 2>&1
echo "pytest mgpu rc=$?"; tail -8 $OUT/pytest_mgpu.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "multi_rank or partition" > $OUT/pytest_multi.log 2>&1
echo "pytest multi rc=$?"; tail -8 $OUT/pytest_multi.log
timeout 600 python tools/route_bench.py > $OUT/route_bench.txt 2>&1
echo "route bench rc=$?"; tail -3 $OUT/route_bench.txt
timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_c3_forcedist.json 2> $OUT/bench_c3_forcedist.err
echo "bench c3 forcedist rc=$?"; tail -c 1200 $OUT/bench_c3_forcedist.json
