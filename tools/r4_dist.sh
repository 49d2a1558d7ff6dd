#!/bin/bash
# the N > 1 code path on one GPU: its tests, the forced-dist timeline and bench lines.  usage: r4_dist.sh TAG
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_mgpu.py tests/test_gpu_cabi.py tests/test_gpu_c5.py -x -q -k "not all_chunks" > $OUT/pytest_mgpu.log 2>&1; echo "mgpu rc=$?"; grep -E "passed|failed|error" $OUT/pytest_mgpu.log | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "partition or native or multi_rank or sharded or exchange" > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed|error" $OUT/pytest_parity.log | tail -3
bash tools/timeline_dist.sh $1 c3 --all > /dev/null; sed -n 1,3p $OUT/timeline_dist_c3.txt
for WL in c3 c5; do
python bench.py --workload $WL --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_${WL}_forcedist.json 2>$OUT/bench_${WL}_forcedist.err
done
python tools/bench_brief.py $OUT/bench_*_forcedist.json
