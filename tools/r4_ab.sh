#!/bin/bash
# parity subset + bench lines of the working tree: tools/r4_ab.sh TAG
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04ab}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_mgpu_extents.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -n 1
for WL in c3 c4 c5 c2; do
for REP in 1 2; do
timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_${WL}_$REP.json 2> $OUT/bench_${WL}_$REP.err
done; done
python tools/bench_brief.py $OUT/bench_*.json | cut -c1-110 | tee $OUT/brief.txt
