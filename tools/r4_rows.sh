#!/bin/bash
# one family of colleague rows (source flag in the entry) against two: parity in both forms, bench lines
set -u
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04rows}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest_one.log 2>&1; echo "one family rc=$?"; tail -n 2 $OUT/pytest_one.log
BT_ROW_FAMILIES=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x > $OUT/pytest_two.log 2>&1; echo "two families rc=$?"; tail -n 2 $OUT/pytest_two.log
for WL in c3 c4 c5; do
for REP in 1 2; do
for FAM in 2 1; do
BT_ROW_FAMILIES=$FAM timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_${WL}_fam${FAM}_$REP.json 2> $OUT/bench_${WL}_fam${FAM}_$REP.err
done; done; done
python tools/bench_brief.py $OUT/bench_*.json | cut -c1-150 | tee $OUT/brief.txt
