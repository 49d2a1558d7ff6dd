#!/bin/bash
# round 4 baseline at the start of a session: whole GPU suite, c3 / forced-dist / loopback bench lines
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04a}
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest.log
for WL in c3 c5; do
timeout 600 python bench.py --workload $WL --steps 10 --warmup 2 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
echo "bench $WL rc=$?"; tail -c 700 $OUT/bench_$WL.json
timeout 600 python bench.py --workload $WL --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_${WL}_forcedist.json 2> $OUT/bench_${WL}_forcedist.err
echo "bench $WL forcedist rc=$?"; tail -c 900 $OUT/bench_${WL}_forcedist.json
done
BT_MGPU_SELF_LOOPBACK=1 timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_c3_forcedist_loopback.json 2> $OUT/bench_c3_forcedist_loopback.err
echo "bench c3 forcedist loopback rc=$?"; tail -c 900 $OUT/bench_c3_forcedist_loopback.json
