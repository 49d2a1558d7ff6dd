#!/bin/bash
# Round-3 GPU session: tools/r3_session.sh TAG [full|quick]
#   -m gpu suite, default bench line, 2 real ranks on the box's GPU(s), kernel stats of
#   c3 / c4, FETCH_SIZE + WRITE_SIZE passes over c3 and c4 (all kernels).
set -u
TAG=$1; MODE=${2:-full}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$MODE" = full ]; then
  timeout 1800 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
fi
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"; tail -c 400 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
timeout 600 python bench.py --gpus 2 --n 20000000 --steps 3 --warmup 1 > $OUT/bench_2ranks_sharedgpu.json 2> $OUT/bench_2ranks.err; echo "2-rank bench rc=$?"; tail -c 1500 $OUT/bench_2ranks_sharedgpu.json; tail -5 $OUT/bench_2ranks.err
for WL in c3 c4; do
  bash tools/kstats.sh $TAG $WL > $OUT/kstats_$WL.txt 2>&1; head -34 $OUT/kstats_$WL.txt
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmck_${WL}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 --rng torch > /tmp/pmck_${WL}_$C.log 2>&1)
    DB=$(find /tmp/pmck_${WL}_$C -name '*.db' | head -1)
    if [ -z "$DB" ]; then echo "$C: no db"; tail -3 /tmp/pmck_${WL}_$C.log; continue; fi
    python tools/pmc_summary.py $DB $OUT/pmc_${C}_${WL}.csv
    head -12 $OUT/pmc_${C}_${WL}.csv
  done
done
