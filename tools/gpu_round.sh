#!/bin/bash
# One GPU session: new config tests, then bench + rocprof kernel stats per workload.
# usage: tools/gpu_round.sh TAG "pytest args" "workloads..."
set -u
TAG=$1; PYT=$2; shift 2
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$PYT" ]; then
  timeout 1500 python -m pytest $PYT -x -q > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
for WL in "$@"; do
  timeout 600 python bench.py --workload $WL --steps 5 --warmup 2 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  echo "bench $WL rc=$?"; tail -c 1500 $OUT/bench_$WL.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 1 --cpu-sample 0 > /tmp/prof_$WL.log 2>&1)
  DB=$(find /tmp/prof_$WL -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; fi
  CSV=$(find /tmp/prof_$WL -name '*kernel_stats.csv' | head -1)
  if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi
  head -12 $OUT/${WL}_kernel_stats.csv
done
