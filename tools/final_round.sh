#!/bin/bash
# Final measurement round: full -m gpu suite, smoke, bench + kernel stats per workload,
# default bench line, forced N>1 path, onesweep PMC traffic.  usage: tools/final_round.sh TAG
set -u
TAG=$1
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"; tail -c 600 $OUT/bench_default.json
for WL in c3 c2 c4 c3c c1; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$WL.json").read().strip().splitlines()[-1])
print("$WL", "%.4g"%d["value"], "%.3f ms"%d["ms_per_step"], "roofline %.3f"%d["roofline"]["frac"])
PY
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fr_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 6 --warmup 2 --cpu-sample 0 > /tmp/fr_$WL.log 2>&1)
  DB=$(find /tmp/fr_$WL -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; python tools/timeline_gaps.py $DB bbox_ 4 --kernels > $OUT/${WL}_timeline.txt 2>&1; fi
  CSV=$(find /tmp/fr_$WL -name '*kernel_stats.csv' | head -1)
  if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi
  head -2 $OUT/${WL}_timeline.txt
done
timeout 600 python bench.py --workload c3 --force-dist --steps 5 --warmup 2 --cpu-sample 0 > $OUT/bench_c3_forcedist.json 2> $OUT/bench_c3_forcedist.err; echo "force-dist rc=$?"; tail -c 300 $OUT/bench_c3_forcedist.json
