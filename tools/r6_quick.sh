#!/bin/bash
# A SHORT first GPU session (about 20 minutes), for a GPU that comes back late in a round:
#   tools/r6_quick.sh [TAG=r06q]
# the whole -m gpu suite with its log kept, smoke, the default bench line, one line per workload, and
# one alternating A/B of the walk variants.  The full session is tools/r6_first.sh.
set -u
TAG=${1:-r06q}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -rs --durations=10 > $OUT/pytest_full.txt 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
for WL in c2 c4 c5; do
  timeout 400 python bench.py --workload $WL --steps 8 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
python tools/bench_brief.py $OUT/bench_*.json | tee $OUT/bench_brief.txt
{
for rep in 1 2; do
  for var in "BT_WALK_G8=0" "BT_WALK_G8=1" "BT_WALK_G8=2" "BT_WALK_TWO_PASS=1"; do
    bash tools/stage_times.sh "c4 c3 c5" $var 2>&1 | cut -c1-420
  done
done
} > $OUT/ab_walk_variants.txt 2>&1
cat $OUT/ab_walk_variants.txt | cut -c1-260
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/q_c3 -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-sample 0 > /tmp/q_c3.log 2>&1)
DB=$(find /tmp/q_c3 -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/c3_kernel_stats.csv
CSV=$(find /tmp/q_c3 -name '*kernel_stats.csv' | head -1); [ -n "$CSV" ] && cp $CSV $OUT/c3_kernel_stats.csv
