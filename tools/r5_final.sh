#!/bin/bash
# Measurement round of round 5: tools/r5_final.sh TAG [notests]
#   -m gpu suite + smoke, default bench line (with the CPU baseline), bench + rocprofv3 kernel
#   stats + timeline per workload, the N > 1 path on one rank (RCCL world 1; c3, c5, c4 and c3
#   with the self-loopback switch) with its timeline, two real rank processes on the box's GPU,
#   FETCH_SIZE / WRITE_SIZE passes over c3 plain and through the N > 1 path (all kernels) and
#   over the keys-only sort alone.  Everything lands in gpurun_out/TAG; copy what is to be judged
#   to profiles/.
set -u
TAG=$1; MODE=${2:-full}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
if [ "$MODE" = full ]; then
  timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -1
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
for WL in c3 c2 c4 c3c c1 c5; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fr_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 6 --warmup 2 --cpu-sample 0 > /tmp/fr_$WL.log 2>&1)
  DB=$(find /tmp/fr_$WL -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; python tools/timeline_gaps.py $DB bbox_ 4 --kernels > $OUT/${WL}_timeline.txt 2>&1; fi
  CSV=$(find /tmp/fr_$WL -name '*kernel_stats.csv' | head -1)
  if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi
done
for WL in c3 c5 c4; do
  timeout 600 python bench.py --workload $WL --force-dist --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_${WL}_forcedist.json 2> $OUT/bench_${WL}_forcedist.err
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/frd_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --force-dist --steps 6 --warmup 2 --cpu-sample 0 > /tmp/frd_$WL.log 2>&1)
  DB=$(find /tmp/frd_$WL -name '*.db' | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_forcedist_kernel_stats.csv; python tools/timeline_gaps.py $DB bbox_ 4 --all > $OUT/${WL}_forcedist_timeline.txt 2>&1; fi
  CSV=$(find /tmp/frd_$WL -name '*kernel_stats.csv' | head -1)
  if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_forcedist_kernel_stats.csv; fi
done
BT_MGPU_SELF_LOOPBACK=1 timeout 600 python bench.py --force-dist --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_c3_forcedist_loopback.json 2> $OUT/bench_c3_forcedist_loopback.err
timeout 600 python bench.py --gpus 2 --n 20000000 --steps 3 --warmup 1 > $OUT/bench_2ranks_sharedgpu.json 2> $OUT/bench_2ranks.err; echo "2-rank bench rc=$?"
python tools/bench_brief.py $OUT/bench_*.json
for WL in c3 c3dist; do
  EXTRA=""; [ $WL = c3dist ] && EXTRA="--force-dist"
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmck_${WL}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c3 $EXTRA --steps 2 --warmup 1 --cpu-sample 0 --rng torch > /tmp/pmck_${WL}_$C.log 2>&1)
    DB=$(find /tmp/pmck_${WL}_$C -name '*.db' | head -1)
    if [ -z "$DB" ]; then echo "$C: no db"; tail -3 /tmp/pmck_${WL}_$C.log; continue; fi
    python tools/pmc_summary.py $DB $OUT/pmc_${C}_${WL}.csv
  done
  KS=$OUT/c3_kernel_stats.csv; [ $WL = c3dist ] && KS=$OUT/c3_forcedist_kernel_stats.csv
  python tools/pmc_traffic.py $WL 100000000 $OUT/pmc_FETCH_SIZE_${WL}.csv $OUT/pmc_WRITE_SIZE_${WL}.csv $KS $OUT/pmc_traffic_${WL}.json
done
# c4: traffic of all kernels, L2 hits / misses and the VALU counters of the walks
for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmck_c4_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c4 --steps 2 --warmup 1 --cpu-sample 0 > /tmp/pmck_c4_$C.log 2>&1)
  DB=$(find /tmp/pmck_c4_$C -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "c4 $C: no db"; continue; fi
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_c4.csv
done
python tools/pmc_traffic.py c4 110000000 $OUT/pmc_FETCH_SIZE_c4.csv $OUT/pmc_WRITE_SIZE_c4.csv $OUT/c4_kernel_stats.csv $OUT/pmc_traffic_c4.json
python tools/route_bench.py > $OUT/route_bench.txt 2>&1; tail -1 $OUT/route_bench.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcs_$C -o p -- python $GRAFT_REPO_ROOT/tools/sort_bench.py keys 100000000 2 36 > /tmp/pmcs_$C.log 2>&1)
  DB=$(find /tmp/pmcs_$C -name '*.db' | head -1)
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_sortbench_keys.csv
  grep -i "onesweep_keys\|keys_hist" $OUT/pmc_${C}_sortbench_keys.csv | head -3
done
python tools/sort_bench.py keys 100000000 3 36 | tee $OUT/sort_bench_keys.txt
python tools/pmc_sort_json.py $OUT/pmc_FETCH_SIZE_sortbench_keys.csv $OUT/pmc_WRITE_SIZE_sortbench_keys.csv 100000000 $OUT/pmc_onesweep_keys.json 2>&1 | tail -2
