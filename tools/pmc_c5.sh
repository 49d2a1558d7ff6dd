cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc5; mkdir -p $OUT
WL=c5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/fr_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 6 --warmup 2 --cpu-sample 0 --rng torch > /tmp/fr_$WL.log 2>&1)
DB=$(find /tmp/fr_$WL -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; fi
CSV=$(find /tmp/fr_$WL -name '*kernel_stats.csv' | head -1)
if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmck_${WL}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 --rng torch > /tmp/pmck_${WL}_$C.log 2>&1)
  DB=$(find /tmp/pmck_${WL}_$C -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "$C: no db"; tail -3 /tmp/pmck_${WL}_$C.log; continue; fi
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_${WL}.csv
done
python tools/pmc_traffic.py $WL 125000000 $OUT/pmc_FETCH_SIZE_${WL}.csv $OUT/pmc_WRITE_SIZE_${WL}.csv $OUT/${WL}_kernel_stats.csv $OUT/pmc_traffic_${WL}.json
