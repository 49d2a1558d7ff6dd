#!/bin/bash
# PMC counters of the kernels matching a pattern: tools/pmc_kernel.sh TAG WORKLOAD PATTERN COUNTER...
# (one rocprofv3 --pmc pass per counter, max value per dispatch = the biggest launch)
TAG=$1; WL=$2; PAT=$3; shift 3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for C in "$@"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmck_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 > /tmp/pmck_$C.log 2>&1)
  DB=$(find /tmp/pmck_$C -name '*.db' | head -1)
  if [ -z "$DB" ]; then echo "$C: no db"; tail -3 /tmp/pmck_$C.log; continue; fi
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_${WL}.csv
  grep -i "$PAT" $OUT/pmc_${C}_${WL}.csv | head -6
done
