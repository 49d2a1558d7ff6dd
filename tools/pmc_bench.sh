#!/bin/bash
# per-kernel PMC averages of one bench run: bash tools/pmc_bench.sh TAG WORKLOAD COUNTER [COUNTER..]
TAG=$1; WL=$2; shift 2
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for C in "$@"; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcb_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --cpu-sample 0 > /tmp/pmcb_$C.log 2>&1)
  DB=$(find /tmp/pmcb_$C -name '*.db' | head -1)
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_${WL}.csv
  head -14 $OUT/pmc_${C}_${WL}.csv
done
