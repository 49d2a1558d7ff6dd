#!/bin/bash
# HBM traffic of one onesweep digit pass from the PMC counters (two separate passes:
# FETCH_SIZE and WRITE_SIZE do not fit together), MI355X_MICROARCH.md "HBM" section.
# usage (on the GPU box): bash tools/pmc_onesweep.sh TAG [n]
TAG=$1; N=${2:-100000000}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -o p -- python $GRAFT_REPO_ROOT/tools/sort_bench.py one $N 2 > /tmp/pmc_$C.log 2>&1)
  DB=$(find /tmp/pmc_$C -name '*.db' | head -1)
  python tools/pmc_summary.py $DB $OUT/pmc_${C}_sortbench.csv
  head -8 $OUT/pmc_${C}_sortbench.csv
done
