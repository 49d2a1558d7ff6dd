#!/bin/bash
# kernel stats of one workload under rocprofv3: tools/kstats.sh TAG WORKLOAD [ENV=VAL ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; WL=$2; shift 2
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$TAG -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 1 --cpu-sample 0 > /tmp/ks_$TAG.log 2>&1)
DB=$(find /tmp/ks_$TAG -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; fi
CSV=$(find /tmp/ks_$TAG -name '*kernel_stats.csv' | head -1)
if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi
python tools/prof_summary.py $OUT/${WL}_kernel_stats.csv | head -${KSTATS_LINES:-30}
