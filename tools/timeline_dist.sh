#!/bin/bash
# kernel timeline of the last step through the N > 1 code path on one rank:
# tools/timeline_dist.sh TAG WORKLOAD [--all|--kernels]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT; WL=$2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tld_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --force-dist --steps 4 --warmup 2 --cpu-sample 0 > /tmp/tld_$WL.log 2>&1)
DB=$(find /tmp/tld_$WL -name '*.db' | head -1)
python tools/timeline_gaps.py $DB bbox_ 4 ${3:---all} > $OUT/timeline_dist_$WL.txt 2>&1
python tools/rocpd_stats.py $DB $OUT/dist_${WL}_kernel_stats.csv > /dev/null 2>&1
head -40 $OUT/timeline_dist_$WL.txt
