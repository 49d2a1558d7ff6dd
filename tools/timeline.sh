#!/bin/bash
# kernel timeline of the last step: tools/timeline.sh TAG WORKLOAD
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT; WL=$2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 4 --warmup 2 --cpu-sample 0 > /tmp/tl_$WL.log 2>&1)
DB=$(find /tmp/tl_$WL -name '*.db' | head -1)
python tools/timeline_gaps.py $DB bbox_ 4 --kernels $3 > $OUT/timeline_$WL.txt 2>&1
cat $OUT/timeline_$WL.txt
