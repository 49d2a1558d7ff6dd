#!/bin/bash
# c2/c3 kernel timelines (idle gaps of one step) + the Python layer's profile at 10^7 points.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-tl}; mkdir -p $OUT
for WL in c2 c3; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/tl_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 4 --warmup 2 --cpu-sample 0 > /tmp/tl_$WL.log 2>&1)
  DB=$(find /tmp/tl_$WL -name '*.db' | head -1)
  python tools/timeline_gaps.py $DB bbox_ 4 > $OUT/gaps_$WL.txt 2>&1
  head -70 $OUT/gaps_$WL.txt
done
timeout 300 python tools/py_overhead.py 1e7 > $OUT/py_overhead.txt 2>&1; head -50 $OUT/py_overhead.txt
