#!/bin/bash
# quick GPU check: host trace at 10^7, c2/c3 bench lines, parity subset
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-chk}; mkdir -p $OUT
BT_HOST_TRACE=1 timeout 300 python tools/host_trace.py 1e7 2> $OUT/trace_1e7.txt
python tools/host_trace_fmt.py $OUT/trace_1e7.txt | grep -v "trav:l\|trav:c\|trav:w\|keygen\|leaves\|ids \|gather\|boxinfo\|extents\|srcscan"
for WL in c2 c3; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$WL.json").read().strip().splitlines()[-1])
print("$WL", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
if [ -n "${2:-}" ]; then
  timeout 2400 python -m pytest $2 -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|error" $OUT/pytest.log | tail -5
fi
