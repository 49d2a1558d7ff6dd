#!/bin/bash
# parity subset first, then bench lines for the given workloads (stage times included)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-chk}; mkdir -p $OUT; shift
PYT=$1; shift
if [ -n "$PYT" ]; then
  timeout 2400 python -m pytest $PYT -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
  grep -E "passed|failed|error" $OUT/pytest.log | tail -5
fi
for WL in "$@"; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$WL.json").read().strip().splitlines()[-1])
print("$WL", "%.4g"%d["value"], "%.3f ms"%d["ms_per_step"], "roofline %.3f"%d["roofline"]["frac"])
print("   ", {k: round(v,2) for k,v in d["stages_ms"].items() if k.startswith("trav")})
PY
done
