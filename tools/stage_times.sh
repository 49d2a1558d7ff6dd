#!/bin/bash
# Stage times of bench.py workloads on the GPU box, one line each:
#   bash tools/stage_times.sh "c3 c5 c4" [ENV=VAL ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
WLS=$1; shift
for kv in "$@"; do export "$kv"; done
for WL in $WLS; do
  timeout 300 python bench.py --workload $WL --steps 6 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('$WL $*', '%.3f ms' % d['ms_per_step'], ' '.join('%s=%.2f' % (k.replace('trav:', 't:'), v) for k, v in s.items() if v > 0.05))"
done
