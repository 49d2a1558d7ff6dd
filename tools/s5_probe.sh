cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/s5; mkdir -p $OUT
run() { WL=$1; shift; for kv in "$@"; do export "$kv"; done
  timeout 300 python bench.py --workload $WL --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('$WL $*', '%.3f ms' % d['ms_per_step'], ' '.join('%s=%.2f' % (k.replace('trav:','t:'), v) for k, v in s.items() if v > 0.05 and not k.startswith('trav')))"
  for kv in "$@"; do unset "${kv%%=*}"; done; }
timeout 900 python -m pytest tests -q -m gpu -x -k "parity or level_restricted or golden or host_side" > $OUT/pytest_k3.log 2>&1; grep -n "passed\|failed" $OUT/pytest_k3.log; grep -n "Error\|assert" $OUT/pytest_k3.log | head -5
run c3
run c5
run c4
run c2
