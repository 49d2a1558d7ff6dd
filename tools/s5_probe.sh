cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/s5; mkdir -p $OUT
run() { WL=$1; shift; for kv in "$@"; do export "$kv"; done
  timeout 300 python bench.py --workload $WL --steps 8 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('$WL $*', '%.3f ms' % d['ms_per_step'], ' '.join('%s=%.2f' % (k[5:], v) for k, v in s.items() if k.startswith('trav:') and v > 0.05))"
  for kv in "$@"; do unset "${kv%%=*}"; done; }
timeout 900 python -m pytest tests -q -m gpu -x -k "trav or list or fmm or golden or parity" > $OUT/pytest_k3.log 2>&1; grep -n "passed\|failed" $OUT/pytest_k3.log
run c5
run c2
run c4
run c3
prof() { WL=$1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_$WL -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 1 --cpu-sample 0 > /tmp/ks_$WL.log 2>&1)
DB=$(find /tmp/ks_$WL -name '*.db' | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats.csv; python tools/timeline_gaps.py $DB bbox_ 4 --kernels > $OUT/${WL}_timeline.txt 2>&1; fi
CSV=$(find /tmp/ks_$WL -name '*kernel_stats.csv' | head -1)
if [ -n "$CSV" ]; then cp $CSV $OUT/${WL}_kernel_stats.csv; fi; }
prof c5
prof c4
