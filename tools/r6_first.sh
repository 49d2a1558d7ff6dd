#!/bin/bash
# Round 6, first GPU session: re-verify HEAD and re-pin every number before any kernel work.
#   tools/r6_first.sh TAG
# (a) the whole -m gpu suite with the log KEPT, smoke; (b) one bench line per workload and per
# forced-dist workload; (c) A/B of level_to_rad (HEAD = ldexp vs the division it replaced, built in
# /tmp on the box) and of the walk variants (BT_WALK_TWO_PASS, BT_WALK_G8), alternating on the same
# box; (d) the parity suites under each variant.  Kernel stats / timelines / PMC are tools/r6_prof.sh.
set -u
TAG=${1:-r06a}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
nproc > $OUT/host.txt; rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -3 >> $OUT/host.txt
timeout 2400 python -m pytest tests -q -m gpu -rs --durations=15 > $OUT/pytest_full.txt 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default bench rc=$?"
for WL in c3 c2 c4 c3c c1 c5; do
  timeout 600 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_$WL.json 2> $OUT/bench_$WL.err
done
for WL in c3 c5 c4; do
  timeout 600 python bench.py --workload $WL --force-dist --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_${WL}_forcedist.json 2> $OUT/bench_${WL}_forcedist.err
done
BT_MGPU_SELF_LOOPBACK=1 timeout 600 python bench.py --force-dist --steps 10 --warmup 3 --cpu-sample 0 > $OUT/bench_c3_forcedist_loopback.json 2> $OUT/bench_c3_forcedist_loopback.err
timeout 600 python bench.py --gpus 2 --n 20000000 --steps 3 --warmup 1 > $OUT/bench_2ranks_sharedgpu.json 2> $OUT/bench_2ranks.err; echo "2-rank bench rc=$?"
python tools/bench_brief.py $OUT/bench_*.json | tee $OUT/bench_brief.txt

# A/B level_to_rad: a second tree in /tmp with the division restored
AB=/tmp/ab_div; rm -rf $AB; mkdir -p $AB
cp -r bench.py boxtree_amd include oracle tests tools __graft_entry__.py BASELINE.json $AB/ 2>/dev/null
python - <<'EOF'
import re
p = "/tmp/ab_div/boxtree_amd/csrc/bt_geom.hpp"
s = open(p).read()
s2 = s.replace("return __builtin_ldexp(root_extent, -(level + 1));",
               "return root_extent * 1 / (double) (1ull << (level + 1));")
s2 = s2.replace("return __builtin_ldexpf(root_extent, -(level + 1));",
                "return root_extent * 1 / (float) (1ull << (level + 1));")
assert s2 != s
open(p, "w").write(s2)
EOF
(cd $AB/boxtree_amd/csrc && touch bt_geom.hpp && make -j16 > /tmp/ab_make.log 2>&1; echo "ab make rc=$?")
{
for rep in 1 2 3; do
  for side in head div; do
    D=$GRAFT_REPO_ROOT; [ $side = div ] && D=$AB
    for WL in c4 c3; do
      (cd $D && timeout 300 python bench.py --workload $WL --steps 8 --warmup 3 --cpu-sample 0 2>/dev/null) | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['stages_ms']
print('$side rep$rep $WL', '%.3f ms' % d['ms_per_step'], ' '.join('%s=%.2f' % (k.replace('trav:', 't:'), v) for k, v in s.items() if k.startswith('trav') and v > 0.05))"
    done
  done
done
} > $OUT/ab_level_to_rad.txt 2>&1
cat $OUT/ab_level_to_rad.txt | cut -c1-300
# the walk variants: default, two passes over the colleagues, 2^d lanes per item -- alternating
{
for rep in 1 2 3; do
  for var in "BT_WALK_G8=0" "BT_WALK_TWO_PASS=1" "BT_WALK_G8=1" "BT_WALK_G8=2"; do
    bash tools/stage_times.sh "c4 c3 c5" $var 2>&1 | cut -c1-420
  done
done
} > $OUT/ab_walk_variants.txt 2>&1
cat $OUT/ab_walk_variants.txt | cut -c1-300
BT_WALK_TWO_PASS=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q > $OUT/pytest_two_pass.txt 2>&1
echo "pytest two-pass rc=$?"; tail -2 $OUT/pytest_two_pass.txt
for g8 in 1 2; do
  BT_WALK_G8=$g8 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_mgpu_extents.py -q > $OUT/pytest_walk_g8_$g8.txt 2>&1
  echo "pytest walk-g8=$g8 rc=$?"; tail -2 $OUT/pytest_walk_g8_$g8.txt
done
# walk kernels alone under rocprofv3: default against 2^d lanes per item in both row layouts
for WL in c4 c3 c5; do
  for g8 in 0 1 2; do
    (cd /tmp && BT_WALK_G8=$g8 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/wg8_${WL}_$g8 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 5 --warmup 2 --cpu-sample 0 > /tmp/wg8_${WL}_$g8.log 2>&1)
    DB=$(find /tmp/wg8_${WL}_$g8 -name '*.db' | head -1)
    [ -n "$DB" ] && python tools/rocpd_stats.py $DB $OUT/${WL}_kernel_stats_walk_g8_$g8.csv && grep -i "walk13\|rows_to_csr\|l3_scatter" $OUT/${WL}_kernel_stats_walk_g8_$g8.csv | cut -c1-200
  done
done
