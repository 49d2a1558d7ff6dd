#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_mgpu_identity.py tests/test_gpu_cabi.py tests/test_bench_contract.py tests/test_gpu_mgpu.py -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; grep -n "Error\|passed\|failed" $OUT/pytest.log | head -20
for ids in 1 0; do
BOXTREE_HIP_BENCH_IDS=$ids timeout 600 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --force-dist > $OUT/bench_c3_forcedist_ids$ids.json 2> $OUT/bench_c3_forcedist_ids$ids.err
echo "bench c3 forcedist ids=$ids rc=$?"; python -c "
import json,sys; d=json.loads(open('$OUT/bench_c3_forcedist_ids$ids.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('sharded_impl'), d['config'].get('exchange_bytes'), d['config'].get('exchange_bytes_ids'))"
done
