#!/usr/bin/env python
"""Random configurations of the distributed constant-one FMM (in-process ranks):
rank 0 must see nsources at every target.   python tools/fuzz_dfmm.py [ncases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_distributed_fmm import test_constantone_distributed as check  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for seed in range(n):
    rng = np.random.default_rng(4000 + seed)
    dims = int(rng.choice([2, 3]))
    nsources = int(rng.choice([300, 5000, 30000]))
    ntargets = None if rng.random() < 0.4 else int(rng.choice([200, 4000, 20000]))
    extent = ntargets is not None and rng.random() < 0.5
    kw = dict(world=int(rng.integers(1, 9)), dims=dims, nsources=nsources, ntargets=ntargets,
              extent=bool(extent), allreduce=bool(rng.random() < 0.25))
    try:
        check(**kw)
    except BaseException:
        print("FAILED", seed, kw, flush=True)
        raise
print(n, "distributed FMM cases ok")
