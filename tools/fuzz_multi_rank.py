#!/usr/bin/env python
"""Random in-process multi-rank runs of the sharded build + local essential tree +
per-rank traversal against the single-GPU result (tests/test_gpu_parity.py::
check_multi_rank_let).   python tools/fuzz_multi_rank.py [ncases] [first_seed] [native]

With "native" the ranks run the library's own bt_mgpu_* entries over the local
communicator (threads), every third case with separate point targets."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import check_multi_rank_let  # noqa: E402

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
native = len(sys.argv) > 3 and sys.argv[3] == "native"
for seed in range(first, first + n):
    rng = np.random.default_rng(90000 + seed)
    dims = int(rng.choice([2, 3]))
    kw = dict(dims=dims, world=int(rng.integers(2, 9)),
              dist_kind=str(rng.choice(["sphere", "uniform", "normal", "clustered"])),
              nway=int(rng.choice([1, 1, 2])), n_per=int(rng.choice([500, 3000, 20000, 50000])),
              mpb=int(rng.choice([4, 8, 30, 64])),
              top_level=int(rng.integers(1, 5) if dims == 3 else rng.integers(2, 6)),
              seed=int(rng.integers(0, 10**6)), expect_partial=False)
    if native:
        kw.update(native=True, sep_targets=seed % 3 == 0)
    try:
        check_multi_rank_let(**kw)
    except BaseException:
        print("FAILED", seed, kw, flush=True)
        raise
print(n, "multi-rank cases ok")
