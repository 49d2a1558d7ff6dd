#!/usr/bin/env python
"""Random in-process multi-rank runs of the sharded build + local essential tree +
per-rank traversal against the single-GPU result (tests/test_gpu_parity.py::
check_multi_rank_let).   python tools/fuzz_multi_rank.py [ncases] [first_seed] [native]

With "native" the ranks run the library's own bt_mgpu_* entries over the local
communicator (threads), every third case with separate point targets; with "extents" every
case has separate targets with radii over four decades (random stick-out factor and norm)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# BOXTREE_EMU=1: against the CPU emulation of the kernels (tests/emu/README.md) instead of a GPU
if os.environ.get("BOXTREE_EMU", "0") == "1":
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_actx
    emu_actx.install_for_tests()
from test_gpu_parity import check_multi_rank_let  # noqa: E402

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
native = len(sys.argv) > 3 and sys.argv[3] in ("native", "extents")
extents = len(sys.argv) > 3 and sys.argv[3] == "extents"
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
    if extents:
        scale = {"uniform": 0.05, "sphere": 0.1, "clustered": 0.3, "normal": 0.4}[kw["dist_kind"]]
        kw.update(sep_targets=True, top_level=max(kw["top_level"], 2),
                  target_extents=(scale * float(rng.choice([0.1, 0.3, 1.0, 3.0])),
                                  float(rng.choice([0.0, 0.1, 0.25, 0.5, 1.0])),
                                  str(rng.choice(["linf", "l2"]))))
    try:
        check_multi_rank_let(**kw)
    except BaseException:
        print("FAILED", seed, kw, flush=True)
        raise
print(n, "multi-rank cases ok")
