#!/usr/bin/env python
"""Random configurations of the particle-identity check (tests/test_gpu_mgpu_identity.py): rank
counts 2-14, 2D / 3D, point particles / separate targets / targets with extents, sizes, leaf
sizes, distributions, stick-out factors and norms.  usage: fuzz_identity.py SECONDS [SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# BOXTREE_EMU=1: against the CPU emulation of the kernels (tests/emu/README.md) instead of a GPU
if os.environ.get("BOXTREE_EMU", "0") == "1":
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_actx
    emu_actx.install_for_tests()
import numpy as np  # noqa: E402

from test_gpu_mgpu_identity import check_identity  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
done = 0
while time.time() - t0 < budget:
    rng = np.random.default_rng(seed0 * 100003 + done)
    cfg = dict(dims=int(rng.choice([2, 3])), world=int(rng.integers(2, 15)),
               dist_kind=str(rng.choice(["uniform", "normal", "blob"])),
               mode=str(rng.choice(["points", "targets", "extents"])),
               n_src=int(rng.choice([600, 5000, 24000, 60000])), n_tgt=int(rng.choice([300, 5000, 20000])),
               mpb=int(rng.choice([4, 20, 64])), seed=int(rng.integers(0, 10**6)),
               sof=float(rng.choice([0.0, 0.25, 0.5])), norm=str(rng.choice(["linf", "l2"])))
    try:
        check_identity(**cfg)
    except BaseException as e:      # noqa: BLE001
        print("FAILED", cfg, repr(e)[:2000], flush=True)
        sys.exit(1)
    done += 1
print(f"fuzz_identity: {done} configurations in {time.time() - t0:.0f} s, all identical", flush=True)
