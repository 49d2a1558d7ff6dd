#!/usr/bin/env python
"""A few mid-size random cases (3*10^5 .. 6*10^6 points) against the oracle: the code
paths that only larger inputs reach (multi-tile scans, grouped inverse permutation,
partial presort + fallback, block-level leaf sorts)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# BOXTREE_EMU=1: against the CPU emulation of the kernels (tests/emu/README.md) instead of a GPU
if os.environ.get("BOXTREE_EMU", "0") == "1":
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import emu_actx
    emu_actx.install_for_tests()
from compare import assert_same_traversal, assert_same_tree  # noqa: E402

from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build_lib()
actx = HIPArrayContext(0)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for seed in range(first, first + ncases):
    rng = np.random.default_rng(31000 + seed)
    dims = int(rng.choice([2, 3]))
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    n = int(rng.choice([3 * 10**5, 10**6, 4500000, 6 * 10**6]))
    dist = str(rng.choice(["uniform", "normal", "sphere", "clustered"]))
    if dist == "uniform":
        p = [rng.random(n) for _ in range(dims)]
    elif dist == "sphere":
        v = rng.standard_normal((dims, n))
        v /= np.sqrt((v * v).sum(axis=0))
        p = [np.ascontiguousarray(v[i]) for i in range(dims)]
    elif dist == "clustered":
        p = [np.where(rng.random(n) < 0.5, 0.3 + 1e-3 * rng.standard_normal(n),
                      rng.standard_normal(n)) for _ in range(dims)]
    else:
        p = [rng.standard_normal(n) for _ in range(dims)]
    p = [a.astype(dtype) for a in p]
    kw = dict(max_particles_in_box=int(rng.choice([16, 64, 200])),
              kind=str(rng.choice(["adaptive", "adaptive", "adaptive-level-restricted"])))
    targets = None
    if rng.random() < 0.3:
        nt = n // 10
        targets = [rng.standard_normal(nt).astype(dtype) for _ in range(dims)]
        if kw["kind"] == "adaptive" and rng.random() < 0.6:
            kw.update(target_radii=(2.0 ** rng.uniform(-14, -5, nt)).astype(dtype),
                      stick_out_factor=0.25)
    t0 = time.time()
    otree = oracle.build_tree(p, targets=targets, **kw)
    otrav = oracle.build_traversal(otree)
    t1 = time.time()
    dkw = dict(kw)
    if "target_radii" in dkw:
        dkw["target_radii"] = actx.from_numpy(dkw["target_radii"])
    from boxtree_amd.tree_build import MaxLevelsExceeded
    try:
        tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p],
                                    targets=None if targets is None else
                                    [actx.from_numpy(a) for a in targets], **dkw)
    except MaxLevelsExceeded:
        # documented deviation: the 64-bit key addresses 21 levels in 3-D / 31 in 2-D
        # (19 / 29 with extents); deeper trees raise instead of differing
        key_levels = (21 if dims == 3 else 31) - (2 if "target_radii" in kw else 0)
        assert otree.nlevels - 1 > key_levels, (otree.nlevels, key_levels)
        print(f"seed {seed}: oracle tree has {otree.nlevels} levels, beyond the key: "
              "MaxLevelsExceeded", flush=True)
        continue
    assert_same_tree(actx.to_numpy(tree), otree)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    assert_same_traversal(actx.to_numpy(trav), otrav)
    print(f"seed {seed}: {dims}D {np.dtype(dtype).name} n={n} {dist} "
          f"{ {k: v for k, v in kw.items() if np.ndim(v) == 0} } targets={targets is not None} "
          f"boxes={otree.nboxes} levels={otree.nlevels}: identical (oracle {t1 - t0:.1f} s)",
          flush=True)
print(ncases, "mid-size cases ok")
