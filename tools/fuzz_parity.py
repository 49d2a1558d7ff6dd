#!/usr/bin/env python
"""Randomised parity runs: device Tree / FMMTraversalInfo (both traversal paths) against
the CPU oracle on random configurations -- dimensions, dtype, sizes, distributions,
tree kinds, separate targets, target radii, refine weights, n-away, list-3 criteria,
user bounding boxes.  Bit-identical or the seed is reported.

    python tools/fuzz_parity.py [ncases] [first_seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_case(seed):
    rng = np.random.default_rng(seed)
    dims = int(rng.choice([1, 2, 3], p=[0.1, 0.4, 0.5]))
    dtype = np.float64 if rng.random() < 0.7 else np.float32
    n = int(rng.choice([1, 2, 7, 60, 500, 3000, 20000, 60000]))
    dist = rng.choice(["normal", "uniform", "clustered", "lattice", "duplicates"])

    def points(m, shift=0.0):
        if dist == "uniform":
            p = [rng.random(m) for _ in range(dims)]
        elif dist == "clustered":
            p = [np.where(rng.random(m) < 0.5, 0.3 + 1e-3 * rng.standard_normal(m),
                          rng.standard_normal(m)) for _ in range(dims)]
        elif dist == "lattice":
            k = max(2, int(round(m ** (1.0 / dims))))
            p = [rng.integers(0, k, m) / float(k) for _ in range(dims)]
        elif dist == "duplicates":
            base = [rng.standard_normal(max(1, m // 4)) for _ in range(dims)]
            idx = rng.integers(0, max(1, m // 4), m)
            p = [b[idx] for b in base]
        else:
            p = [rng.standard_normal(m) for _ in range(dims)]
        return [(a + shift).astype(dtype) for a in p]

    kw = {}
    particles = points(n)
    targets = None
    kind = str(rng.choice(["adaptive", "adaptive", "non-adaptive", "adaptive-level-restricted"]))
    kw["kind"] = kind
    mpb = int(rng.choice([1, 3, 10, 30, 64]))
    if dist in ("lattice", "duplicates"):
        mpb = max(mpb, 30)                      # many coincident points
    if kind == "non-adaptive":
        # upstream (and the oracle, literally) materialise every box of a complete
        # 2^d-tree down to the deepest level before pruning: keep that shallow
        mpb = max(mpb, 10)
        if dist not in ("normal", "uniform") or n > 20000:
            kind = kw["kind"] = "adaptive"
    use_weights = kind == "adaptive" and rng.random() < 0.15
    if rng.random() < 0.4:
        targets = points(int(rng.choice([1, 50, 2000, 15000])), shift=float(rng.random()))
    trav_kw = {"well_sep_is_n_away": int(rng.choice([1, 1, 2]))}
    if targets is not None and kind != "adaptive-level-restricted" and rng.random() < 0.5:
        nt = len(targets[0])
        kw["target_radii"] = (2.0 ** rng.uniform(-12, -2, nt)).astype(dtype)
        kw["stick_out_factor"] = float(rng.choice([0.0, 0.1, 0.25]))
        norm = str(rng.choice(["linf", "l2"]))
        kw["extent_norm"] = norm
        crits = ["precise_linf", "static_linf"] if norm == "linf" else ["precise_linf", "static_l2"]
        trav_kw["from_sep_smaller_crit"] = str(rng.choice(crits))
        use_weights = False
    if use_weights:
        ntot = n + (len(targets[0]) if targets is not None else 0)
        kw["refine_weights"] = rng.integers(0, 5, ntot).astype(np.int32)
        kw["max_leaf_refine_weight"] = max(int(mpb * 3), 5)
    else:
        kw["max_particles_in_box"] = mpb
    if rng.random() < 0.1 and "target_radii" not in kw:
        allp = particles if targets is None else [np.concatenate([a, b])
                                                  for a, b in zip(particles, targets)]
        lo = min(float(a.min()) for a in allp) - 0.5
        hi = max(float(a.max()) for a in allp) + 0.5
        bbox = np.empty((dims, 2), dtype)
        bbox[:, 0], bbox[:, 1] = lo, hi
        kw["bbox"] = bbox
    if kind == "adaptive" and rng.random() < 0.1 and "target_radii" not in kw:
        kw["skip_prune"] = True
        trav_kw = None                           # traversal needs a pruned tree
    return particles, targets, kw, trav_kw


def run(ncases, first_seed, verbose=True):
    from compare import assert_same_traversal, assert_same_tree
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.tree_build import MaxLevelsExceeded
    from oracle import oracle
    oracle.build_lib()
    actx = HIPArrayContext(0)
    t0 = time.time()
    stats = {"ok": 0, "max_levels": 0, "csr_limit": 0}
    for seed in range(first_seed, first_seed + ncases):
        particles, targets, kw, trav_kw = make_case(seed)
        dev = lambda arrs: None if arrs is None else [actx.from_numpy(a) for a in arrs]  # noqa: E731
        dkw = dict(kw)
        for name in ("target_radii", "refine_weights"):
            if dkw.get(name) is not None:
                dkw[name] = actx.from_numpy(dkw[name])
        try:
            otree = oracle.build_tree(particles, targets=targets, **kw)
            oerr = None
        except oracle.MaxLevelsExceeded as e:
            oerr = e
        try:
            tree, _ = TreeBuilder(actx)(actx, dev(particles), targets=dev(targets), **dkw)
            derr = None
        except MaxLevelsExceeded as e:
            derr = e
        except Exception:
            print(f"EXCEPTION at seed {seed}: dims={len(particles)} n={len(particles[0])} "
                  f"kw={ {k: (v if np.ndim(v) == 0 else '...') for k, v in kw.items()} }",
                  flush=True)
            raise
        assert (oerr is None) == (derr is None), (seed, oerr, derr)
        if oerr is not None:
            stats["max_levels"] += 1
            continue
        try:
            assert_same_tree(actx.to_numpy(tree), otree)
            if trav_kw is not None:
                tkw = dict(trav_kw)
                otrav = oracle.build_traversal(otree, **tkw)
                for force_generic in (True, False):
                    trav, _ = FMMTraversalBuilder(actx, **tkw)(actx, tree,
                                                               _force_generic=force_generic)
                    assert_same_traversal(actx.to_numpy(trav), otrav)
        except AssertionError:
            print(f"MISMATCH at seed {seed}: dims={len(particles)} n={len(particles[0])} kw="
                  f"{ {k: (v if np.ndim(v) == 0 else '...') for k, v in kw.items()} } "
                  f"trav={trav_kw}", flush=True)
            raise
        stats["ok"] += 1
    if verbose:
        print(f"{ncases} cases from seed {first_seed}: {stats} in {time.time() - t0:.1f} s")
    return stats


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200,
        int(sys.argv[2]) if len(sys.argv) > 2 else 0)
