#!/usr/bin/env python
"""Quick throughput sweep over dimensions / dtypes / distributions (one GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder

actx = HIPArrayContext(0)
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
g = torch.Generator(device="cuda")


def run(name, pts, mpb=64, reps=3, **kw):
    best = None
    tree = trav = None
    try:
        return _run(name, pts, mpb, reps, **kw)
    except (NotImplementedError, RuntimeError) as e:
        print(f"{name:42s} n={len(pts[0]):.1e} -> {type(e).__name__}: {str(e)[:110]}")


def _run(name, pts, mpb=64, reps=3, **kw):
    best = None
    for r in range(reps + 2):           # two warm-up steps: the scratch pool's first allocations
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tree, _ = tb(actx, pts, max_particles_in_box=mpb, **kw)
        if os.environ.get("SWEEP_DEBUG"):
            print("   tree", tree.nboxes, tree.nlevels, flush=True)
        trav, _ = tg(actx, tree)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if r >= 2:
            best = dt if best is None else min(best, dt)
    n = len(pts[0])
    print(f"{name:42s} n={n:.1e} boxes={tree.nboxes:9d} levels={tree.nlevels:3d} "
          f"{1e3 * best:8.2f} ms  {n / best / 1e9:6.2f} Gparticles/s")
    del tree, trav


for dims, dtype, n in [(2, torch.float64, 10**8), (3, torch.float32, 10**8), (2, torch.float32, 10**8),
                       (3, torch.float64, 10**6), (3, torch.float64, 2 * 10**8), (1, torch.float64, 10**7)]:
    g.manual_seed(1)
    pts = [torch.rand(n, generator=g, dtype=dtype, device="cuda") for _ in range(dims)]
    run(f"uniform {dims}D {str(dtype)[6:]}", pts)
    del pts
g.manual_seed(2)
n = 5 * 10**7
pts = [torch.randn(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
run("normal 3D float64", pts)
run("normal 3D float64 level-restricted", pts, kind="adaptive-level-restricted", reps=1)
run("normal 3D float64 mpb=8", pts, mpb=8, reps=1)
