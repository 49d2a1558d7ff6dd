#!/usr/bin/env python
"""Per-shape step times (min / median of several steps after warm-up) for shapes bench.py has no name for."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder

actx = HIPArrayContext(0)
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
g = torch.Generator(device="cuda")
which = sys.argv[1:] or ["u2d", "u3f", "n3d", "u3d5"]


def run(name, pts, mpb=64, reps=6, warm=2, **kw):
    ts = []
    for r in range(warm + reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tree, _ = tb(actx, pts, max_particles_in_box=mpb, **kw)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trav, _ = tg(actx, tree)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if r >= warm:
            ts.append((t2 - t0, t1 - t0, t2 - t1))
        nb, nl = tree.nboxes, tree.nlevels
        del tree, trav
    ts.sort()
    n = len(pts[0])
    best, med = ts[0], ts[len(ts) // 2]
    print(f"{name:28s} n={n:.1e} boxes={nb:9d} levels={nl:3d} min {1e3 * best[0]:7.2f} ms "
          f"(tree {1e3 * best[1]:6.2f} trav {1e3 * best[2]:6.2f})  median {1e3 * med[0]:7.2f} ms  "
          f"{n / best[0] / 1e9:5.2f} G/s", flush=True)


shapes = {
    "u2d": (2, torch.float64, 10**8, "rand"),
    "u3f": (3, torch.float32, 10**8, "rand"),
    "n3d": (3, torch.float64, 5 * 10**7, "randn"),
    "u3d5": (3, torch.float64, 5 * 10**7, "rand"),
    "u2d7": (2, torch.float64, 10**7, "rand"),
    "n2d": (2, torch.float64, 5 * 10**7, "randn"),
}
for k in which:
    dims, dtype, n, dist = shapes[k]
    g.manual_seed(1)
    fn = torch.rand if dist == "rand" else torch.randn
    pts = [fn(n, generator=g, dtype=dtype, device="cuda") for _ in range(dims)]
    run(f"{dist} {dims}D {str(dtype)[6:]}", pts)
    del pts
