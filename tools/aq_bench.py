#!/usr/bin/env python
"""Times peer lists / area query / leaves-to-balls / space invader on one GPU.

    python tools/aq_bench.py [--n 10000000] [--nballs 10000000] [--radius 1e-3]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10**7)
    ap.add_argument("--nballs", type=int, default=10**7)
    ap.add_argument("--radius", type=float, default=1e-3)
    ap.add_argument("--mpb", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from boxtree_amd import (AreaQueryBuilder, HIPArrayContext, LeavesToBallsLookupBuilder,
                             PeerListFinder, SpaceInvaderQueryBuilder, TreeBuilder)
    actx = HIPArrayContext(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    f64 = torch.float64
    pts = [torch.rand(args.n, generator=g, dtype=f64, device="cuda") for _ in range(3)]
    bc = [torch.rand(args.nballs, generator=g, dtype=f64, device="cuda") for _ in range(3)]
    br = torch.full((args.nballs,), args.radius, dtype=f64, device="cuda")
    tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=args.mpb)

    def timed(f):
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return r, best * 1e3

    (pl, _), t_pl = timed(lambda: PeerListFinder(actx)(actx, tree))
    (aq, _), t_aq = timed(lambda: AreaQueryBuilder(actx)(actx, tree, bc, br, peer_lists=pl))
    (lbl, _), t_lbl = timed(lambda: LeavesToBallsLookupBuilder(actx)(actx, tree, bc, br,
                                                                      peer_lists=pl))
    (_, _), t_si = timed(lambda: SpaceInvaderQueryBuilder(actx)(actx, tree, bc, br,
                                                                peer_lists=pl))
    print(json.dumps({
        "n": args.n, "nballs": args.nballs, "radius": args.radius, "nboxes": int(tree.nboxes),
        "nlevels": int(tree.nlevels), "peer_entries": int(pl.peer_lists.shape[0]),
        "aq_entries": int(aq.leaves_near_ball_lists.shape[0]),
        "ms": {"peer_lists": t_pl, "area_query": t_aq, "leaves_to_balls(incl. aq)": t_lbl,
               "space_invader": t_si},
        "balls_per_s_area_query": args.nballs / (t_aq * 1e-3),
    }))


if __name__ == "__main__":
    main()
