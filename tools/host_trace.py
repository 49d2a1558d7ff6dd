#!/usr/bin/env python
"""Host-side timeline of one build + traversal step (BT_HOST_TRACE=1: the library prints
CLOCK_MONOTONIC stamps at its stage marks; this script adds the Python layer's).

usage: BT_HOST_TRACE=1 python tools/host_trace.py [n] 2> trace.txt; the last step's
lines are the ones after the final "[py] step" line."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder  # noqa: E402


def stamp(name):
    sys.stderr.write(f"[py]      {name:<14s} {time.monotonic_ns() * 1e-3:.1f}\n")


actx = HIPArrayContext(0)
g = torch.Generator(device="cuda")
g.manual_seed(15)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
actx.set_stage_timing(os.environ.get("BT_STAGE_EVENTS", "0") == "1")
for _ in range(4):
    torch.cuda.synchronize()
    stamp("step")
    tree, _ = tb(actx, pts, max_particles_in_box=64)
    stamp("tree done")
    trav, _ = tg(actx, tree)
    stamp("trav done")
    torch.cuda.synchronize()
    stamp("synced")
