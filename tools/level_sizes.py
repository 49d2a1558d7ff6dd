#!/usr/bin/env python
"""Boxes per level (all / with children) of a bench workload: tools/level_sizes.py c3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOAD_MPB, make_workload  # noqa: E402
from boxtree_amd import HIPArrayContext, TreeBuilder  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
actx = HIPArrayContext(0)
w = make_workload(torch, torch.device("cuda", 0), wl, None, 15)
tree, _ = TreeBuilder(actx)(actx, w["particles"], targets=w["targets"],
                            max_particles_in_box=WORKLOAD_MPB.get(wl, 64), **w["kw"])
ls = actx.to_numpy(tree.level_start_box_nrs)
flags = actx.to_numpy(tree.box_flags)
haschild = (flags & 12) != 0
for lev in range(len(ls) - 1):
    a, b = ls[lev], ls[lev + 1]
    print(f"level {lev:2d}: {b - a:9d} boxes, {int(haschild[a:b].sum()):9d} with children")
