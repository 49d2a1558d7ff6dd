#!/usr/bin/env python
"""Times the stages of the N>1 path with one rank (world=1, nccl)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29513", RANK="0", WORLD_SIZE="1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
from boxtree_amd.distributed import (build_local_essential_tree, exchange_particles,
                                     gather_global_box_tree, number_sharded_tree)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**8
actx = HIPArrayContext(0)
g = torch.Generator(device="cuda")
g.manual_seed(15)
v = [torch.randn(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
nrm = torch.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
pts = [(c / nrm).contiguous() for c in v]
del v, nrm
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(3):
    t0 = T()
    p2, _, kw, st = exchange_particles(actx, dist, pts, None, {}, max_particles_in_box=64)
    t1 = T()
    tree, _ = tb(actx, p2, max_particles_in_box=64, **kw)
    t2 = T()
    num = number_sharded_tree(dist, tree, st)
    t3 = T()
    let = None
    if os.environ.get("BOXTREE_HIP_SHARDED", "let") == "gather":
        gt = gather_global_box_tree(actx, dist, tree, num)
        mask, ranges = num["target_boxes_mask"], num["active_level_ranges"]
    else:
        gt, let = build_local_essential_tree(actx, dist, tree, st, num)
        mask, ranges = let["target_boxes_mask"], let["active_level_ranges"]
    t4 = T()
    trav, _ = tg(actx, gt, _target_boxes_mask=mask, _active_level_ranges=ranges)
    t5 = T()
    if "times_ms" in st:
        print("   exchange:", "  ".join(f"{k} {v:.2f}" for k, v in st["times_ms"].items()))
    if "times_ms" in let if isinstance(let, dict) else False:
        print("   LET:", "  ".join(f"{k} {v:.2f}" for k, v in let["times_ms"].items()))
    print(f"exchange {1e3*(t1-t0):.2f}  build {1e3*(t2-t1):.2f}  number {1e3*(t3-t2):.2f}  "
          f"LET/gather {1e3*(t4-t3):.2f}  traversal {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms")
dist.destroy_process_group()
