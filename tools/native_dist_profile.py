"""Stage times of the N > 1 path through the bt_mgpu_* entries, one rank (RCCL world 1):
    python tools/native_dist_profile.py [n] [uniform]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
from boxtree_amd.distributed import native as nat
actx = HIPArrayContext(0)
class OneRank:
    get_rank = staticmethod(lambda: 0)
    get_world_size = staticmethod(lambda: 1)
comm = nat.rccl_comm(actx, OneRank)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**8
g = torch.Generator(device="cuda"); g.manual_seed(15)
if len(sys.argv) > 2 and sys.argv[2] == "uniform":
    pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
else:
    v = [torch.randn(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
    nrm = torch.sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); pts = [(c/nrm).contiguous() for c in v]; del v, nrm
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(8):
    t0=T(); p2, kw, st = nat.exchange_particles(actx, comm, pts, 64)
    t1=T(); tree,_ = tb(actx, p2, max_particles_in_box=64, **kw)
    t2=T(); num = nat.number_sharded_tree(actx, comm, tree)
    t3=T(); let, info = nat.build_local_essential_tree(actx, comm, tree, num)
    t4=T(); trav,_ = tg(actx, let, _target_boxes_mask=info["target_boxes_mask"], _active_level_ranges=info["active_level_ranges"])
    t5=T()
    if os.environ.get("STAGES"):
        print({k: round(v, 2) for k, v in tb.last_stage_times.items() if v > 0.2})
    print(f"exchange {1e3*(t1-t0):.2f} build {1e3*(t2-t1):.2f} number {1e3*(t3-t2):.2f} LET {1e3*(t4-t3):.2f} trav {1e3*(t5-t4):.2f} total {1e3*(t5-t0):.2f}")
