import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
n = int(float(sys.argv[1]))
actx = HIPArrayContext(0)
g = torch.Generator(device="cuda"); g.manual_seed(15)
v = [torch.randn(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
nrm = torch.sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); pts = [(c/nrm).contiguous() for c in v]; del v, nrm
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tree, _ = tb(actx, pts, max_particles_in_box=64)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    trav, _ = tg(actx, tree)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    free, total = torch.cuda.mem_get_info()
    s = torch.cuda.memory_stats()
    print(f"it{it} build {1e3*(t1-t0):.1f} trav {1e3*(t2-t1):.1f} ms; device used {(total-free)/2**30:.1f} GiB of {total/2**30:.0f}; "
          f"torch reserved {s['reserved_bytes.all.current']/2**30:.1f} allocated {s['allocated_bytes.all.current']/2**30:.1f}; "
          f"keygen stage {dict(tb.last_stage_times).get('keygen'):.1f}", flush=True)
    del tree, trav
