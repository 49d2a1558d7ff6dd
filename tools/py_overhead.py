import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
actx = HIPArrayContext(0)
g = torch.Generator(device="cuda"); g.manual_seed(15)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10**7
pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
tb, tg = TreeBuilder(actx), FMMTraversalBuilder(actx)
for _ in range(3):
    tree, _ = tb(actx, pts, max_particles_in_box=64); trav, _ = tg(actx, tree)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tree, _ = tb(actx, pts, max_particles_in_box=64)
    trav, _ = tg(actx, tree)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:4500])
