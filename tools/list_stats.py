import torch, numpy as np, sys
sys.path.insert(0,'.')
from boxtree_amd import HIPArrayContext, TreeBuilder, FMMTraversalBuilder
import bench
actx=HIPArrayContext(0)
w=bench.make_workload(torch, torch.device('cuda',0), 'c3', None, 15)
tree,_=TreeBuilder(actx)(actx,w['particles'],max_particles_in_box=64)
trav,_=FMMTraversalBuilder(actx)(actx,tree)
tb=trav.target_boxes.long(); lev=tree.box_levels.long()[tb]
l1=torch.diff(trav.neighbor_source_boxes_starts.long())
l3=torch.zeros_like(l1)
for bl in trav.from_sep_smaller_by_level:
    l3[bl.nonempty_indices.long()]+=torch.diff(bl.starts.long())
print("ntb",len(tb),"l1 mean %.1f max %d  l3 mean %.1f max %d"%(l1.float().mean(),l1.max(),l3.float().mean(),l3.max()))
for l in range(int(lev.min()),int(lev.max())+1):
    m=lev==l
    if m.any(): print(" lev",l,"n",int(m.sum()),"l1 mean %.1f max %d l3 mean %.1f max %d"%(l1[m].float().mean(),l1[m].max(),l3[m].float().mean(),l3[m].max()))
w64=(l1+l3).float()
# per-wave max/mean imbalance (64 consecutive target boxes)
n=len(w64)//64*64
ww=w64[:n].view(-1,64)
print("wave imbalance: mean of (max/mean) = %.2f"%( (ww.max(dim=1).values/ww.mean(dim=1).clamp(min=1)).mean()))
coll=torch.diff(trav.same_level_non_well_sep_boxes_starts.long()); print("coll mean %.1f max %d"%(coll.float().mean(), coll.max()))
