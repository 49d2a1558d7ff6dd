"""boxtree_amd -- MI355X-native tree build and FMM traversal generation.

Drop-in for the hot path of inducer/boxtree behind its own call surface
(boxtree/__init__.py:26-52)::

    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.traversal import FMMTraversalBuilder
    actx = HIPArrayContext()
    tree, _ = TreeBuilder(actx)(actx, particles, max_particles_in_box=30)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)

All device work runs in hand-written gfx950 HIP kernels in
``libboxtree_hip.so`` (C ABI: ``include/boxtree_hip.h``).  There is no CPU
fallback: without a HIP device or the built library, constructing an array
context raises.
"""

from boxtree_amd.area_query import (
    AreaQueryBuilder, LeavesToBallsLookupBuilder, PeerListFinder, SpaceInvaderQueryBuilder)
from boxtree_amd.array_context import HIPArrayContext
from boxtree_amd.bounding_box import BoundingBoxFinder
from boxtree_amd.traversal import BuiltList, FMMTraversalBuilder, FMMTraversalInfo
from boxtree_amd.tree import Tree, TreeOfBoxes, TreeWithLinkedPointSources, box_flags_enum
from boxtree_amd.tree_build import ExtentNorm, MaxLevelsExceeded, TreeBuilder, TreeKind

__all__ = [
    "AreaQueryBuilder", "LeavesToBallsLookupBuilder", "PeerListFinder",
    "SpaceInvaderQueryBuilder",
    "BoundingBoxFinder", "BuiltList", "FMMTraversalBuilder", "FMMTraversalInfo",
    "ExtentNorm", "HIPArrayContext", "MaxLevelsExceeded", "Tree", "TreeBuilder", "TreeKind",
    "TreeOfBoxes", "TreeWithLinkedPointSources", "box_flags_enum",
]
# (the reference's particle fixtures live in boxtree_amd.tools, as upstream's do in boxtree.tools;
# boxtree/__init__.py:26-52 does not export them either)

__version__ = "0.1"
