"""The call surface a user of the reference finds after switching, checked
without a GPU: names exported, class relations of the output records."""

import dataclasses

import numpy as np


def test_exports_match_the_reference_package():
    """boxtree/__init__.py:26-52 exports exactly these builder / record names (the array context is
    this package's own); the particle fixtures stay in ``tools`` as upstream's do."""
    import boxtree_amd
    for name in ("AreaQueryBuilder", "LeavesToBallsLookupBuilder", "PeerListFinder", "SpaceInvaderQueryBuilder",
                 "BoundingBoxFinder", "FMMTraversalBuilder", "FMMTraversalInfo", "Tree", "TreeBuilder",
                 "TreeOfBoxes", "TreeWithLinkedPointSources", "box_flags_enum"):
        assert name in boxtree_amd.__all__ and hasattr(boxtree_amd, name)
    assert not any(n.startswith("make_") for n in boxtree_amd.__all__)
    from boxtree_amd import tools
    for name in ("make_normal_particle_array", "make_uniform_particle_array", "make_surface_particle_array"):
        assert callable(getattr(tools, name))


def test_tree_is_a_tree_of_boxes():
    """boxtree/tree.py:298 ``class Tree(TreeOfBoxes)``: a downstream ``isinstance(tree, TreeOfBoxes)``
    holds; ``bounding_box`` is a FIELD of Tree and a derived (cached) value of a TreeOfBoxes
    (tree.py:247-251, 571-574)."""
    from boxtree_amd.tree import Tree, TreeOfBoxes, TreeWithLinkedPointSources
    assert issubclass(Tree, TreeOfBoxes) and issubclass(TreeWithLinkedPointSources, Tree)
    kw = {f.name: None for f in dataclasses.fields(Tree)}
    kw.update(bounding_box=(np.zeros(2), np.ones(2)), sources=[np.zeros(3), np.zeros(3)], targets=[np.zeros(4)] * 2,
              box_flags=np.zeros(5, np.uint8), level_start_box_nrs=np.array([0, 1, 5], np.int32),
              box_child_ids=np.zeros((4, 32), np.int32))
    t = Tree(**kw)
    assert isinstance(t, TreeOfBoxes)
    assert np.array_equal(t.bounding_box[1], np.ones(2))
    assert (t.dimensions, t.nboxes, t.nlevels, t.nsources, t.ntargets, t.aligned_nboxes) == (2, 5, 2, 3, 4, 32)
    # the copy every to_numpy / freeze makes keeps the field
    assert np.array_equal(t._map_arrays(lambda v: v).bounding_box[0], np.zeros(2))
    # a pure TreeOfBoxes derives its bounding box from the root's centre
    kb = {f.name: None for f in dataclasses.fields(TreeOfBoxes)}
    kb.update(root_extent=2.0, box_centers=np.array([[1.0], [3.0]]), box_levels=np.zeros(1, np.uint8))
    lo, hi = TreeOfBoxes(**kb).bounding_box
    assert np.array_equal(lo, [0.0, 2.0]) and np.array_equal(hi, [2.0, 4.0])
