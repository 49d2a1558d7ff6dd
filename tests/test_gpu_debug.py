"""``debug=True`` (boxtree/tree_build.py:1039-1052, 1087-1098, 1545-1559; traversal.py:2035-2039):
both builders run the reference's assertions on what they built (boxtree_amd/debug.py) -- on every
kind of tree, and the checks do fail on a damaged container."""

import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def cloud(n, dims, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(n) for _ in range(dims)]


@pytest.mark.parametrize("case", ["points2", "points3", "targets", "extents", "level-restricted",
                                  "non-adaptive", "skip_prune", "weights", "f32"])
def test_debug_builds(actx, case):
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    dims = 2 if case in ("points2", "level-restricted", "non-adaptive") else 3
    dt = np.float32 if case == "f32" else np.float64
    p = [actx.from_numpy(a.astype(dt)) for a in cloud(30000, dims, 1)]
    kw = dict(max_particles_in_box=25)
    if case in ("targets", "extents"):
        kw["targets"] = [actx.from_numpy(a) for a in cloud(7000, dims, 2)]
    if case == "extents":
        kw.update(target_radii=actx.from_numpy(0.2 * 2.0 ** np.random.default_rng(3).uniform(-8, 0, 7000)),
                  stick_out_factor=0.25)
    if case == "level-restricted":
        kw["kind"] = "adaptive-level-restricted"
    if case == "non-adaptive":
        kw["kind"] = "non-adaptive"
    if case == "skip_prune":
        kw["skip_prune"] = True
    if case == "weights":
        del kw["max_particles_in_box"]
        w = torch.from_numpy(np.random.default_rng(4).integers(0, 5, 30000).astype(np.int32)).cuda()
        kw.update(refine_weights=w, max_leaf_refine_weight=40)
    tree, _ = TreeBuilder(actx)(actx, p, debug=True, **kw)
    if case != "skip_prune":
        trav, _ = FMMTraversalBuilder(actx)(actx, tree, debug=True)
        assert int(trav.target_boxes.shape[0]) > 0


def test_debug_checks_catch_damage(actx):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    from boxtree_amd.debug import check_traversal, check_tree
    torch = actx.torch
    p = [actx.from_numpy(a) for a in cloud(20000, 3, 5)]
    tree, _ = TreeBuilder(actx)(actx, p, max_particles_in_box=30)
    check_tree(torch, tree)

    def damaged(**changes):
        return dataclasses.replace(tree, **changes)

    bad = tree.box_parent_ids.clone()
    bad[5] = 7
    with pytest.raises(AssertionError, match="point back"):
        check_tree(torch, damaged(box_parent_ids=bad))
    bad = tree.user_source_ids.clone()
    bad[3] = bad[4]
    with pytest.raises(AssertionError, match="permutation|ids outside"):
        check_tree(torch, damaged(user_source_ids=bad))
    bad = tree.box_levels.clone()
    bad[2] = 3
    with pytest.raises(AssertionError, match="level"):
        check_tree(torch, damaged(box_levels=bad))
    bad = tree.box_source_counts_cumul.clone()
    bad[1] += 1
    with pytest.raises(AssertionError, match="cumul|root"):
        check_tree(torch, damaged(box_source_counts_cumul=bad, box_target_counts_cumul=bad))

    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    check_traversal(torch, trav, tree.nboxes)
    bad = trav.from_sep_siblings_lists.clone()
    bad[0] = int(tree.nboxes)
    with pytest.raises(AssertionError, match="not a box"):
        check_traversal(torch, dataclasses.replace(trav, from_sep_siblings_lists=bad), tree.nboxes)
    with pytest.raises(AssertionError, match="entries, starts end"):
        check_traversal(torch, dataclasses.replace(
            trav, neighbor_source_boxes_lists=trav.neighbor_source_boxes_lists[:-1]), tree.nboxes)
