"""Randomised parity against the oracle (tools/fuzz_parity.py: random dimensions,
dtypes, sizes, distributions, kinds, targets, radii, weights, n-away, criteria,
bounding boxes), plus the edge cases it has found."""

import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "tools"))


@pytest.mark.parametrize("first_seed", [0, 1000, 2000])
def test_random_configurations(first_seed):
    import fuzz_parity
    stats = fuzz_parity.run(80, first_seed, verbose=False, aux=False)
    assert (stats["ok"] + stats["max_levels"] + stats.get("beyond_int_shift", 0)
            + stats.get("beyond_key_depth", 0)) == 80
    assert stats["ok"] >= 70


def test_random_configurations_with_the_callers_next_to_the_path():
    """... plus peer lists, area / space-invader queries, target filters, translation
    and rotation classes, cost-model loops, depth-first order and work partition."""
    import fuzz_parity
    stats = fuzz_parity.run(40, 3000, verbose=False, aux=True)
    assert (stats["ok"] + stats["max_levels"] + stats.get("beyond_int_shift", 0)
            + stats.get("beyond_key_depth", 0)) == 40
    assert stats["ok"] >= 35


@pytest.mark.parametrize("kind", ["adaptive", "non-adaptive", "adaptive-level-restricted"])
@pytest.mark.parametrize("n", [1, 5])
def test_coincident_points_make_one_box(oracle, kind, n):
    """Bounding box of zero extent (tree_build.py:464-476 gives root_extent 0): one box
    as long as it need not split."""
    from compare import assert_same_tree
    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    p = [np.full(n, 0.25) for _ in range(3)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p], kind=kind,
                                max_particles_in_box=30)
    otree = oracle.build_tree(p, kind=kind, max_particles_in_box=30)
    assert_same_tree(actx.to_numpy(tree), otree)
    assert tree.nboxes == 1 and float(tree.root_extent) == 0.0


def test_coincident_points_that_must_split_exceed_the_levels(oracle):
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.tree_build import MaxLevelsExceeded
    actx = HIPArrayContext(0)
    p = [np.full(100, 0.25) for _ in range(2)]
    with pytest.raises(oracle.MaxLevelsExceeded):
        oracle.build_tree(p, max_particles_in_box=30)
    with pytest.raises(MaxLevelsExceeded):
        TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p], max_particles_in_box=30)


@pytest.mark.parametrize("n", [1, 3, 64, 65])
def test_non_adaptive_root_not_overfull(oracle, n):
    """tree_build.py:676: the level loop is skipped when the root's weight is within
    the limit -- also for kind='non-adaptive', which otherwise splits every box."""
    from compare import assert_same_tree
    from boxtree_amd import HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    rng = np.random.default_rng(n)
    p = [rng.standard_normal(n) for _ in range(3)]
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p], kind="non-adaptive",
                                max_particles_in_box=64)
    otree = oracle.build_tree(p, kind="non-adaptive", max_particles_in_box=64)
    assert_same_tree(actx.to_numpy(tree), otree)
    assert (tree.nboxes == 1) == (n <= 64)


@pytest.mark.parametrize("seed", [60020])
def test_points_stop_where_a_box_is_finer_than_the_coordinate_spacing(oracle, seed):
    """float32, 26 levels: below level ~22 a box is smaller than the spacing of the
    coordinates, and upstream's stick-out test fires for radius-0 points as well; the
    key generator may skip the test for points only above that depth (seed found by the
    fuzzer after the skip had been applied to every level)."""
    import fuzz_parity
    from compare import assert_same_traversal, assert_same_tree
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    p, t, kw, tkw = fuzz_parity.make_case(seed)
    assert p[0].dtype == np.float32 and "target_radii" in kw
    otree = oracle.build_tree(p, targets=t, **kw)
    assert otree.nlevels > 22
    dkw = dict(kw, target_radii=actx.from_numpy(kw["target_radii"]))
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p],
                                targets=[actx.from_numpy(a) for a in t], **dkw)
    assert_same_tree(actx.to_numpy(tree), otree)
    trav, _ = FMMTraversalBuilder(actx, **tkw)(actx, tree)
    assert_same_traversal(actx.to_numpy(trav), oracle.build_traversal(otree, **tkw))
