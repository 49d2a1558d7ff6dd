"""kind="adaptive-level-restricted" on the device vs the CPU oracle (all arrays
identical) and the reference's own check (test/test_tree.py:900-974)."""

import numpy as np
import pytest

from compare import assert_same_traversal, assert_same_tree
from invariants import check_tree
from test_oracle_level_restricted import check_level_restriction, surface_particles

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def build(actx, oracle, particles, **kw):
    from boxtree_amd import TreeBuilder
    dkw = dict(kw)
    if dkw.get("targets") is not None:
        dkw["targets"] = [actx.from_numpy(a) for a in dkw["targets"]]
    for name in ("source_radii", "target_radii", "refine_weights"):
        if dkw.get(name) is not None:
            dkw[name] = actx.from_numpy(dkw[name])
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in particles], **dkw)
    otree = oracle.build_tree(particles, **kw)
    htree = actx.to_numpy(tree)
    assert_same_tree(htree, otree)
    return tree, htree, otree


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("skip_prune", [True, False])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_level_restricted_tree(actx, oracle, dims, skip_prune, dtype):
    p = [a.astype(dtype) for a in surface_particles(30000, dims)]
    mpb = 30 if dtype == np.float64 else 100
    _, htree, otree = build(actx, oracle, p, kind="adaptive-level-restricted",
                            max_particles_in_box=mpb, skip_prune=skip_prune)
    check_level_restriction(htree)
    if not skip_prune:
        check_tree(htree, p, max_particles_in_box=mpb)


@pytest.mark.parametrize("dims", [2, 3])
def test_level_restricted_normal_cloud(actx, oracle, dims):
    rng = np.random.default_rng(15)
    p = [rng.standard_normal(10**5) for _ in range(dims)]
    _, htree, _ = build(actx, oracle, p, kind="adaptive-level-restricted",
                        max_particles_in_box=30)
    check_tree(htree, p, max_particles_in_box=30)


def test_level_restricted_single_box_and_balanced(actx, oracle):
    rng = np.random.default_rng(1)
    build(actx, oracle, [rng.random(20) for _ in range(2)], kind="adaptive-level-restricted",
          max_particles_in_box=30)
    build(actx, oracle, [rng.random(5000) for _ in range(3)], kind="adaptive-level-restricted",
          max_particles_in_box=200)


def test_level_restricted_targets_extents_weights(actx, oracle):
    rng = np.random.default_rng(4)
    s = surface_particles(8000, 2, seed=4)
    t = surface_particles(6000, 2, seed=5)
    tr = 2.0 ** rng.uniform(-12, -4, 6000)
    build(actx, oracle, s, targets=t, target_radii=tr, stick_out_factor=0.25,
          kind="adaptive-level-restricted", max_particles_in_box=20)
    rw = rng.integers(0, 4, 14000).astype(np.int32)
    build(actx, oracle, s, targets=t, refine_weights=rw, max_leaf_refine_weight=40,
          kind="adaptive-level-restricted")


@pytest.mark.parametrize("dims", [2, 3])
def test_level_restricted_traversal_and_area_query(actx, oracle, dims):
    """The reference's own test: neighbouring leaves found by an area query differ
    by at most one level (test_tree.py:928-971); plus traversal parity (boxes of a
    level are no longer in Morton order: the walk-from-root kernels run)."""
    from boxtree_amd import AreaQueryBuilder, FMMTraversalBuilder
    p = surface_particles(20000, dims, seed=9)
    tree, htree, otree = build(actx, oracle, p, kind="adaptive-level-restricted",
                               max_particles_in_box=30)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    assert_same_traversal(actx.to_numpy(trav), oracle.build_traversal(otree))
    leaf_boxes, = ((htree.box_flags & 12) == 0).nonzero()
    rad = htree.root_extent * 0.5 ** (htree.box_levels[leaf_boxes].astype(np.float64) + 1)
    centers = [np.ascontiguousarray(htree.box_centers[ax, leaf_boxes]) for ax in range(dims)]
    ball_radii = np.min(rad) / 2 + rad
    aq, _ = AreaQueryBuilder(actx)(actx, tree, [actx.from_numpy(c) for c in centers],
                                   actx.from_numpy(ball_radii))
    aq = actx.to_numpy(aq)
    lev = htree.box_levels.astype(np.int64)
    owner = np.repeat(np.arange(len(leaf_boxes)), np.diff(aq.leaves_near_ball_starts))
    assert np.all(np.abs(lev[aq.leaves_near_ball_lists] - lev[leaf_boxes][owner]) <= 1)
