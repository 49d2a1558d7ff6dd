"""Oracle: kind="adaptive-level-restricted" pinned by the property the reference
checks (test/test_tree.py:900-974: neighbouring leaves differ by at most one
level) plus the tree invariants of the plain adaptive kind."""

import numpy as np
import pytest

from invariants import check_tree

HAS_CHILDREN = 12


def surface_particles(n, dims, seed=15):
    """Points on a circle / sphere with a dense patch: deep, strongly graded trees."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((dims, n))
    v /= np.sqrt((v * v).sum(axis=0))
    v[:, : n // 3] = v[:, :1] + 1e-3 * rng.standard_normal((dims, n // 3))
    return [np.ascontiguousarray(v[i]) for i in range(dims)]


def check_level_restriction(tree):
    """Brute force over leaf pairs: touching leaves are at most one level apart."""
    nb = tree.nboxes
    leaf = np.nonzero((tree.box_flags[:nb] & HAS_CHILDREN) == 0)[0]
    lev = tree.box_levels[leaf].astype(np.int64)
    rad = tree.root_extent * 0.5 ** (lev + 1.0)
    ctr = tree.box_centers[:, leaf].astype(np.float64)
    # compare every leaf with all leaves at least two levels shallower
    for L in range(int(lev.max()), 1, -1):
        deep = np.nonzero(lev == L)[0]
        shallow = np.nonzero(lev <= L - 2)[0]
        if len(deep) == 0 or len(shallow) == 0:
            continue
        for chunk in np.array_split(deep, max(1, len(deep) // 2000)):
            d = np.max(np.abs(ctr[:, chunk][:, :, None] - ctr[:, shallow][:, None, :]), axis=0)
            touching = d <= (rad[chunk][:, None] + rad[shallow][None, :]) * (1 + 1e-9)
            assert not touching.any(), (L, np.argwhere(touching)[:3])
    return len(leaf)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("skip_prune", [True, False])
def test_oracle_level_restriction(oracle, dims, skip_prune):
    p = surface_particles(20000, dims)
    tree = oracle.build_tree(p, kind="adaptive-level-restricted", max_particles_in_box=30,
                             skip_prune=skip_prune)
    plain = oracle.build_tree(p, kind="adaptive", max_particles_in_box=30,
                              skip_prune=skip_prune)
    assert tree.nboxes > plain.nboxes          # this geometry needs balancing splits
    assert tree.nlevels == plain.nlevels
    check_level_restriction(tree)
    with pytest.raises(AssertionError):
        check_level_restriction(plain)
    assert sorted(tree.user_source_ids.tolist()) == list(range(20000))
    if not skip_prune:
        check_tree(tree, p, max_particles_in_box=30)


@pytest.mark.parametrize("dims", [2, 3])
def test_oracle_level_restricted_equals_adaptive_when_balanced(oracle, dims):
    rng = np.random.default_rng(2)
    p = [rng.random(5000) for _ in range(dims)]
    a = oracle.build_tree(p, kind="adaptive", max_particles_in_box=200)
    b = oracle.build_tree(p, kind="adaptive-level-restricted", max_particles_in_box=200)
    # a uniform cloud at this resolution is already 2:1 balanced
    assert a.nboxes == b.nboxes
    for name in ("box_levels", "box_parent_ids", "box_child_ids", "box_centers",
                 "box_source_starts", "box_source_counts_cumul", "user_source_ids"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_oracle_level_restricted_with_targets_and_extents(oracle):
    rng = np.random.default_rng(4)
    s = surface_particles(8000, 2, seed=4)
    t = surface_particles(6000, 2, seed=5)
    tr = 2.0 ** rng.uniform(-12, -4, 6000)
    tree = oracle.build_tree(s, targets=t, target_radii=tr, stick_out_factor=0.25,
                             kind="adaptive-level-restricted", max_particles_in_box=20)
    check_level_restriction(tree)
    check_tree(tree, s, targets=t, target_radii=tr, max_particles_in_box=20,
               extent_norm="linf")
