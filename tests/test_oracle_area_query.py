"""Oracle area queries pinned by the properties the reference's own tests check
(test/test_tree.py:669-842 brute-force leaf/ball overlap, :985-1041 space invader)."""

import numpy as np
import pytest

HAS_CHILDREN = 12


def normal_particles(n, dims, dtype, seed):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(n).astype(dtype) for _ in range(dims)]


def leaf_geometry(tree):
    leaves, = ((tree.box_flags & HAS_CHILDREN) == 0).nonzero()
    rad = tree.root_extent * 0.5 ** (tree.box_levels[leaves].astype(np.float64) + 1)
    ctr = tree.box_centers[:, leaves].T.astype(np.float64)
    return leaves, rad, ctr


def brute_force_area_query(tree, ball_centers, ball_radii):
    """test_tree.py:742-768"""
    leaves, rad, ctr = leaf_geometry(tree)
    bc = np.array(ball_centers, dtype=np.float64).T
    res = []
    for c, r in zip(bc, ball_radii):
        d = np.max(np.abs(c - ctr), axis=-1)
        res.append(set(leaves[d < r + rad].tolist()))
    return res


def check_area_query(tree, aq, ball_centers, ball_radii, first=None):
    """Brute-force check (of the first *first* balls only, if given)."""
    assert len(aq.leaves_near_ball_starts) == len(ball_radii) + 1
    if first is not None:
        ball_centers = [b[:first] for b in ball_centers]
        ball_radii = ball_radii[:first]
    expect = brute_force_area_query(tree, ball_centers, ball_radii)
    for i, e in enumerate(expect):
        s, t = aq.leaves_near_ball_starts[i:i + 2]
        found = aq.leaves_near_ball_lists[s:t]
        assert len(set(found.tolist())) == len(found)
        assert set(found.tolist()) == e, (i, found, e)


def check_peer_lists(tree, pl):
    """Definition from area_query.py:1067-1096: peers are adjacent-or-overlapping,
    of at least the box's size (or leaves), with no adjacent child."""
    starts, lists = pl.peer_list_starts, pl.peer_lists
    assert len(starts) == tree.nboxes + 1
    lev = tree.box_levels.astype(np.int64)
    rad = tree.root_extent * 0.5 ** (lev + 1.0)
    ctr = tree.box_centers[:, :tree.nboxes].astype(np.float64)
    def adjacent(b, others):
        # traversal.py:279-305: slack of half the smaller box separates "touching"
        # from "one box apart" robustly in floating point
        d = np.max(np.abs(ctr[:, others] - ctr[:, [b]]), axis=0)
        return d <= rad[others] + rad[b] + np.minimum(rad[others], rad[b])

    for b in range(tree.nboxes):
        peers = lists[starts[b]:starts[b + 1]]
        assert len(peers) >= 1
        assert len(set(peers.tolist())) == len(peers)
        assert np.all(adjacent(b, peers))
        assert np.all(lev[peers] <= lev[b])
        for p in peers[lev[peers] < lev[b]]:
            if tree.box_flags[p] & HAS_CHILDREN:
                ch = tree.box_child_ids[:, p]
                ch = ch[ch != 0]
                assert not np.any(adjacent(b, ch))


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_area_query(oracle, dims, dtype):
    particles = normal_particles(20000, dims, dtype, 3)
    tree = oracle.build_tree(particles, max_particles_in_box=30)
    nballs = 600
    ball_centers = normal_particles(nballs, dims, dtype, 4)
    ball_radii = np.full(nballs, 0.1, dtype)
    pl = oracle.peer_lists(tree)
    check_peer_lists(tree, pl)
    aq = oracle.area_query(tree, ball_centers, ball_radii, peer_lists=pl)
    check_area_query(tree, aq, ball_centers, ball_radii)


@pytest.mark.parametrize("dims", [2, 3])
def test_oracle_area_query_balls_outside_bbox(oracle, dims):
    dtype = np.float64
    particles = normal_particles(10000, dims, dtype, 5)
    tree = oracle.build_tree(particles, max_particles_in_box=30)
    nballs = 600
    rng = np.random.default_rng(13)
    lo, hi = tree.bounding_box[0].min(), tree.bounding_box[1].max()
    ball_centers = [rng.uniform(lo - 1, hi + 1, nballs).astype(dtype) for _ in range(dims)]
    ball_radii = np.full(nballs, 0.1, dtype)
    aq = oracle.area_query(tree, ball_centers, ball_radii)
    check_area_query(tree, aq, ball_centers, ball_radii)


@pytest.mark.parametrize("dims", [2, 3])
def test_oracle_area_query_mixed_radii(oracle, dims):
    dtype = np.float64
    particles = normal_particles(10000, dims, dtype, 6)
    tree = oracle.build_tree(particles, max_particles_in_box=10)
    nballs = 500
    rng = np.random.default_rng(14)
    ball_centers = normal_particles(nballs, dims, dtype, 7)
    ball_radii = (2.0 ** rng.uniform(-12, 2, nballs)).astype(dtype)
    aq = oracle.area_query(tree, ball_centers, ball_radii)
    check_area_query(tree, aq, ball_centers, ball_radii)


@pytest.mark.parametrize("dims", [2, 3])
def test_oracle_leaves_to_balls_and_space_invader(oracle, dims):
    dtype = np.float64
    particles = normal_particles(10000, dims, dtype, 8)
    tree = oracle.build_tree(particles, max_particles_in_box=30)
    nballs = 500
    ball_centers = normal_particles(nballs, dims, dtype, 9)
    ball_radii = np.full(nballs, 0.1, dtype)
    lbl = oracle.leaves_to_balls(tree, ball_centers, ball_radii)
    assert len(lbl.balls_near_box_starts) == tree.nboxes + 1
    leaves, rad, ctr = leaf_geometry(tree)
    bc = np.array(ball_centers).T
    expect_dist = np.zeros(tree.nboxes)
    for leaf, r, c in zip(leaves, rad, ctr):
        d = np.max(np.abs(bc - c), axis=-1)
        near, = np.where(d - ball_radii < r)
        s, t = lbl.balls_near_box_starts[leaf:leaf + 2]
        got = lbl.balls_near_box_lists[s:t]
        assert np.array_equal(got, near)          # stable sort: ascending ball number
        if len(near):
            expect_dist[leaf] = d[near].max()
    nonleaf = (tree.box_flags & HAS_CHILDREN) != 0
    assert np.all(np.diff(lbl.balls_near_box_starts)[nonleaf] == 0)
    siq = oracle.space_invader_query(tree, ball_centers, ball_radii)
    assert siq.dtype == dtype and siq.shape == (tree.nboxes,)
    assert np.allclose(siq, expect_dist)          # test_tree.py:1041


def test_oracle_area_query_arg_errors(oracle):
    particles = normal_particles(1000, 2, np.float64, 1)
    tree = oracle.build_tree(particles, max_particles_in_box=30)
    bc32 = normal_particles(10, 2, np.float32, 2)
    bc64 = normal_particles(10, 2, np.float64, 2)
    with pytest.raises(TypeError):
        oracle.area_query(tree, bc32, np.ones(10))
    with pytest.raises(TypeError):
        oracle.area_query(tree, bc64, np.ones(10, np.float32))
