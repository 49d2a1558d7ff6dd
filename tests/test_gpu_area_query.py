"""HIP area queries vs the CPU oracle (bit-exact lists, same order) and vs the
brute-force properties the reference's tests check (test/test_tree.py:669-842,
:985-1041).  All cases call through the C ABI."""

import numpy as np
import pytest

from test_oracle_area_query import (check_area_query, check_peer_lists, leaf_geometry,
                                    normal_particles)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def build(actx, oracle, particles, **kw):
    from boxtree_amd import TreeBuilder
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in particles], **kw)
    otree = oracle.build_tree(particles, **kw)
    return tree, otree


def assert_same_csr(starts, lists, ostarts, olists):
    assert starts.dtype == np.int32 and lists.dtype == np.int32
    assert np.array_equal(starts, ostarts)
    assert np.array_equal(lists, olists)


def ball_sets(kind, tree, nballs, dims, dtype, seed):
    rng = np.random.default_rng(seed)
    if kind == "normal":            # test_tree.py:797-799
        return normal_particles(nballs, dims, dtype, seed), np.full(nballs, 0.1, dtype)
    if kind == "outside":           # test_tree.py:826-835
        lo, hi = tree.bounding_box[0].min(), tree.bounding_box[1].max()
        return ([rng.uniform(lo - 1, hi + 1, nballs).astype(dtype) for _ in range(dims)],
                np.full(nballs, 0.1, dtype))
    if kind == "mixed":             # radii from far below a leaf to beyond the root box
        return (normal_particles(nballs, dims, dtype, seed),
                (2.0 ** rng.uniform(-14, 3, nballs)).astype(dtype))
    raise ValueError(kind)


@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("mpb", [30, 3])
def test_peer_lists(actx, oracle, dims, dtype, mpb):
    from boxtree_amd import PeerListFinder
    particles = normal_particles(30000, dims, dtype, 3)
    tree, otree = build(actx, oracle, particles, max_particles_in_box=mpb)
    pl, _ = PeerListFinder(actx)(actx, tree)
    hpl = actx.to_numpy(pl)
    opl = oracle.peer_lists(otree)
    assert_same_csr(hpl.peer_list_starts, hpl.peer_lists, opl.peer_list_starts, opl.peer_lists)
    if dims > 1:
        check_peer_lists(otree, hpl)


def test_peer_lists_single_box(actx, oracle):
    from boxtree_amd import AreaQueryBuilder, PeerListFinder
    particles = normal_particles(5, 2, np.float64, 3)
    tree, otree = build(actx, oracle, particles, max_particles_in_box=30)
    assert tree.nboxes == 1
    pl, _ = PeerListFinder(actx)(actx, tree)
    hpl = actx.to_numpy(pl)
    assert hpl.peer_list_starts.tolist() == [0, 1] and hpl.peer_lists.tolist() == [0]
    bc = normal_particles(7, 2, np.float64, 4)
    br = np.full(7, 0.1)
    aq, _ = AreaQueryBuilder(actx)(actx, tree, [actx.from_numpy(b) for b in bc],
                                   actx.from_numpy(br))
    oaq = oracle.area_query(otree, bc, br)
    haq = actx.to_numpy(aq)
    assert_same_csr(haq.leaves_near_ball_starts, haq.leaves_near_ball_lists,
                    oaq.leaves_near_ball_starts, oaq.leaves_near_ball_lists)


@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["normal", "outside", "mixed"])
def test_area_query(actx, oracle, dims, dtype, kind):
    from boxtree_amd import AreaQueryBuilder
    particles = normal_particles(20000, dims, dtype, 5)
    tree, otree = build(actx, oracle, particles, max_particles_in_box=30)
    nballs = 1500
    bc, br = ball_sets(kind, otree, nballs, dims, dtype, 11)
    aq, _ = AreaQueryBuilder(actx)(actx, tree, [actx.from_numpy(b) for b in bc],
                                   actx.from_numpy(br))
    haq = actx.to_numpy(aq)
    oaq = oracle.area_query(otree, bc, br)
    assert_same_csr(haq.leaves_near_ball_starts, haq.leaves_near_ball_lists,
                    oaq.leaves_near_ball_starts, oaq.leaves_near_ball_lists)
    if dims > 1 and kind != "mixed":
        check_area_query(otree, haq, bc, br, first=300)


@pytest.mark.parametrize("dims", [2, 3])
def test_area_query_reference_sizes(actx, dims):
    """test_area_query (test_tree.py:773-799) at the reference's own sizes,
    checked by its brute-force property."""
    from boxtree_amd import AreaQueryBuilder, TreeBuilder
    dtype = np.float64
    particles = normal_particles(10**5, dims, dtype, 15)
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(p) for p in particles],
                                max_particles_in_box=30)
    nballs = 10**4
    bc = normal_particles(nballs, dims, dtype, 16)
    br = np.full(nballs, 0.1, dtype)
    aq, _ = AreaQueryBuilder(actx)(actx, tree, [actx.from_numpy(b) for b in bc],
                                   actx.from_numpy(br))
    htree = actx.to_numpy(tree)
    haq = actx.to_numpy(aq)
    assert len(haq.leaves_near_ball_starts) == nballs + 1
    check_area_query(htree, haq, bc, br, first=400)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_leaves_to_balls_and_space_invader(actx, oracle, dims, dtype):
    from boxtree_amd import (LeavesToBallsLookupBuilder, PeerListFinder,
                             SpaceInvaderQueryBuilder)
    particles = normal_particles(20000, dims, dtype, 8)
    tree, otree = build(actx, oracle, particles, max_particles_in_box=30)
    nballs = 3000
    bc = normal_particles(nballs, dims, dtype, 9)
    br = np.full(nballs, 0.1, dtype)
    dbc = [actx.from_numpy(b) for b in bc]
    dbr = actx.from_numpy(br)
    pl, _ = PeerListFinder(actx)(actx, tree)
    lbl, _ = LeavesToBallsLookupBuilder(actx)(actx, tree, dbc, dbr, peer_lists=pl)
    hl = actx.to_numpy(lbl)
    ol = oracle.leaves_to_balls(otree, bc, br)
    assert_same_csr(hl.balls_near_box_starts, hl.balls_near_box_lists,
                    ol.balls_near_box_starts, ol.balls_near_box_lists)
    siq, _ = SpaceInvaderQueryBuilder(actx)(actx, tree, dbc, dbr)
    hs = actx.to_numpy(siq)
    osq = oracle.space_invader_query(otree, bc, br)
    assert hs.dtype == dtype and hs.shape == (otree.nboxes,)
    assert np.array_equal(hs, osq)
    # test_tree.py:1027-1041
    leaves, rad, ctr = leaf_geometry(otree)
    expect = np.zeros(otree.nboxes)
    bca = np.array(bc, dtype=np.float64)
    for leaf in leaves:
        s, t = hl.balls_near_box_starts[leaf:leaf + 2]
        inv = hl.balls_near_box_lists[s:t]
        if len(inv):
            expect[leaf] = np.max(np.abs(
                otree.box_centers[:, leaf].reshape(-1, 1).astype(np.float64) - bca[:, inv]))
    assert np.allclose(hs, expect, rtol=1e-6 if dtype == np.float32 else 1e-7)


def test_area_query_empty_and_errors(actx, oracle):
    from boxtree_amd import AreaQueryBuilder, LeavesToBallsLookupBuilder, PeerListFinder
    from boxtree_amd.area_query import PeerListLookup
    particles = normal_particles(3000, 2, np.float64, 1)
    tree, otree = build(actx, oracle, particles, max_particles_in_box=30)
    aqb = AreaQueryBuilder(actx)
    empty = [actx.from_numpy(np.zeros(0)) for _ in range(2)]
    aq, _ = aqb(actx, tree, empty, actx.from_numpy(np.zeros(0)))
    haq = actx.to_numpy(aq)
    assert haq.leaves_near_ball_starts.tolist() == [0] and len(haq.leaves_near_ball_lists) == 0
    lbl, _ = LeavesToBallsLookupBuilder(actx)(actx, tree, empty, actx.from_numpy(np.zeros(0)))
    hl = actx.to_numpy(lbl)
    assert np.array_equal(hl.balls_near_box_starts, np.zeros(otree.nboxes + 1, np.int32))
    # balls that touch nothing: far outside the bounding box
    far = [actx.from_numpy(np.full(4, 1e3)) for _ in range(2)]
    aq, _ = aqb(actx, tree, far, actx.from_numpy(np.full(4, 0.1)))
    assert actx.to_numpy(aq).leaves_near_ball_starts.tolist() == [0] * 5
    bc32 = [actx.from_numpy(b) for b in normal_particles(10, 2, np.float32, 2)]
    bc64 = [actx.from_numpy(b) for b in normal_particles(10, 2, np.float64, 2)]
    with pytest.raises(TypeError):          # area_query.py:764-766
        aqb(actx, tree, bc32, actx.from_numpy(np.ones(10)))
    with pytest.raises(TypeError):          # area_query.py:767-768
        aqb(actx, tree, bc64, actx.from_numpy(np.ones(10, np.float32)))
    pl, _ = PeerListFinder(actx)(actx, tree)
    bad = PeerListLookup(tree=tree, peer_list_starts=pl.peer_list_starts[:-1],
                         peer_lists=pl.peer_lists)
    with pytest.raises(ValueError):         # area_query.py:781-782
        aqb(actx, tree, bc64, actx.from_numpy(np.ones(10)), peer_lists=bad)


def test_area_query_large(actx):
    """10^6 particles / 10^6 balls: size-independent properties (every ball's own
    leaf is found; transposing twice gives the query back)."""
    import torch
    from boxtree_amd import AreaQueryBuilder, LeavesToBallsLookupBuilder, TreeBuilder
    n = 10**6
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
    tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=32)
    radii = torch.full((n,), 1e-3, dtype=torch.float64, device="cuda")
    aq, _ = AreaQueryBuilder(actx)(actx, tree, pts, radii)
    starts = aq.leaves_near_ball_starts.long()
    lists = aq.leaves_near_ball_lists.long()
    assert int(starts[-1]) == lists.shape[0]
    counts = starts[1:] - starts[:-1]
    assert int(counts.min()) >= 1
    # the leaf that holds sorted particle i is among the leaves of ball user_source_ids[i]
    flags = tree.box_flags.long()
    assert bool(((flags[lists] & 12) == 0).all())
    leaves = torch.nonzero((flags & 12) == 0).flatten()
    cn = tree.box_source_counts_nonchild.long()[leaves]
    leaves = leaves[cn > 0]
    st, order = torch.sort(tree.box_source_starts.long()[leaves])
    leaf_of_sorted = leaves[order][torch.searchsorted(
        st, torch.arange(n, device="cuda"), right=True) - 1]
    own_leaf = torch.empty(n, dtype=torch.int64, device="cuda")
    own_leaf[tree.user_source_ids.long()] = leaf_of_sorted
    ball_of_entry = torch.repeat_interleave(torch.arange(n, device="cuda"), counts)
    hit = torch.zeros(n, dtype=torch.int64, device="cuda")
    hit.index_add_(0, ball_of_entry, (lists == own_leaf[ball_of_entry]).long())
    assert bool((hit == 1).all())
    lbl, _ = LeavesToBallsLookupBuilder(actx)(actx, tree, pts, radii)
    bstarts = lbl.balls_near_box_starts.long()
    blists = lbl.balls_near_box_lists.long()
    assert blists.shape[0] == lists.shape[0]
    bcounts = bstarts[1:] - bstarts[:-1]
    box_of_entry = torch.repeat_interleave(torch.arange(tree.nboxes, device="cuda"), bcounts)
    # (ball, leaf) pairs of both tables agree as multisets
    a = torch.sort(ball_of_entry * tree.nboxes + lists).values
    b = torch.sort(blists * tree.nboxes + box_of_entry).values
    assert torch.equal(a, b)
    # stable: ball numbers ascend within every box
    same_box = box_of_entry[1:] == box_of_entry[:-1]
    assert bool((blists[1:][same_box] > blists[:-1][same_box]).all())
