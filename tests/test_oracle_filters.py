"""Oracle target filters and point-source linking, pinned by what the reference's
tests check: filtered FMM completeness (test/test_fmm.py:244-285) and the
point-source construction of test_extent_tree (test/test_tree.py:636-661)."""

import numpy as np
import pytest

from invariants import constant_one_potentials


def normal(n, dims, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(n).astype(dtype) for _ in range(dims)]


def filter_case(oracle, dims, ntargets, extent):
    rng = np.random.default_rng(22)
    sources = normal(4000, dims, 1)
    kw = dict(max_particles_in_box=30)
    targets = None
    if ntargets:
        targets = normal(ntargets, dims, 2)
        kw["targets"] = targets
    if extent:
        kw["target_radii"] = 2.0 ** rng.uniform(-10, 0, ntargets)
        kw["stick_out_factor"] = 0.25
    tree = oracle.build_tree(sources, **kw)
    trav = oracle.build_traversal(tree)
    flags = rng.integers(0, 2, ntargets or 4000, dtype=np.int8)
    return tree, trav, flags


def check_filters(tree, trav, flags, fu, ft):
    nsources = tree.nsources
    sel = flags > 0
    # user order (test_fmm.py:255-259, :283-285)
    pot = constant_one_potentials(tree, trav, filtered_user=fu)
    assert np.all(pot[sel] == nsources) and np.all(pot[~sel] == 0)
    assert fu.nfiltered_targets == int(sel.sum()) == len(fu.target_lists)
    assert len(fu.target_starts) == tree.nboxes + 1
    assert sorted(fu.target_lists.tolist()) == np.nonzero(sel)[0].tolist()
    # tree order (test_fmm.py:260-264)
    pot = constant_one_potentials(tree, trav, filtered_tree=ft)
    assert np.all(pot[sel] == nsources) and np.all(pot[~sel] == 0)
    assert ft.nfiltered_targets == int(sel.sum())
    for ax in range(tree.dimensions):
        assert np.array_equal(
            ft.targets[ax],
            np.asarray(tree.targets[ax])[ft.unfiltered_from_filtered_target_indices])


@pytest.mark.parametrize("dims,ntargets,extent", [(2, None, False), (3, 3000, False),
                                                  (3, 3000, True)])
def test_oracle_target_filters(oracle, dims, ntargets, extent):
    tree, trav, flags = filter_case(oracle, dims, ntargets, extent)
    fu = oracle.filter_target_lists_in_user_order(tree, flags)
    ft = oracle.filter_target_lists_in_tree_order(tree, flags)
    check_filters(tree, trav, flags, fu, ft)


def point_source_case(oracle, dims, per_source, seed=5):
    """test_tree.py:636-655"""
    rng = np.random.default_rng(seed)
    nsources = 3000
    sources = normal(nsources, dims, 3)
    radii = 2.0 ** rng.uniform(-10, 0, nsources)
    tree = oracle.build_tree(sources, source_radii=radii, targets=normal(500, dims, 4),
                             stick_out_factor=0.25, max_particles_in_box=10)
    if per_source == "ragged":
        counts = rng.integers(0, 5, nsources)
        counts[-1] = 2
    else:
        counts = np.full(nsources, per_source)
    starts = np.zeros(nsources + 1, np.int32)
    starts[1:] = np.cumsum(counts)
    npts = int(starts[-1])
    owner = np.repeat(np.arange(nsources), counts)
    point_sources = [sources[i][owner] + radii[owner] * rng.uniform(-1, 1, npts)
                     for i in range(dims)]
    return tree, starts, point_sources, owner


def check_point_sources(tree, starts, point_sources, owner, r):
    npts = int(starts[-1])
    assert r.npoint_sources == npts
    assert sorted(r.user_point_source_ids.tolist()) == list(range(npts))
    usi = tree.user_source_ids
    for isrc in range(tree.nsources):
        s, c = r.point_source_starts[isrc], r.point_source_counts[isrc]
        u = usi[isrc]
        assert c == starts[u + 1] - starts[u]
        assert np.array_equal(r.user_point_source_ids[s:s + c], np.arange(starts[u], starts[u + 1]))
    for ax in range(tree.dimensions):
        assert np.array_equal(r.point_sources[ax], point_sources[ax][r.user_point_source_ids])
    # the point sources of a box are exactly those of its sources
    for ibox in range(tree.nboxes):
        for kind in ("nonchild", "cumul"):
            cnt = getattr(tree, "box_source_counts_" + kind)[ibox]
            s0 = tree.box_source_starts[ibox]
            expect = owner_count(starts, usi[s0:s0 + cnt])
            got = getattr(r, "box_point_source_counts_" + kind)[ibox]
            assert got == expect
            if cnt:
                ps0 = r.box_point_source_starts[ibox]
                ids = r.user_point_source_ids[ps0:ps0 + got]
                assert set(owner[ids].tolist()) <= set(usi[s0:s0 + cnt].tolist())


def owner_count(starts, users):
    return int(np.sum(starts[users + 1] - starts[users]))


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("per_source", [16, 1, "ragged"])
def test_oracle_link_point_sources(oracle, dims, per_source):
    tree, starts, point_sources, owner = point_source_case(oracle, dims, per_source)
    r = oracle.link_point_sources(tree, starts, point_sources)
    check_point_sources(tree, starts, point_sources, owner, r)


def test_oracle_link_point_sources_needs_extent(oracle):
    tree = oracle.build_tree(normal(100, 2, 1), max_particles_in_box=10)
    with pytest.raises(ValueError):
        oracle.link_point_sources(tree, np.arange(101, dtype=np.int32), normal(100, 2, 2))
