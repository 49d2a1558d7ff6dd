"""HIP target filters and point-source linking vs the CPU oracle (bit-exact) and
the properties of the reference's tests (test_fmm.py:244-285, test_tree.py:636-661)."""

import numpy as np
import pytest

from test_oracle_filters import (check_filters, check_point_sources, filter_case, normal,
                                 point_source_case)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def device_tree(actx, otree_kw, sources, **kw):
    from boxtree_amd import TreeBuilder
    dkw = dict(kw)
    for name in ("targets",):
        if dkw.get(name) is not None:
            dkw[name] = [actx.from_numpy(a) for a in dkw[name]]
    for name in ("source_radii", "target_radii"):
        if dkw.get(name) is not None:
            dkw[name] = actx.from_numpy(dkw[name])
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(s) for s in sources], **dkw)
    return tree


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.parametrize("dims,ntargets,extent", [(2, None, False), (3, 3000, False),
                                                  (3, 3000, True)])
def test_target_filters(actx, oracle, dims, ntargets, extent):
    from boxtree_amd import FMMTraversalBuilder
    from boxtree_amd.tree import ParticleListFilter
    otree, otrav, flags = filter_case(oracle, dims, ntargets, extent)
    rng = np.random.default_rng(22)
    sources = normal(4000, dims, 1)
    kw = dict(max_particles_in_box=30)
    if ntargets:
        kw["targets"] = normal(ntargets, dims, 2)
    if extent:
        kw["target_radii"] = 2.0 ** rng.uniform(-10, 0, ntargets)
        kw["stick_out_factor"] = 0.25
    tree = device_tree(actx, None, sources, **kw)
    plfilt = ParticleListFilter(actx)
    dflags = actx.from_numpy(flags)
    fu = actx.to_numpy(plfilt.filter_target_lists_in_user_order(actx, tree, dflags))
    ft = actx.to_numpy(plfilt.filter_target_lists_in_tree_order(actx, tree, dflags))
    ofu = oracle.filter_target_lists_in_user_order(otree, flags)
    oft = oracle.filter_target_lists_in_tree_order(otree, flags)
    assert fu.nfiltered_targets == ofu.nfiltered_targets
    same(fu.target_starts, ofu.target_starts)
    same(fu.target_lists, ofu.target_lists)
    assert ft.nfiltered_targets == oft.nfiltered_targets
    same(ft.box_target_starts, oft.box_target_starts)
    same(ft.box_target_counts_nonchild, oft.box_target_counts_nonchild)
    same(ft.unfiltered_from_filtered_target_indices, oft.unfiltered_from_filtered_target_indices)
    for ax in range(dims):
        same(ft.targets[ax], oft.targets[ax])
    # FMM completeness with the device traversal and the device filters
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    check_filters(actx.to_numpy(tree), actx.to_numpy(trav), flags, fu, ft)


@pytest.mark.parametrize("flagkind", ["none", "all"])
def test_target_filters_trivial(actx, oracle, flagkind):
    from boxtree_amd.tree import ParticleListFilter
    sources = normal(2000, 2, 1)
    tree = device_tree(actx, None, sources, max_particles_in_box=30)
    otree = oracle.build_tree(sources, max_particles_in_box=30)
    flags = np.zeros(2000, np.int8) if flagkind == "none" else np.full(2000, 3, np.int8)
    plfilt = ParticleListFilter(actx)
    fu = actx.to_numpy(plfilt.filter_target_lists_in_user_order(actx, tree, actx.from_numpy(flags)))
    ft = actx.to_numpy(plfilt.filter_target_lists_in_tree_order(actx, tree, actx.from_numpy(flags)))
    ofu = oracle.filter_target_lists_in_user_order(otree, flags)
    oft = oracle.filter_target_lists_in_tree_order(otree, flags)
    same(fu.target_starts, ofu.target_starts)
    same(fu.target_lists, ofu.target_lists)
    same(ft.box_target_starts, oft.box_target_starts)
    same(ft.box_target_counts_nonchild, oft.box_target_counts_nonchild)
    assert ft.nfiltered_targets == (0 if flagkind == "none" else 2000)
    with pytest.raises(TypeError):
        plfilt.filter_target_lists_in_user_order(actx, tree, actx.from_numpy(flags.astype(np.int32)))


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("per_source", [16, 1, "ragged"])
def test_link_point_sources(actx, oracle, dims, per_source):
    from boxtree_amd.tree import TreeWithLinkedPointSources, link_point_sources
    otree, starts, point_sources, owner = point_source_case(oracle, dims, per_source)
    rng = np.random.default_rng(5)
    nsources = 3000
    sources = normal(nsources, dims, 3)
    radii = 2.0 ** rng.uniform(-10, 0, nsources)
    tree = device_tree(actx, None, sources, source_radii=radii, targets=normal(500, dims, 4),
                       stick_out_factor=0.25, max_particles_in_box=10)
    r = link_point_sources(actx, tree, actx.from_numpy(starts),
                           [actx.from_numpy(p) for p in point_sources], debug=True)
    assert isinstance(r, TreeWithLinkedPointSources)
    assert r.sources_have_extent and r.nboxes == otree.nboxes
    h = actx.to_numpy(r)
    o = oracle.link_point_sources(otree, starts, point_sources)
    assert h.npoint_sources == o.npoint_sources
    for name in ("point_source_starts", "point_source_counts", "user_point_source_ids",
                 "box_point_source_starts", "box_point_source_counts_nonchild",
                 "box_point_source_counts_cumul"):
        same(getattr(h, name), getattr(o, name))
    for ax in range(dims):
        same(h.point_sources[ax], o.point_sources[ax])
    check_point_sources(otree, starts, point_sources, owner, h)


def test_link_point_sources_2d_point_arrays(actx, oracle):
    """The reference's own test hands [nsources, k] arrays (test/test_tree.py:638-656);
    cl_array.take indexes their flat storage."""
    from boxtree_amd.tree import link_point_sources
    dims, nsources, k = 3, 2000, 8
    rng = np.random.default_rng(8)
    sources = normal(nsources, dims, 3)
    radii = 2.0 ** rng.uniform(-10, 0, nsources)
    kw = dict(source_radii=radii, targets=normal(400, dims, 4), stick_out_factor=0.25,
              max_particles_in_box=10)
    tree = device_tree(actx, None, sources, **kw)
    otree = oracle.build_tree(sources, **kw)
    point_sources = [s[:, None] + radii[:, None] * rng.uniform(-1, 1, (nsources, k))
                     for s in sources]
    starts = np.arange(0, (nsources + 1) * k, k, dtype=np.int32)
    r = actx.to_numpy(link_point_sources(actx, tree, actx.from_numpy(starts),
                                         [actx.from_numpy(p) for p in point_sources]))
    o = oracle.link_point_sources(otree, starts, point_sources)
    assert r.npoint_sources == o.npoint_sources == nsources * k
    same(r.user_point_source_ids, o.user_point_source_ids)
    for ax in range(dims):
        same(r.point_sources[ax], o.point_sources[ax])
        same(r.point_sources[ax], point_sources[ax].reshape(-1)[o.user_point_source_ids])


def test_link_point_sources_errors(actx):
    from boxtree_amd.tree import link_point_sources
    sources = normal(500, 2, 1)
    tree = device_tree(actx, None, sources, max_particles_in_box=10)
    with pytest.raises(ValueError):     # tree.py:800-801
        link_point_sources(actx, tree, actx.from_numpy(np.arange(501, dtype=np.int32)),
                           [actx.from_numpy(s) for s in sources])
