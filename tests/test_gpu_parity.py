"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs -- every Tree / FMMTraversalInfo array must be identical
-- plus the reference tests' invariants on the GPU output."""

import ctypes as ct

import numpy as np
import pytest

from compare import assert_same_traversal, assert_same_tree
from invariants import check_traversal, check_tree, constant_one_potentials

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def normal_particles(n, dims, dtype, seed=15):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(n, dtype=dtype) for _ in range(dims)]


def build_both(actx, oracle, particles, targets=None, trav_kw=None, **kw):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    dev = lambda arrs: None if arrs is None else [actx.from_numpy(a) for a in arrs]  # noqa: E731
    dkw = dict(kw)
    for name in ("source_radii", "target_radii", "refine_weights"):
        if dkw.get(name) is not None:
            dkw[name] = actx.from_numpy(dkw[name])
    tree, _ = TreeBuilder(actx)(actx, dev(particles), targets=dev(targets), **dkw)
    otree = oracle.build_tree(particles, targets=targets, **kw)
    htree = actx.to_numpy(tree)
    assert_same_tree(htree, otree)
    if trav_kw is None:
        return htree, otree, None, None
    tkw = dict(trav_kw)
    call_kw = {}
    if "_from_sep_smaller_min_nsources_cumul" in tkw:
        call_kw["_from_sep_smaller_min_nsources_cumul"] = tkw.pop(
            "_from_sep_smaller_min_nsources_cumul")
    otrav = oracle.build_traversal(otree, **trav_kw)
    # all device paths: walk-from-root kernels, parent-colleague kernels with float
    # predicates, and the default (the integer-lattice form where it applies)
    for force_generic in (True, "float", False):
        trav, _ = FMMTraversalBuilder(actx, **tkw)(actx, tree, _force_generic=force_generic,
                                                   **call_kw)
        htrav = actx.to_numpy(trav)
        assert_same_traversal(htrav, otrav)
    return htree, otree, htrav, otrav


# ---- primitives ---------------------------------------------------------------

@pytest.mark.parametrize("n", [0, 1, 63, 8192, 8193, 100003, 3 * 10**6])
@pytest.mark.parametrize("bits", [(0, 64), (0, 63), (5, 29)])
def test_radix_sort_u64(actx, n, bits):
    import torch
    rng = np.random.default_rng(n + bits[1])
    keys = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
    if n > 10:
        keys[: n // 3] = keys[0]        # heavy duplicates: stability matters
    vals = rng.integers(0, 2**31, size=n, dtype=np.int64).astype(np.uint32)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    dv = torch.from_numpy(vals.view(np.int32)).cuda()
    ok = torch.empty_like(dk)
    ov = torch.empty_like(dv)
    from boxtree_amd import _lib
    torch.cuda.synchronize()
    _lib.check(actx.lib.bt_radix_sort_u64_u32(
        actx.handle, ct.c_void_p(dk.data_ptr()), ct.c_void_p(dv.data_ptr()),
        ct.c_void_p(ok.data_ptr()), ct.c_void_p(ov.data_ptr()), n, bits[0], bits[1]))
    mask = np.uint64(((1 << (bits[1] - bits[0])) - 1) << bits[0]) if bits[1] - bits[0] < 64 \
        else np.uint64(2**64 - 1)
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(ok.cpu().numpy().view(np.uint64), keys[order])
    assert np.array_equal(ov.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("n", [1, 2, 63, 8192, 8193, 100003, 3 * 10**6])
@pytest.mark.parametrize("bits", [(17, 53), (22, 46), (5, 14), (27, 63), (0, 64), (3, 3)])
def test_radix_sort_u64_keys(actx, n, bits):
    """Keys-only sort (packed-key tree builds): stable on [begin, end), the low bits ride
    along; 8- and 9-bit digits (36 bits = 4 x 9, 24 = 3 x 8, 9 = 1 x 9)."""
    import torch
    rng = np.random.default_rng(n + 7 * bits[1])
    keys = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64)
    if n > 10:
        keys[: n // 3] &= np.uint64((1 << bits[0]) - 1)      # heavy duplicates of the sorted bits
        keys[n // 2:] |= np.uint64(1 << 63)
    dk = torch.from_numpy(keys.view(np.int64)).cuda()
    ok = torch.empty_like(dk)
    from boxtree_amd import _lib
    torch.cuda.synchronize()
    _lib.check(actx.lib.bt_radix_sort_u64_keys(
        actx.handle, ct.c_void_p(dk.data_ptr()), ct.c_void_p(ok.data_ptr()), n, bits[0], bits[1]))
    width = bits[1] - bits[0]
    mask = np.uint64((((1 << width) - 1) << bits[0]) & (2**64 - 1))
    order = np.argsort(keys & mask, kind="stable")
    assert np.array_equal(ok.cpu().numpy().view(np.uint64), keys[order])
    if n and width:
        st = _lib.SortStats()
        actx.lib.bt_get_sort_stats(actx.handle, st)
        assert st.bytes_per_element_per_pass == 16 and st.passes == min(-(-width // 8), -(-width // 9))


@pytest.mark.parametrize("mode", ["packed", "packed-shallow", "packed-rb8", "pairs"])
@pytest.mark.parametrize("dims,n,mpb", [(3, 60000, 30), (2, 40000, 5), (3, 7, 2), (3, 2, 1),
                                        (2, 100003, 64)])
def test_packed_key_build_modes(actx, mode, dims, n, mpb, monkeypatch):
    """Point-particle builds sort one word per particle (path bits over the id).  The same
    trees must come out when the packed path bits end above the tree's depth (the build
    returns to full keys there), with 8-bit digits, and with the (key, id) pair sort."""
    from oracle import oracle
    if mode == "packed-shallow":
        monkeypatch.setenv("BT_PACKED_LEVELS", "3")
    elif mode == "packed-rb8":
        monkeypatch.setenv("BT_SORT_KEYS_RB", "8")
    elif mode == "pairs":
        monkeypatch.setenv("BT_NO_PACKED_KEYS", "1")
    rng = np.random.default_rng(n + dims)
    pts = [rng.standard_normal(n) for _ in range(dims)]
    build_both(actx, oracle, pts, max_particles_in_box=mpb, trav_kw={})
    tg = [rng.standard_normal(n // 3 + 1) for _ in range(dims)]
    build_both(actx, oracle, pts, targets=tg, max_particles_in_box=mpb)
    # extents: the cap rides between the path bits and the id
    radii = 2.0 ** rng.uniform(-10, 0, len(tg[0])) * 0.1
    for norm in ("linf", "l2"):
        build_both(actx, oracle, pts, targets=tg, max_particles_in_box=mpb, target_radii=radii,
                   stick_out_factor=0.25, extent_norm=norm, trav_kw={} if norm == "linf" else None)
    if n > 1000:
        sr = 2.0 ** rng.uniform(-12, -3, n) * 0.05
        build_both(actx, oracle, pts, targets=tg, max_particles_in_box=mpb, source_radii=sr,
                   target_radii=radii, stick_out_factor=0.1)


@pytest.mark.parametrize("k1,k3,spill", [(2, 2, 1), (64, 3, 1), (64, 3, 0), (5, 64, 1)])
@pytest.mark.parametrize("dims,n,mpb", [(3, 50000, 20), (2, 30000, 6)])
def test_walk_rows_overflow(actx, k1, k3, spill, dims, n, mpb, monkeypatch):
    """Lists 1 and 3 (+ close) are written into fixed-capacity scratch rows by one walk; a
    list 3 that outgrows its row continues in a spill chunk (trees without extents) as long
    as chunks last, and an item whose lists still do not fit is walked a second time straight
    into the final lists.  With rows of a few entries nearly every item takes one of those
    routes (with the default 64 / 24 almost none does): same lists."""
    from oracle import oracle
    monkeypatch.setenv("BT_V2_K1", str(k1))
    monkeypatch.setenv("BT_V2_K3", str(k3))
    monkeypatch.setenv("BT_V2_SPILL", str(spill))
    rng = np.random.default_rng(1000 * k1 + k3 + dims)
    # clustered + uniform: leaves of several levels side by side (long lists 1, 3 and 4)
    pts = [np.concatenate([rng.random(n // 2), 0.3 + 0.02 * rng.standard_normal(n - n // 2)])
           for _ in range(dims)]
    build_both(actx, oracle, pts, max_particles_in_box=mpb, trav_kw={})
    tg = [rng.random(n // 4) for _ in range(dims)]
    radii = 2.0 ** rng.uniform(-10, 0, n // 4) * 0.05
    build_both(actx, oracle, pts, targets=tg, max_particles_in_box=mpb, target_radii=radii,
               stick_out_factor=0.25, trav_kw={})


@pytest.mark.parametrize("families", [1, 2])
@pytest.mark.parametrize("case", ["points", "separate_targets", "extents"])
def test_colleague_row_families(actx, families, case, monkeypatch):
    """3D trees of fewer than 2^25 boxes keep one family of colleague rows (the source flag in
    the entry, a mask per box of the entries that carry it), other trees a second family with the
    source colleagues; Lists 1 and 4 read either.  Same lists in both forms, where source boxes
    are few among the colleagues (separate targets, extents) and where they are all of them."""
    from oracle import oracle
    monkeypatch.setenv("BT_ROW_FAMILIES", str(families))
    rng = np.random.default_rng(77)
    n = 60000
    pts = [np.concatenate([rng.random(n // 2), 0.6 + 0.03 * rng.standard_normal(n - n // 2)])
           for _ in range(3)]
    if case == "points":
        build_both(actx, oracle, pts, max_particles_in_box=24, trav_kw={})
    elif case == "separate_targets":
        tg = [0.5 + 0.2 * rng.standard_normal(n // 3) for _ in range(3)]
        build_both(actx, oracle, pts, targets=tg, max_particles_in_box=24, trav_kw={})
    else:
        tg = [rng.random(n // 4) for _ in range(3)]
        radii = 2.0 ** rng.uniform(-10, 0, n // 4) * 0.04
        build_both(actx, oracle, pts, targets=tg, max_particles_in_box=24, target_radii=radii,
                   stick_out_factor=0.25, trav_kw={})


@pytest.mark.parametrize("kind", ["uniform", "surface", "blob"])
def test_depth_probe_large_build(actx, kind):
    """Builds of 2^20 particles and more look at a sample of the particles before they choose
    how many key bits to sort (depth probe): volume-filling, surface and clustered clouds,
    the last one deeper than the probe's estimate can cover (back to full keys mid-build)."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    n = (1 << 20) + 12345
    if kind == "uniform":
        pts = [rng.random(n) for _ in range(3)]
    elif kind == "surface":
        v = rng.standard_normal((3, n))
        v /= np.sqrt((v * v).sum(axis=0))
        pts = [np.ascontiguousarray(v[i]) for i in range(3)]
    else:
        pts = [np.concatenate([rng.random(n - 200000), 0.5 + 1e-7 * rng.standard_normal(200000)])
               for _ in range(3)]
    build_both(actx, oracle, pts, max_particles_in_box=32)
    tg = [rng.random(n // 8) for _ in range(3)]
    radii = 2.0 ** rng.uniform(-10, 0, n // 8) * 2.0 ** -7
    build_both(actx, oracle, pts, targets=tg, max_particles_in_box=32, target_radii=radii,
               stick_out_factor=0.25)


@pytest.mark.parametrize("n", [1, 777, 10**6])
def test_radix_sort_u32(actx, n):
    import torch
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 2**20, size=n, dtype=np.int64).astype(np.uint32)
    vals = np.arange(n, dtype=np.uint32)
    dk = torch.from_numpy(keys.view(np.int32)).cuda()
    dv = torch.from_numpy(vals.view(np.int32)).cuda()
    ok = torch.empty_like(dk)
    ov = torch.empty_like(dv)
    from boxtree_amd import _lib
    torch.cuda.synchronize()
    _lib.check(actx.lib.bt_radix_sort_u32_u32(
        actx.handle, ct.c_void_p(dk.data_ptr()), ct.c_void_p(dv.data_ptr()),
        ct.c_void_p(ok.data_ptr()), ct.c_void_p(ov.data_ptr()), n, 0, 20))
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(ok.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(ov.cpu().numpy().view(np.uint32), vals[order])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("n", [9, 4096, 10**5])
def test_bounding_box(actx, dtype, dims, n):
    # test/test_tree.py:50-79
    from boxtree_amd import BoundingBoxFinder
    p = normal_particles(n, dims, dtype)
    bbox, _ = BoundingBoxFinder(actx)(actx, [actx.from_numpy(x) for x in p], None)
    for i, ax in enumerate("xyz"[:dims]):
        assert bbox[f"min_{ax}"] == np.min(p[i])
        assert bbox[f"max_{ax}"] == np.max(p[i])


# ---- trees -----------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("n,mpb", [(4, 30), (50, 30), (1000, 5), (10**5, 5), (10**5, 30)])
def test_particle_tree(actx, oracle, dtype, dims, n, mpb):
    p = normal_particles(n, dims, dtype)
    htree, _, _, _ = build_both(actx, oracle, p, max_particles_in_box=mpb)
    check_tree(htree, p, max_particles_in_box=mpb)


@pytest.mark.parametrize("dims", [2, 3])
def test_explicit_refine_weights(actx, oracle, dims):
    n = 10**5
    p = normal_particles(n, dims, np.float64)
    rw = np.random.default_rng(10).integers(1, 10, (n,), dtype=np.int32)
    htree, _, _, _ = build_both(actx, oracle, p, refine_weights=rw,
                                max_leaf_refine_weight=100)
    check_tree(htree, p, refine_weights=rw, max_leaf_refine_weight=100)


@pytest.mark.parametrize("n_heavy", [0, 300])
def test_zero_weight_leaves(actx, oracle, n_heavy):
    # leaves far larger than max_leaf_refine_weight (zero-weight particles): runs of
    # >64 and >4096 ids exercise the workgroup sort and the global fix-up sort
    n = 20000
    p = normal_particles(n, 3, np.float64, seed=5)
    rw = np.zeros(n, np.int32)
    rw[np.random.default_rng(1).choice(n, n_heavy, replace=False)] = 1
    build_both(actx, oracle, p, refine_weights=rw, max_leaf_refine_weight=2, trav_kw={})


@pytest.mark.parametrize("dims", [2, 3])
def test_non_adaptive(actx, oracle, dims):
    p = normal_particles(10**4, dims, np.float64)
    build_both(actx, oracle, p, max_particles_in_box=30, kind="non-adaptive")


@pytest.mark.parametrize("dims", [2, 3])
def test_source_target_tree(actx, oracle, dims):
    s = normal_particles(2 * 10**5, dims, np.float64, seed=12)
    t = normal_particles(3 * 10**5, dims, np.float64, seed=19)
    htree, _, _, _ = build_both(actx, oracle, s, targets=t, max_particles_in_box=10)
    check_tree(htree, s, targets=t, max_particles_in_box=10)


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("extent_norm", ["linf", "l2"])
def test_extent_tree(actx, oracle, dims, extent_norm):
    ns, nt = 100000, 200000
    s = normal_particles(ns, dims, np.float64, seed=12)
    t = normal_particles(nt, dims, np.float64, seed=19)
    rw = np.zeros(ns + nt, np.int32)
    rw[:ns] = 1
    rng = np.random.default_rng(13)
    sr = 2**rng.uniform(-10, 0, (ns,))
    tr = 2**rng.uniform(-10, 0, (nt,))
    htree, _, _, _ = build_both(
        actx, oracle, s, targets=t, source_radii=sr, target_radii=tr,
        extent_norm=extent_norm, refine_weights=rw, max_leaf_refine_weight=20,
        stick_out_factor=0)
    check_tree(htree, s, targets=t, source_radii=sr, target_radii=tr,
               extent_norm=extent_norm)


def test_user_bbox(actx, oracle):
    p = [np.random.default_rng(3).random(20000) for _ in range(3)]
    bbox = np.array([[-0.5, 1.5]] * 3)
    build_both(actx, oracle, p, max_particles_in_box=20, bbox=bbox,
               trav_kw={})


def clustered_points(dims, n_bg, n_cl, scale, seed):
    """Uniform background plus a cluster of width `scale` around an interior point."""
    rng = np.random.default_rng(seed)
    bg = [rng.random(n_bg) for _ in range(dims)]
    c = rng.random(dims) * 0.5 + 0.25
    cl = [c[d] + scale * rng.random(n_cl) for d in range(dims)]
    return [np.concatenate([bg[d], cl[d]]) for d in range(dims)]


@pytest.mark.parametrize("log2_scale,n_cl", [(-20, 400), (-22, 3000), (-24, 800)])
def test_deep_tree_below_the_key(actx, oracle, log2_scale, n_cl):
    """3D trees deeper than the 21 levels of the 64-bit Morton key: the boxes of level
    21 that still have to split are re-keyed for the levels below (the reference
    keeps splitting, tree_build.py:622, 705-709)."""
    p = clustered_points(3, 20000, n_cl, 2.0 ** log2_scale, seed=7 - log2_scale)
    htree, otree, htrav, _ = build_both(actx, oracle, p, max_particles_in_box=8, trav_kw={})
    assert 23 <= htree.nlevels <= 30          # (level 30 and below: LAB_NOTES.md section 2, int-shift deviation)
    check_tree(htree, p, max_particles_in_box=8)
    check_traversal(htree, htrav)


def test_deep_tree_below_the_key_two_clusters_weights(actx, oracle):
    """Several re-keyed boxes at once, explicit refine weights (the weight prefix sums
    are rebuilt after the re-sort)."""
    rng = np.random.default_rng(5)
    a = clustered_points(3, 5000, 500, 2.0 ** -22, seed=1)
    b = clustered_points(3, 5000, 700, 2.0 ** -23, seed=2)
    p = [np.concatenate([a[d], b[d]]) for d in range(3)]
    rw = rng.integers(1, 5, len(p[0]), dtype=np.int32)
    htree, _, _, _ = build_both(actx, oracle, p, refine_weights=rw, max_leaf_refine_weight=20)
    assert htree.nlevels >= 24
    check_tree(htree, p, refine_weights=rw, max_leaf_refine_weight=20)


def test_deep_tree_below_the_key_with_extents(actx, oracle):
    """Target radii: the key has 19 levels, the continuation key carries the stop
    level of every re-keyed particle (tbk:388-428 evaluated below level 19)."""
    src = clustered_points(3, 8000, 600, 2.0 ** -22, seed=3)
    rng = np.random.default_rng(9)
    c = [float(np.median(src[d][-600:])) for d in range(3)]
    tgt = [np.concatenate([rng.random(2000), c[d] + 2.0 ** -22 * rng.random(300)])
           for d in range(3)]
    radii = np.concatenate([2.0 ** rng.uniform(-12, -6, 2000), 2.0 ** rng.uniform(-30, -20, 300)])
    htree, _, htrav, _ = build_both(actx, oracle, src, targets=tgt, target_radii=radii,
                                    stick_out_factor=0.25, max_particles_in_box=8, trav_kw={})
    assert htree.nlevels >= 21
    check_traversal(htree, htrav)


@pytest.mark.parametrize("log2_scale,n_cl,mpb", [(-20, 300, 8), (-22, 1500, 8), (-23, 600, 30)])
def test_deep_level_restricted_tree_below_the_key(actx, oracle, log2_scale, n_cl, mpb):
    """kind="adaptive-level-restricted" deeper than the 21 levels of the 64-bit key
    (tree_build.py:622 allows it: nlevels_max = 2 (nmant + 1) for every kind): when the level
    loop first needs a box below the key's reach, every non-empty box of level 21 is re-keyed --
    any leaf may be split later by the restriction -- and parents of level >= 21 are searched in
    the continuation key."""
    p = clustered_points(3, 6000, n_cl, 2.0 ** log2_scale, seed=11 - log2_scale)
    htree, otree, htrav, _ = build_both(actx, oracle, p, kind="adaptive-level-restricted",
                                        max_particles_in_box=mpb, trav_kw={})
    assert 23 <= htree.nlevels <= 30
    check_tree(htree, p, max_particles_in_box=mpb)
    check_traversal(htree, htrav)


def test_deep_level_restricted_tree_two_clusters_unpruned(actx, oracle):
    """Several re-keyed boxes, refine weights, and skip_prune (empty boxes are kept and may be
    force-split below the key as well)."""
    rng = np.random.default_rng(6)
    a = clustered_points(3, 3000, 400, 2.0 ** -22, seed=4)
    b = clustered_points(3, 3000, 500, 2.0 ** -23, seed=5)
    p = [np.concatenate([a[d], b[d]]) for d in range(3)]
    rw = rng.integers(1, 5, len(p[0]), dtype=np.int32)
    htree, _, _, _ = build_both(actx, oracle, p, kind="adaptive-level-restricted", refine_weights=rw,
                                max_leaf_refine_weight=20)
    assert htree.nlevels >= 24
    htree2, _, _, _ = build_both(actx, oracle, p, kind="adaptive-level-restricted",
                                 max_particles_in_box=10, skip_prune=True)
    assert htree2.nlevels >= 24


def test_max_levels_exceeded(actx):
    from boxtree_amd import MaxLevelsExceeded, TreeBuilder
    p = [np.zeros(100), np.zeros(100)]
    p[0][:50] = 1
    with pytest.raises(MaxLevelsExceeded):
        TreeBuilder(actx)(actx, [actx.from_numpy(x) for x in p], max_particles_in_box=10)


def test_argument_errors(actx):
    from boxtree_amd import TreeBuilder
    tb = TreeBuilder(actx)
    p = [actx.from_numpy(np.zeros(10)) for _ in range(2)]
    with pytest.raises(ValueError):
        tb(actx, p, kind="bogus", max_particles_in_box=3)
    with pytest.raises(ValueError):
        tb(actx, p)
    with pytest.raises(ValueError):
        tb(actx, p, max_particles_in_box=3, refine_weights=actx.from_numpy(
            np.ones(10, np.int32)), max_leaf_refine_weight=3)
    with pytest.raises(ValueError):
        tb(actx, p, max_particles_in_box=3, source_radii=actx.from_numpy(np.zeros(10)))
    with pytest.raises(TypeError):
        tb(actx, p, targets=p, max_particles_in_box=3, stick_out_factor=0,
           target_radii=actx.from_numpy(np.zeros(10, np.float32)))


def test_level_restriction_with_extents_refuses_orphaned_particles(actx):
    """LAB_NOTES.md section 2: upstream's algorithm can leave particles that no leaf owns
    when level restriction meets extents; such a tree is not handed out."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import fuzz_parity
    from boxtree_amd import TreeBuilder
    p, t, kw, _ = fuzz_parity.make_case(100310, lr_extents=True)
    assert kw["kind"] == "adaptive-level-restricted" and "target_radii" in kw
    dkw = dict(kw, target_radii=actx.from_numpy(kw["target_radii"]))
    with pytest.raises(RuntimeError, match="no leaf owns"):
        TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in p],
                          targets=[actx.from_numpy(a) for a in t], **dkw)


# ---- traversals ---------------------------------------------------------------------

@pytest.mark.parametrize("dims,sat", [(2, True), (2, False), (3, True), (3, False)])
def test_tree_connectivity(actx, oracle, dims, sat):
    s = normal_particles(10**5, dims, np.float64)
    t = None if sat else normal_particles(2 * 10**5, dims, np.float64)
    htree, _, htrav, _ = build_both(actx, oracle, s, targets=t,
                                    max_particles_in_box=30, trav_kw={})
    check_traversal(htree, htrav)


@pytest.mark.parametrize("well_sep_is_n_away", [1, 2])
@pytest.mark.parametrize("dims,ns,nt,ext,extent_norm,crit", [
    (2, 10**5, None, "", "linf", "static_linf"),
    (2, 5 * 10**4, 4 * 10**4, "", "linf", "static_linf"),
    (2, 10**5, 4 * 10**4, "t", "linf", "static_linf"),
    (3, 10**5, None, "", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "t", "linf", "static_linf"),
    (3, 10**5, 4 * 10**4, "t", "linf", "precise_linf"),
    (3, 10**5, 4 * 10**4, "t", "l2", "precise_linf"),
    (3, 10**5, 4 * 10**4, "t", "l2", "static_l2"),
])
def test_fmm_completeness(actx, oracle, dims, ns, nt, ext, extent_norm, crit,
                          well_sep_is_n_away):
    s = normal_particles(ns, dims, np.float64, seed=15)
    t = None if nt is None else normal_particles(nt, dims, np.float64, seed=16)
    rng = np.random.default_rng(12)
    tr = 2**rng.uniform(-10, 0, (nt,)) if "t" in ext else None
    htree, _, htrav, _ = build_both(
        actx, oracle, s, targets=t, max_particles_in_box=30, target_radii=tr,
        stick_out_factor=0.25, extent_norm=extent_norm,
        trav_kw=dict(well_sep_is_n_away=well_sep_is_n_away, from_sep_smaller_crit=crit))
    pot = constant_one_potentials(htree, htrav)
    assert np.all(pot == ns)


def test_from_sep_smaller_threshold(actx, oracle):
    # test/test_fmm.py:617-665 (source-count thresholding of list 3)
    s = normal_particles(5 * 10**4, 3, np.float64, seed=15)
    t = normal_particles(10**4, 3, np.float64, seed=16)
    tr = 2**np.random.default_rng(12).uniform(-10, 0, (10**4,))
    build_both(actx, oracle, s, targets=t, max_particles_in_box=30, target_radii=tr,
               stick_out_factor=0.25,
               trav_kw=dict(_from_sep_smaller_min_nsources_cumul=15))


def test_uniform_config_c2_small(actx, oracle):
    # BASELINE.json configs[1] recipe (3D uniform, mpb=64) at a size the oracle
    # finishes in seconds
    rng = np.random.default_rng(15)
    p = [rng.random(10**6) for _ in range(3)]
    htree, _, htrav, _ = build_both(actx, oracle, p, max_particles_in_box=64, trav_kw={})
    check_traversal(htree, htrav)


@pytest.mark.parametrize("sat", [True, False])
def test_one_dimensional(actx, oracle, sat):
    # test/test_fmm.py:146 exercises dims=1
    s = normal_particles(5 * 10**4, 1, np.float64, seed=15)
    t = None if sat else normal_particles(2 * 10**4, 1, np.float64, seed=16)
    htree, _, htrav, _ = build_both(actx, oracle, s, targets=t, max_particles_in_box=30,
                                    trav_kw={})
    pot = constant_one_potentials(htree, htrav)
    assert np.all(pot == 5 * 10**4)


def test_float32_traversal(actx, oracle):
    # test/test_fmm.py:672-719 (float32 coordinates)
    s = normal_particles(10**5, 3, np.float32, seed=15)
    htree, _, htrav, _ = build_both(actx, oracle, s, max_particles_in_box=30, trav_kw={})
    pot = constant_one_potentials(htree, htrav)
    assert np.all(pot == 10**5)


def test_full_size_properties_c2(actx):
    """BASELINE configs[1] at full size (3D uniform 1e7, mpb=64): size-independent
    properties instead of the oracle (which would need minutes)."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    n = 10**7
    g = torch.Generator(device="cuda")
    g.manual_seed(15)
    pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
    tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=64)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)

    usi = tree.user_source_ids.long()
    # a permutation, and its inverse
    assert int(torch.bincount(usi, minlength=n).max()) == 1
    assert bool((usi[tree.sorted_target_ids.long()] == torch.arange(n, device="cuda")).all())
    # coordinates are the gathered inputs
    for d in range(3):
        assert bool((tree.sources[d] == pts[d][usi]).all())
    nb = tree.nboxes
    starts = tree.box_source_starts.long()
    cumul = tree.box_source_counts_cumul.long()
    nonchild = tree.box_source_counts_nonchild.long()
    child = tree.box_child_ids[:, :nb].long()
    kid_sum = torch.where(child != 0, cumul[child], torch.zeros_like(child)).sum(dim=0)
    assert bool((nonchild + kid_sum == cumul).all())
    assert int(cumul[0]) == n and int(nonchild.sum()) == n
    leaf = (child == 0).all(dim=0)
    assert int(cumul[leaf].max()) <= 64
    assert bool((cumul[~leaf] > 64).all())
    # inside a leaf the particles are in ascending user order (stable renumbering)
    owner = torch.repeat_interleave(torch.arange(nb, device="cuda")[leaf], cumul[leaf])
    order = torch.argsort(starts[leaf])
    owner_sorted = torch.repeat_interleave(
        torch.arange(nb, device="cuda")[leaf][order], cumul[leaf][order])
    same_leaf = owner_sorted[1:] == owner_sorted[:-1]
    assert bool((usi[1:][same_leaf] > usi[:-1][same_leaf]).all())
    del owner
    # every particle lies inside its leaf box
    lev = tree.box_levels.long()
    half = 0.5 * float(tree.root_extent) / (2.0 ** lev.double())
    ctr = tree.box_centers[:, :nb]
    for d in range(3):
        x = tree.sources[d]
        lo = (ctr[d] - half)[owner_sorted]
        hi = (ctr[d] + half)[owner_sorted]
        assert bool(((x >= lo - 1e-12) & (x < hi + 1e-12)).all())
    # constant-one completeness on the device lists
    htree = actx.to_numpy(tree)
    htrav = actx.to_numpy(trav)
    pot = constant_one_potentials(htree, htrav)
    assert np.all(pot == n)


# ---- multi-GPU exchange kernels (single GPU: routing only, no collective) ----------

@pytest.mark.parametrize("dims,level", [(3, 5), (2, 7)])
def test_shard_kernels_match_torch_routing(actx, dims, level):
    import torch
    from boxtree_amd import _lib
    from boxtree_amd.distributed import morton_cells, partition_cells
    n = 300001
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    pts = [torch.randn(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(dims)]
    bmin = np.array([float(p.min()) for p in pts])
    ext = max(float(p.max()) - float(p.min()) for p in pts) * (1 + 1e-4)
    bmax = bmin + ext
    ncells = 1 << (dims * level)
    ref_cells = morton_cells(pts, bmin, bmax, level)

    cells = torch.empty(n, dtype=torch.int32, device="cuda")
    hist = torch.zeros(ncells, dtype=torch.int32, device="cuda")
    ptrs = (ct.c_void_p * dims)(*[p.data_ptr() for p in pts])
    cmin = (ct.c_double * dims)(*bmin.tolist())
    cmax = (ct.c_double * dims)(*bmax.tolist())
    torch.cuda.synchronize()
    _lib.check(actx.lib.bt_morton_cells(actx.handle, dims, _lib.BT_F64, ptrs, n, cmin, cmax, level,
                                        ct.c_void_p(cells.data_ptr()), ct.c_void_p(hist.data_ptr())))
    assert bool((cells.long() == ref_cells).all())
    assert bool((hist.long() == torch.bincount(ref_cells, minlength=ncells)).all())

    world = 8
    owner = partition_cells(hist.cpu().numpy(), world)
    owner_t = torch.from_numpy(owner).cuda()
    owner32 = owner_t.to(torch.int32)
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    _lib.check(actx.lib.bt_bucket_permutation(actx.handle, ct.c_void_p(cells.data_ptr()), n,
                                              ct.c_void_p(owner32.data_ptr()), world,
                                              ct.c_void_p(perm.data_ptr())))
    ref_perm = torch.argsort(owner_t[ref_cells], stable=True)
    assert bool((perm.long() == ref_perm).all())

    out = torch.empty_like(pts[0])
    _lib.check(actx.lib.bt_gather(actx.handle, 8, ct.c_void_p(pts[0].data_ptr()),
                                  ct.c_void_p(perm.data_ptr()), n, ct.c_void_p(out.data_ptr())))
    assert bool((out == pts[0][ref_perm]).all())


def test_merge_close_lists(actx, oracle):
    # traversal.py:1650-1693; consumer: test_fmm.py:231-233
    s = normal_particles(3 * 10**4, 3, np.float64, seed=15)
    t = normal_particles(10**4, 3, np.float64, seed=16)
    tr = 2**np.random.default_rng(12).uniform(-10, 0, (10**4,))
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(a) for a in s],
                                targets=[actx.from_numpy(a) for a in t],
                                target_radii=actx.from_numpy(tr), stick_out_factor=0.25,
                                max_particles_in_box=30)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    merged = actx.to_numpy(trav.merge_close_lists(actx))
    h = actx.to_numpy(trav)
    assert merged.from_sep_close_smaller_starts is None
    n = len(h.target_boxes)
    for i in range(0, n, max(1, n // 500)):
        exp = np.concatenate([
            h.neighbor_source_boxes_lists[h.neighbor_source_boxes_starts[i]:
                                          h.neighbor_source_boxes_starts[i + 1]],
            h.from_sep_close_smaller_lists[h.from_sep_close_smaller_starts[i]:
                                           h.from_sep_close_smaller_starts[i + 1]],
            h.from_sep_close_bigger_lists[h.from_sep_close_bigger_starts[i]:
                                          h.from_sep_close_bigger_starts[i + 1]]])
        got = merged.neighbor_source_boxes_lists[merged.neighbor_source_boxes_starts[i]:
                                                 merged.neighbor_source_boxes_starts[i + 1]]
        assert np.array_equal(exp, got)
    pot = constant_one_potentials(actx.to_numpy(tree), merged)
    assert np.all(pot == 3 * 10**4)


def test_sharded_build_union_equals_global_tree(actx):
    """What the N-GPU build relies on (distributed.py step 5): with the global root
    box, the trees built per owner over contiguous Morton ranges of heavy top-level
    cells are exactly the global tree restricted to those cells."""
    import torch
    from boxtree_amd import TreeBuilder
    from boxtree_amd.distributed import ROOT_EXTENT_STRETCH_FACTOR, morton_cells, partition_cells
    rng = np.random.default_rng(7)
    n, world, level = 400000, 4, 3
    pts = [torch.from_numpy(rng.random(n)).cuda() for _ in range(3)]
    gmin = np.array([float(p.min()) for p in pts])
    gmax = np.array([float(p.max()) for p in pts])
    root_extent = max(gmax - gmin) * (1 + ROOT_EXTENT_STRETCH_FACTOR)
    bbox_min = gmin.copy()
    bbox_max = bbox_min + root_extent
    cells = morton_cells(pts, bbox_min, bbox_max, level)
    hist = torch.bincount(cells, minlength=1 << (3 * level)).cpu().numpy()
    owner = torch.from_numpy(partition_cells(hist, world)).cuda()[cells]
    tb = TreeBuilder(actx)

    def leaves(tree):
        t = actx.to_numpy(tree)
        nb = t.nboxes
        leaf = (t.box_child_ids[:, :nb] == 0).all(axis=0)
        rows = np.concatenate([t.box_levels[leaf][None, :].astype(np.float64),
                               t.box_centers[:, :nb][:, leaf],
                               t.box_source_counts_cumul[leaf][None, :].astype(np.float64)])
        rows = rows.T
        return rows[np.lexsort(rows.T[::-1])]

    gtree, _ = tb(actx, pts, max_particles_in_box=30)
    parts = []
    for r in range(world):
        m = owner == r
        sub = [p[m].contiguous() for p in pts]
        tree, _ = tb(actx, sub, max_particles_in_box=30,
                     _root_box=(bbox_min, bbox_max, root_extent))
        parts.append(leaves(tree))
    union = np.concatenate(parts)
    union = union[np.lexsort(union.T[::-1])]
    assert np.array_equal(union, leaves(gtree))


@pytest.mark.gpu
@pytest.mark.parametrize("dist_kind", ["normal", "uniform", "clustered"])
@pytest.mark.parametrize("dims", [2, 3])
def test_sharded_build_global_numbering(actx, dims, dist_kind):
    """distributed.py steps 3-5 without a process group: the ranks' trees, built
    from their shares with the global root box and top-tree counts and renumbered by
    global_box_numbering, are slices of the tree one GPU builds from all points --
    same box numbers, same particle order."""
    import torch
    from boxtree_amd import TreeBuilder
    from boxtree_amd.distributed import (ROOT_EXTENT_STRETCH_FACTOR, global_box_numbering,
                                         local_to_global_box_ids, morton_cells,
                                         partition_cells, top_tree_plan)
    rng = np.random.default_rng(11)
    n, world, level, mpb = 300000, 4, 3 if dims == 3 else 4, 30
    if dist_kind == "normal":
        host = [rng.standard_normal(n) for _ in range(dims)]
    elif dist_kind == "uniform":
        host = [rng.random(n) for _ in range(dims)]
    else:       # a dense blob plus a thin background: light and empty top cells
        host = [np.concatenate([0.02 * rng.standard_normal(n - 500) + 0.7, rng.random(500)])
                for _ in range(dims)]
    pts = [torch.from_numpy(h).cuda() for h in host]
    gmin = np.array([float(p.min()) for p in pts])
    gmax = np.array([float(p.max()) for p in pts])
    root_extent = max(gmax - gmin) * (1 + ROOT_EXTENT_STRETCH_FACTOR)
    bbox_min = gmin.copy()
    bbox_max = bbox_min + root_extent
    cells = morton_cells(pts, bbox_min, bbox_max, level)
    C = 1 << dims
    hist = torch.bincount(cells, minlength=C ** level).cpu().numpy()
    plan = top_tree_plan(hist, dims, level, mpb)
    owner_of_cell = partition_cells(hist, world, plan["unit_start"])
    assert np.all(np.diff(owner_of_cell) >= 0)
    owner = torch.from_numpy(owner_of_cell).cuda()[cells]
    prefix = torch.from_numpy(plan["cell_prefix"]).cuda()
    tb = TreeBuilder(actx)
    g = actx.to_numpy(tb(actx, pts, max_particles_in_box=mpb)[0])

    locs, gids = [], []
    for r in range(world):
        sel = torch.nonzero(owner == r).flatten()
        gids.append(sel.cpu().numpy())
        if len(sel) == 0:           # one cell can outweigh a whole rank's share
            locs.append(None)
            continue
        sub = [p[sel].contiguous() for p in pts]
        tree, _ = tb(actx, sub, max_particles_in_box=mpb,
                     _root_box=(bbox_min, bbox_max, root_extent), _top_tree=(level, prefix))
        locs.append(tree)
    nmax = 64
    lc = np.zeros((world, nmax), np.int64)
    for r, t in enumerate(locs):
        if t is None:
            continue
        d = np.diff(actx.to_numpy(t.level_start_box_nrs))
        lc[r, :len(d)] = d
    hits = np.zeros(g.nboxes, np.int64)
    cumul = np.zeros(g.nboxes, np.int64)
    src_off = 0
    for r, t in enumerate(locs):
        if t is None:
            continue
        starts, deep = global_box_numbering(plan, lc, r)
        assert np.array_equal(starts, g.level_start_box_nrs)
        m = local_to_global_box_ids(t, plan, starts, deep, bbox_min, root_extent).cpu().numpy()
        h = actx.to_numpy(t)
        nb = h.nboxes
        assert len(set(m.tolist())) == nb
        hits[m] += 1
        assert np.array_equal(g.box_levels[m], h.box_levels)
        assert np.array_equal(g.box_centers[:, m], h.box_centers[:, :nb])
        assert np.array_equal(g.box_parent_ids[m], m[h.box_parent_ids])
        ch = h.box_child_ids[:, :nb]
        mapped = np.where(ch != 0, m[ch], 0)
        gch = g.box_child_ids[:, m]
        deep_box = h.box_levels > level
        assert np.array_equal(mapped[:, deep_box], gch[:, deep_box])
        assert np.all((mapped == 0) | (mapped == gch))          # shared top boxes: a subset
        cumul[m] += h.box_source_counts_cumul
        for name in ("box_source_counts_cumul", "box_source_counts_nonchild", "box_flags"):
            assert np.array_equal(getattr(g, name)[m][deep_box], getattr(h, name)[deep_box])
        assert np.array_equal(g.box_source_starts[m][deep_box],
                              h.box_source_starts[deep_box] + src_off)
        # a shared top box starts where its first owner's share starts
        first = hits[m] == 1
        top_first = first & ~deep_box
        assert np.array_equal(g.box_source_starts[m][top_first],
                              h.box_source_starts[top_first] + src_off)
        # particle order: the rank's slice of the global tree order
        ns = h.nsources
        assert np.array_equal(g.user_source_ids[src_off:src_off + ns], gids[r][h.user_source_ids])
        for ax in range(dims):
            assert np.array_equal(g.sources[ax][src_off:src_off + ns], h.sources[ax])
        src_off += ns
    assert src_off == n
    assert np.all(hits >= 1)
    assert np.all(hits[g.box_levels > level] == 1)
    assert np.array_equal(cumul, g.box_source_counts_cumul)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,dist_kind", [(3, "uniform"), (3, "clustered"), (2, "normal")])
def test_sharded_traversal_is_a_slice_of_the_global_one(actx, dims, dist_kind):
    """distributed.py step 6: the lists built for one rank's boxes (all shared top
    boxes + its own subtrees) on the complete box arrays are the rows of the
    single-GPU traversal for those boxes."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    from boxtree_amd.distributed import (ROOT_EXTENT_STRETCH_FACTOR, active_boxes,
                                         global_box_numbering, morton_cells, partition_cells,
                                         top_tree_plan)
    rng = np.random.default_rng(12)
    n, world, level, mpb = 200000, 3, 3 if dims == 3 else 4, 30
    if dist_kind == "normal":
        host = [rng.standard_normal(n) for _ in range(dims)]
    elif dist_kind == "uniform":
        host = [rng.random(n) for _ in range(dims)]
    else:
        host = [np.concatenate([0.03 * rng.standard_normal(n - 500) + 0.6, rng.random(500)])
                for _ in range(dims)]
    pts = [torch.from_numpy(h).cuda() for h in host]
    gmin = np.array([float(p.min()) for p in pts])
    gmax = np.array([float(p.max()) for p in pts])
    root_extent = max(gmax - gmin) * (1 + ROOT_EXTENT_STRETCH_FACTOR)
    bbox_min = gmin.copy()
    cells = morton_cells(pts, bbox_min, bbox_min + root_extent, level).cpu().numpy()
    C = 1 << dims
    hist = np.bincount(cells, minlength=C ** level)
    plan = top_tree_plan(hist, dims, level, mpb)
    owner_of_cell = partition_cells(hist, world, plan["unit_start"])
    tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=mpb)
    g = actx.to_numpy(tree)
    full = actx.to_numpy(FMMTraversalBuilder(actx)(actx, tree)[0])

    # boxes per level and rank, as the ranks' local trees would report them: a box
    # below the top levels belongs to the owner of its level-`level` cell
    lvl = g.box_levels.astype(np.int64)
    nb = g.nboxes
    path = np.zeros(nb, np.int64)
    for ax in range(dims):
        v = np.floor((g.box_centers[ax, :nb] - bbox_min[ax]) / root_extent * 2.0 ** lvl).astype(np.int64)
        for b in range(int(lvl.max()) + 1):
            path |= ((v >> b) & 1) << (dims * b + (dims - 1 - ax))
    deep = lvl > level
    cell_of_box = path >> np.where(deep, dims * (lvl - level), 0)
    box_owner = np.where(deep, owner_of_cell[np.where(deep, cell_of_box, 0)], -1)
    lc = np.zeros((world, 64), np.int64)
    for r in range(world):
        lc[r, :level + 1] = 1
        for lev in range(level + 1, g.nlevels):
            lc[r, lev] = int(np.sum((lvl == lev) & (box_owner == r)))

    def rows(starts, lists, sel):
        return [lists[starts[i]:starts[i + 1]].tolist() for i in sel]

    for r in range(world):
        starts, deep_base = global_box_numbering(plan, lc, r)
        assert np.array_equal(starts, g.level_start_box_nrs)
        mask, ranges = active_boxes(plan, starts, deep_base, lc[r], "cuda")
        hm = mask.cpu().numpy().astype(bool)
        assert np.array_equal(hm, (~deep) | (box_owner == r))
        t = actx.to_numpy(FMMTraversalBuilder(actx)(
            actx, tree, _target_boxes_mask=mask, _active_level_ranges=ranges)[0])
        sel_t = np.nonzero(hm[full.target_boxes])[0]
        sel_p = np.nonzero(hm[full.target_or_target_parent_boxes])[0]
        assert np.array_equal(t.target_boxes, full.target_boxes[sel_t])
        assert np.array_equal(t.target_or_target_parent_boxes,
                              full.target_or_target_parent_boxes[sel_p])
        assert np.array_equal(t.source_boxes, full.source_boxes)
        act = np.nonzero(hm)[0]
        assert rows(t.same_level_non_well_sep_boxes_starts, t.same_level_non_well_sep_boxes_lists,
                    act) == rows(full.same_level_non_well_sep_boxes_starts,
                                 full.same_level_non_well_sep_boxes_lists, act)
        assert np.all(np.diff(t.same_level_non_well_sep_boxes_starts) >= 0)
        for name, sel in (("neighbor_source_boxes", sel_t), ("from_sep_siblings", sel_p),
                          ("from_sep_bigger", sel_p)):
            got = rows(getattr(t, name + "_starts"), getattr(t, name + "_lists"),
                       range(len(sel)))
            want = rows(getattr(full, name + "_starts"), getattr(full, name + "_lists"), sel)
            assert got == want, name
        for lev in range(g.nlevels):
            a, b = t.from_sep_smaller_by_level[lev], full.from_sep_smaller_by_level[lev]
            got = {int(tb): a.lists[a.starts[i]:a.starts[i + 1]].tolist()
                   for i, tb in enumerate(t.target_boxes_sep_smaller_by_source_level[lev])}
            want = {int(tb): b.lists[b.starts[i]:b.starts[i + 1]].tolist()
                    for i, tb in enumerate(full.target_boxes_sep_smaller_by_source_level[lev])
                    if hm[tb]}
            assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [1, 2, 3])
@pytest.mark.parametrize("kind", ["adaptive", "non-adaptive"])
def test_skip_prune(actx, oracle, dims, kind):
    """Unpruned trees (tree_build.py:1328-1332: skip_prune keeps the empty
    children of every split box)."""
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    p = normal_particles(20000, dims, np.float64, seed=3)
    htree, otree, _, _ = build_both(actx, oracle, p, max_particles_in_box=25, kind=kind,
                                    skip_prune=True)
    assert not htree._is_pruned
    assert np.any(htree.box_source_counts_cumul == 0) or dims == 1
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(x) for x in p],
                                max_particles_in_box=25, skip_prune=True)
    with pytest.raises(ValueError):         # traversal.py:1999-2000
        FMMTraversalBuilder(actx)(actx, tree)


@pytest.mark.gpu
def test_skip_prune_source_target_extents(actx, oracle):
    s = normal_particles(20000, 3, np.float64, seed=12)
    t = normal_particles(30000, 3, np.float64, seed=19)
    tr = 2 ** np.random.default_rng(13).uniform(-10, 0, 30000)
    build_both(actx, oracle, s, targets=t, target_radii=tr, stick_out_factor=0.25,
               max_particles_in_box=20, skip_prune=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,spread", [(3, 1e-5), (2, 1e-7), (3, 1e-2)])
def test_deep_tree_beyond_the_presorted_bits(actx, oracle, dims, spread):
    """Trees deeper than the 40 key bits sorted up front (13 levels in 3D, 20 in
    2D): the level loop sorts the remaining bits on demand."""
    rng = np.random.default_rng(8)
    n = 40000
    p = [np.concatenate([rng.random(n // 2), 0.3 + spread * rng.standard_normal(n // 2)])
         for _ in range(dims)]
    htree, otree, _, _ = build_both(actx, oracle, p, max_particles_in_box=8, trav_kw={})
    if spread < 1e-3:
        assert htree.nlevels > (14 if dims == 3 else 21)
    check_tree(htree, p, max_particles_in_box=8)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind", [(3, 3, "sphere"), (3, 4, "uniform"),
                                                  (2, 2, "normal")])
def test_multi_rank_pipeline_in_process(dims, world, dist_kind):
    """The whole N-rank path of distributed.py / bench.py (global bbox, cell
    histogram, exchange, per-rank build, numbering, box all-gather, per-rank lists)
    with the ranks as threads on one GPU and an in-process torch.distributed
    stand-in: every rank ends up with the box arrays of the tree one rank builds
    from all points, and its lists are the rows of the global traversal."""
    import threading

    import torch
    from fake_dist import FakeWorld
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import (exchange_particles, gather_global_box_tree,
                                         number_sharded_tree)
    n_per, mpb = 60000, 30
    top_level = 3 if dims == 3 else 4

    def chunk(rank):
        rng = np.random.default_rng(100 + rank)
        if dist_kind == "sphere":
            v = rng.standard_normal((dims, n_per))
            v /= np.sqrt((v * v).sum(axis=0))
            return [np.ascontiguousarray(v[i]) for i in range(dims)]
        if dist_kind == "uniform":
            return [rng.random(n_per) for _ in range(dims)]
        return [rng.standard_normal(n_per) for _ in range(dims)]

    chunks = [chunk(r) for r in range(world)]
    fw = FakeWorld(world)
    results = [None] * world
    errors = []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            dist = fw.rank_view(rank)
            pts = [torch.from_numpy(a).cuda() for a in chunks[rank]]
            p2, _, kw, stats = exchange_particles(actx, dist, pts, None, {}, top_level=top_level,
                                                  max_particles_in_box=mpb)
            tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
            num = number_sharded_tree(dist, tree, stats)
            gtree = gather_global_box_tree(actx, dist, tree, num)
            trav, _ = FMMTraversalBuilder(actx)(
                actx, gtree, _target_boxes_mask=num["target_boxes_mask"],
                _active_level_ranges=num["active_level_ranges"])
            results[rank] = dict(num=num, gtree=actx.to_numpy(gtree), trav=actx.to_numpy(trav),
                                 nlocal=int(tree.nsources),
                                 mask=num["target_boxes_mask"].cpu().numpy().astype(bool))
        except BaseException as e:      # noqa: BLE001
            errors.append((rank, repr(e)))
            try:
                fw.barrier.abort()
            except Exception:           # noqa: BLE001
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors

    # the single-rank reference over the concatenated input
    actx = HIPArrayContext(0)
    allpts = [torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()
              for ax in range(dims)]
    gt, _ = TreeBuilder(actx)(actx, allpts, max_particles_in_box=mpb)
    full = actx.to_numpy(FMMTraversalBuilder(actx)(actx, gt)[0])
    g = actx.to_numpy(gt)
    assert sum(r["nlocal"] for r in results) == world * n_per

    def rows(starts, lists, sel):
        return [lists[starts[i]:starts[i + 1]].tolist() for i in sel]

    covered = np.zeros(g.nboxes, bool)
    for r in results:
        t, tr, hm = r["gtree"], r["trav"], r["mask"]
        assert r["num"]["nboxes"] == g.nboxes
        assert np.array_equal(t.level_start_box_nrs, g.level_start_box_nrs)
        for name in ("box_centers", "box_levels", "box_parent_ids", "box_child_ids", "box_flags"):
            assert np.array_equal(getattr(t, name), getattr(g, name)), name
        covered |= hm
        sel_t = np.nonzero(hm[full.target_boxes])[0]
        sel_p = np.nonzero(hm[full.target_or_target_parent_boxes])[0]
        assert np.array_equal(tr.target_boxes, full.target_boxes[sel_t])
        for name, sel in (("neighbor_source_boxes", sel_t), ("from_sep_siblings", sel_p),
                          ("from_sep_bigger", sel_p)):
            got = rows(getattr(tr, name + "_starts"), getattr(tr, name + "_lists"),
                       range(len(sel)))
            want = rows(getattr(full, name + "_starts"), getattr(full, name + "_lists"), sel)
            assert got == want, name
        for lev in range(g.nlevels):
            a, b = tr.from_sep_smaller_by_level[lev], full.from_sep_smaller_by_level[lev]
            got = {int(tb): a.lists[a.starts[i]:a.starts[i + 1]].tolist()
                   for i, tb in enumerate(tr.target_boxes_sep_smaller_by_source_level[lev])}
            want = {int(tb): b.lists[b.starts[i]:b.starts[i + 1]].tolist()
                    for i, tb in enumerate(full.target_boxes_sep_smaller_by_source_level[lev])
                    if hm[tb]}
            assert got == want
    assert covered.all()


@pytest.mark.gpu
def test_cuda_array_interface_inputs(actx, oracle):
    """SURVEY 8b input layout: any object exposing __cuda_array_interface__ is
    accepted as a coordinate array (wrapped without a copy)."""
    from boxtree_amd import TreeBuilder

    class Foreign:
        def __init__(self, t):
            self._t = t
            self.__cuda_array_interface__ = t.__cuda_array_interface__

        def __len__(self):
            return len(self._t)

    p = normal_particles(5000, 3, np.float64, seed=2)
    dev = [actx.from_numpy(x) for x in p]
    tree, _ = TreeBuilder(actx)(actx, [Foreign(t) for t in dev], max_particles_in_box=30)
    assert_same_tree(actx.to_numpy(tree), oracle.build_tree(p, max_particles_in_box=30))


@pytest.mark.gpu
@pytest.mark.parametrize("dims,dtype,world,n", [(3, np.float64, 8, 300001), (2, np.float32, 3, 50000),
                                                (1, np.float64, 2, 777), (3, np.float64, 1, 4096),
                                                (3, np.float32, 256, 200000)])
def test_partition_pack_equals_permutation_and_gather(dims, dtype, world, n):
    """bt_partition_pack (one sweep: count per wave and owner, scan, scatter of the
    interleaved coordinates) writes the records bt_bucket_permutation + bt_gather_pack write,
    the own segment straight into the receive buffer at the given offset."""
    import ctypes as ct

    import torch
    from boxtree_amd import HIPArrayContext, _lib
    actx = HIPArrayContext(0)
    rng = np.random.default_rng(n + world)
    pts = [actx.from_numpy(rng.random(n).astype(dtype)) for _ in range(dims)]
    ncells = 4096
    cells = actx.from_numpy(rng.integers(0, ncells, n).astype(np.int32))
    owner_np = np.sort(rng.integers(0, world, ncells)).astype(np.int32)
    owner = actx.from_numpy(owner_np)
    counts = np.bincount(owner_np[actx.to_numpy(cells)], minlength=world)
    me = int(rng.integers(0, world))
    s_off = np.concatenate([[0], np.cumsum(counts)])
    es = np.dtype(dtype).itemsize
    ptrs = (ct.c_void_p * dims)(*[p.data_ptr() for p in pts])
    tdt = pts[0].dtype
    perm = torch.empty(n, dtype=torch.int32, device="cuda")
    want = torch.empty(dims * n, dtype=tdt, device="cuda")
    _lib.check(actx.lib.bt_bucket_permutation(actx.handle, ct.c_void_p(cells.data_ptr()), n,
                                              ct.c_void_p(owner.data_ptr()), world,
                                              ct.c_void_p(perm.data_ptr())))
    _lib.check(actx.lib.bt_gather_pack(actx.handle, dims, es, ptrs, ct.c_void_p(perm.data_ptr()), n,
                                       ct.c_void_p(want.data_ptr())))
    pad = 17                                               # the own segment lands behind `pad` records
    send = torch.full((dims * n,), -1, dtype=tdt, device="cuda")
    recv = torch.full((dims * (pad + int(counts[me])),), -1, dtype=tdt, device="cuda")
    _lib.check(actx.lib.bt_partition_pack(
        actx.handle, dims, es, ptrs, ct.c_void_p(cells.data_ptr()), n, ct.c_void_p(owner.data_ptr()),
        world, me, int(s_off[me]), pad, ct.c_void_p(send.data_ptr()), ct.c_void_p(recv.data_ptr())))
    actx.synchronize()
    lo, hi = dims * int(s_off[me]), dims * int(s_off[me + 1])
    assert torch.equal(send[:lo], want[:lo]) and torch.equal(send[hi:], want[hi:])
    assert torch.equal(recv[dims * pad:], want[lo:hi])
    assert bool((recv[:dims * pad] == -1).all()) and bool((send[lo:hi] == -1).all())


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 5])
def test_all_to_all_chunked_on_views(world):
    """The RCCL path of all_to_all_chunked -- views of the send and receive buffers handed
    to the list form of the collective, a rank's own segment already in place -- against
    the plain single-buffer exchange, with several rounds (ranks as threads on one GPU;
    tests/fake_dist.py plays the backend)."""
    import threading

    import torch
    from boxtree_amd.distributed import all_to_all_chunked
    from fake_dist import FakeWorld
    rng = np.random.default_rng(world)
    splits = rng.integers(1500, 3000, size=(world, world))
    splits[0, world - 1] = 0                            # an empty message
    sends = [torch.from_numpy(rng.random(int(splits[r].sum()))).cuda() for r in range(world)]
    results = {}

    def run(backend, self_in_place):
        fw = FakeWorld(world, backend=backend)
        out = [None] * world
        errs = []

        def worker(rank):
            try:
                d = fw.rank_view(rank)
                s_split = [int(c) for c in splits[rank]]
                r_split = [int(splits[src][rank]) for src in range(world)]
                recv = torch.full((sum(r_split),), -1.0, dtype=torch.float64, device="cuda")
                if self_in_place:
                    s0 = sum(s_split[:rank])
                    r0 = sum(r_split[:rank])
                    recv[r0:r0 + r_split[rank]] = sends[rank][s0:s0 + s_split[rank]]
                rounds = all_to_all_chunked(d, recv, sends[rank], r_split, s_split,
                                            limit_bytes=8 * 700, self_in_place=self_in_place)
                out[rank] = (rounds, recv.cpu())
            except Exception as exc:          # noqa: BLE001
                errs.append(exc)
                fw.barrier.abort()
        threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        return out

    plain = run(None, False)
    views = run("nccl", True)
    assert max(v[0] for v in views) > 1           # several rounds were needed
    for rank in range(world):
        assert views[rank][0] == plain[rank][0]
        assert torch.equal(views[rank][1], plain[rank][1])
        assert not bool((views[rank][1] < 0).any())


@pytest.mark.gpu
def test_exchange_large_messages_nccl_single_rank():
    """Regression: a 1.44 GB all_to_all_single message (6*10^7 packed 3D points, one
    rank sending to itself over RCCL) arrived with its second half corrupted; the
    exchange now moves such buffers in rounds of at most 512 MiB per message."""
    import os

    import torch
    import torch.distributed as dist
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.distributed import exchange_particles
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        actx = HIPArrayContext(0)
        n = 6 * 10**7
        g = torch.Generator(device="cuda")
        g.manual_seed(3)
        pts = [torch.rand(n, generator=g, dtype=torch.float64, device="cuda") for _ in range(3)]
        p2, _, kw, st = exchange_particles(actx, dist, pts, None, {}, max_particles_in_box=64)
        torch.cuda.synchronize()
        assert kw.get("_point_stride") == 3
        for a, b in zip(pts, p2):
            assert bool(torch.equal(a, b))          # one rank: nothing moves
        # the list form of the collective as all_to_all_chunked calls it, on the real backend
        src = torch.arange(1000, dtype=torch.float64, device="cuda")
        dst = torch.zeros(1000, dtype=torch.float64, device="cuda")
        dist.all_to_all([dst[100:600]], [src[200:700]])
        dist.all_to_all([dst[:0]], [src[:0]])        # an empty segment
        torch.cuda.synchronize()
        assert bool(torch.equal(dst[100:600], src[200:700])) and float(dst[:100].sum()) == 0.0
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind,nway", [(3, 3, "sphere", 1), (3, 4, "uniform", 1),
                                                       (2, 2, "normal", 1), (2, 4, "uniform", 2),
                                                       (3, 8, "sphere", 1)])
def test_multi_rank_local_essential_tree(dims, world, dist_kind, nway):
    """Step 6 with a halo (build_local_essential_tree): the lists a rank builds on its
    local essential tree, mapped to global box numbers, are the rows of the global
    traversal for the rank's boxes -- and the tree itself matches the global tree box
    for box (centres, levels, flags, parent/child links)."""
    check_multi_rank_let(dims, world, dist_kind, nway)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind,nway", [(3, 3, "sphere", 1), (3, 4, "uniform", 1),
                                                       (2, 2, "normal", 1), (2, 4, "uniform", 2),
                                                       (3, 8, "sphere", 1), (3, 1, "normal", 1)])
def test_multi_rank_native_entries(dims, world, dist_kind, nway):
    """The same check with steps 1-6 run by the library (bt_mgpu_exchange, bt_mgpu_number,
    bt_mgpu_let_build/_export; ranks = threads over a local communicator)."""
    check_multi_rank_let(dims, world, dist_kind, nway, native=True, expect_partial=world > 2)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind,nway", [(3, 3, "sphere", 1), (2, 4, "uniform", 2),
                                                       (3, 5, "clustered", 1), (2, 2, "normal", 1)])
def test_multi_rank_native_entries_separate_targets(dims, world, dist_kind, nway):
    """Sources and separate point targets through bt_mgpu_exchange: cells are counted over
    both sets, the flags of the shared top boxes come from the per-cell source and target
    counts, and the per-rank trees and lists are those of the single-GPU build."""
    check_multi_rank_let(dims, world, dist_kind, nway, native=True, expect_partial=False,
                         sep_targets=True)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind,nway,shift,ext", [
    (3, 3, "uniform", 1, 1.5, False), (2, 4, "uniform", 2, 0.7, False), (3, 4, "sphere", 1, 2.5, False),
    (3, 3, "uniform", 1, 1.5, True), (2, 5, "uniform", 1, 0.7, True)])
def test_multi_rank_native_entries_disjoint_clouds(dims, world, dist_kind, nway, shift, ext):
    """Sources and targets in different places: shared top boxes that hold only sources, or only
    targets, must carry the flags of the single-GPU tree -- which, like the reference, gives every
    box with children BOTH child flags (tree_build_kernels.py:1254 sets
    HAS_SOURCE_OR_TARGET_CHILD_BOXES before the per-side tests of :1262-1272).  The flags decide
    which top boxes are source parents, target parents and List-2 members."""
    check_multi_rank_let(dims, world, dist_kind, nway, native=True, expect_partial=False, sep_targets=True,
                         target_shift=shift, target_extents=(0.05, 0.25, "linf") if ext else None)


@pytest.mark.gpu
@pytest.mark.parametrize("dims,world,dist_kind,nway,norm", [(3, 2, "uniform", 1, "linf"), (3, 3, "sphere", 1, "linf"),
                                                            (2, 4, "uniform", 2, "l2"), (3, 5, "clustered", 1, "linf"),
                                                            (2, 2, "normal", 1, "l2")])
def test_multi_rank_native_entries_extents(dims, world, dist_kind, nway, norm):
    """Targets with extents (BASELINE configs[3] on N ranks): the exchange counts where the
    targets stop over the shared top levels, a target that stays in a top box travels to the
    owner of the box's first cell, the LET carries target bounding boxes and source counts
    (halo boxes from their owners, shared top boxes by an all-reduce), and the lists -- the
    close lists included -- are those of the single-GPU build with target_radii."""
    scale = {"uniform": 0.05, "sphere": 0.1, "clustered": 0.3, "normal": 0.4}[dist_kind]
    check_multi_rank_let(dims, world, dist_kind, nway, native=True, expect_partial=False,
                         sep_targets=True, target_extents=(scale, 0.25, norm))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_multi_rank_native_entries_extents_random(seed):
    rng = np.random.default_rng(12000 + seed)
    dims = int(rng.choice([2, 3]))
    dist_kind = str(rng.choice(["sphere", "uniform", "normal", "clustered"]))
    scale = {"uniform": 0.05, "sphere": 0.1, "clustered": 0.3, "normal": 0.4}[dist_kind]
    check_multi_rank_let(
        dims, world=int(rng.integers(2, 7)), dist_kind=dist_kind, nway=int(rng.choice([1, 1, 2])),
        n_per=int(rng.choice([3000, 20000])), mpb=int(rng.choice([8, 30, 64])),
        top_level=int(rng.integers(2, 5) if dims == 3 else rng.integers(3, 6)),
        seed=int(rng.integers(0, 10**6)), expect_partial=False, native=True, sep_targets=True,
        target_extents=(scale * float(rng.choice([0.3, 1.0, 3.0])), float(rng.choice([0.0, 0.25, 0.5])),
                        str(rng.choice(["linf", "l2"]))))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_multi_rank_native_entries_random(seed):
    rng = np.random.default_rng(9000 + seed)
    dims = int(rng.choice([2, 3]))
    check_multi_rank_let(
        dims, world=int(rng.integers(2, 9)),
        dist_kind=str(rng.choice(["sphere", "uniform", "normal", "clustered"])),
        nway=int(rng.choice([1, 1, 2])), n_per=int(rng.choice([3000, 20000, 50000])),
        mpb=int(rng.choice([8, 30, 64])),
        top_level=int(rng.integers(2, 5) if dims == 3 else rng.integers(3, 6)),
        seed=int(rng.integers(0, 10**6)), expect_partial=False, native=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(10))
def test_multi_rank_local_essential_tree_random(seed):
    """The same check on random rank counts, sizes, distributions, leaf sizes and
    top levels."""
    rng = np.random.default_rng(7000 + seed)
    dims = int(rng.choice([2, 3]))
    check_multi_rank_let(
        dims, world=int(rng.integers(2, 9)),
        dist_kind=str(rng.choice(["sphere", "uniform", "normal", "clustered"])),
        nway=int(rng.choice([1, 1, 2])), n_per=int(rng.choice([3000, 20000, 50000])),
        mpb=int(rng.choice([8, 30, 64])),
        top_level=int(rng.integers(2, 5) if dims == 3 else rng.integers(3, 6)),
        seed=int(rng.integers(0, 10**6)), expect_partial=False)


def check_multi_rank_let(dims, world, dist_kind, nway, n_per=40000, mpb=30, top_level=None,
                         seed=200, expect_partial=True, native=False, sep_targets=False,
                         target_extents=None, target_shift=0.0):
    """native: steps 1-6 through the library's bt_mgpu_* entries, the ranks being threads
    that share a LocalGroup; otherwise the torch implementation over tests/fake_dist.py."""
    import threading

    import torch
    from fake_dist import FakeWorld
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import (build_local_essential_tree, exchange_particles,
                                         number_sharded_tree)
    if top_level is None:
        top_level = 3 if dims == 3 else 4

    def chunk(rank):
        rng = np.random.default_rng(seed + rank)
        if dist_kind == "clustered":
            return [np.where(rng.random(n_per) < 0.5, 0.3 + 1e-2 * rng.standard_normal(n_per),
                             rng.standard_normal(n_per)) for _ in range(dims)]
        if dist_kind == "sphere":
            v = rng.standard_normal((dims, n_per))
            v /= np.sqrt((v * v).sum(axis=0))
            return [np.ascontiguousarray(v[i]) for i in range(dims)]
        if dist_kind == "uniform":
            return [rng.random(n_per) for _ in range(dims)]
        return [rng.standard_normal(n_per) for _ in range(dims)]

    chunks = [chunk(r) for r in range(world)]
    # separate point targets (native entries only): a third as many, another stream
    tchunks = None
    if sep_targets:
        assert native
        full_n = n_per
        n_per = max(n_per // 3, 1)
        # (target_shift: the target cloud beside the source cloud -- top boxes that hold particles
        # of one kind only)
        tchunks = [[a + target_shift for a in chunk(1000 + r)] for r in range(world)]
        n_per = full_n
    # ... with extents: target_extents = (radius scale, stick_out_factor, extent_norm); radii over
    # four decades, so that most targets go deep and some stay in boxes of the shared top levels
    rchunks = None
    if target_extents is not None:
        assert sep_targets
        rchunks = [target_extents[0] * 10.0 ** np.random.default_rng(seed + 5000 + r).uniform(
            -4, 0, len(tchunks[r][0])) for r in range(world)]
    fw = FakeWorld(world)
    group = None
    if native:
        from boxtree_amd.distributed import native as nat
        group = nat.LocalGroup(world)
    results = [None] * world
    errors = []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            pts = [torch.from_numpy(a).cuda() for a in chunks[rank]]
            if native:
                comm = group.comm(rank)
                if rchunks is not None:
                    tg = [torch.from_numpy(a).cuda() for a in tchunks[rank]]
                    p2, t2, r2, kw, stats = nat.exchange_particles(
                        actx, comm, pts, mpb, top_level=top_level, targets=tg,
                        target_radii=torch.from_numpy(rchunks[rank]).cuda(),
                        stick_out_factor=target_extents[1], extent_norm=target_extents[2])
                    tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, target_radii=r2,
                                                max_particles_in_box=mpb, **kw)
                elif sep_targets:
                    tg = [torch.from_numpy(a).cuda() for a in tchunks[rank]]
                    p2, t2, kw, stats = nat.exchange_particles(actx, comm, pts, mpb, top_level=top_level,
                                                               targets=tg)
                    tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, max_particles_in_box=mpb, **kw)
                else:
                    p2, kw, stats = nat.exchange_particles(actx, comm, pts, mpb, top_level=top_level)
                    tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
                num = nat.number_sharded_tree(actx, comm, tree)
                let, info = nat.build_local_essential_tree(actx, comm, tree, num,
                                                           well_sep_is_n_away=nway)
                comm.close()
            else:
                dist = fw.rank_view(rank)
                p2, _, kw, stats = exchange_particles(actx, dist, pts, None, {}, top_level=top_level,
                                                      max_particles_in_box=mpb)
                tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
                num = number_sharded_tree(dist, tree, stats)
                let, info = build_local_essential_tree(actx, dist, tree, stats, num,
                                                       well_sep_is_n_away=nway)
            trav, _ = FMMTraversalBuilder(actx, well_sep_is_n_away=nway)(
                actx, let, _target_boxes_mask=info["target_boxes_mask"],
                _active_level_ranges=info["active_level_ranges"])
            if native:
                # the LET comes with subtree sizes (own boxes' from the local tree, halo boxes'
                # from their owners, shared top boxes summed): the sizes of its own child table
                sz = getattr(let, "_subtree_sizes", None)
                assert sz is not None
                ch = let.box_child_ids.cpu().numpy()[:, :int(let.nboxes)]
                lv = let.box_levels.cpu().numpy().astype(np.int64)
                want = np.ones(int(let.nboxes), np.int64)
                for l in range(int(lv.max()), 0, -1):
                    idx = np.nonzero(lv == l)[0]
                    par = let.box_parent_ids.cpu().numpy()[idx]
                    np.add.at(want, par, want[idx])
                assert np.array_equal(sz.cpu().numpy().astype(np.int64), want), "LET subtree sizes"
                del ch
            results[rank] = dict(let=actx.to_numpy(let), trav=actx.to_numpy(trav),
                                 gid=info["global_box_ids"].cpu().numpy().astype(np.int64),
                                 mask=info["target_boxes_mask"].cpu().numpy().astype(np.int8),
                                 nhalo=info["halo_boxes_received"], nboxes=info["nboxes"],
                                 nglobal=num["nboxes"],
                                 ext={k: info[k].cpu().numpy() for k in (
                                     "box_target_bounding_box_min", "box_target_bounding_box_max",
                                     "box_source_counts_cumul") if k in info})
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()[-1500:]))
            try:
                fw.barrier.abort()
            except Exception:           # noqa: BLE001
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank is stuck in a collective"
    if group is not None:
        group.close()

    actx = HIPArrayContext(0)
    allpts = [torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()
              for ax in range(dims)]
    alltgts = None
    if sep_targets:
        alltgts = [torch.from_numpy(np.concatenate([c[ax] for c in tchunks])).cuda()
                   for ax in range(dims)]
    ext_kw = {}
    if rchunks is not None:
        ext_kw = dict(target_radii=torch.from_numpy(np.concatenate(rchunks)).cuda(),
                      stick_out_factor=target_extents[1], extent_norm=target_extents[2])
    gt, _ = TreeBuilder(actx)(actx, allpts, targets=alltgts, max_particles_in_box=mpb, **ext_kw)
    full = actx.to_numpy(FMMTraversalBuilder(actx, well_sep_is_n_away=nway)(actx, gt)[0])
    g = actx.to_numpy(gt)

    def rows(starts, lists, sel, m=None):
        """the rows *sel* of a CSR list, box numbers mapped through *m*: (row lengths, entries)
        as bytes, comparable with == (vectorised: the 8-rank rehearsal compares 10^8 entries)"""
        sel = np.asarray(list(sel) if not isinstance(sel, np.ndarray) else sel, np.int64)
        st = np.asarray(starts, np.int64)
        ln = st[sel + 1] - st[sel] if len(sel) else np.zeros(0, np.int64)
        first = np.concatenate([[0], np.cumsum(ln)[:-1]]) if len(ln) else np.zeros(0, np.int64)
        idx = np.repeat(st[sel] - first, ln) + np.arange(int(ln.sum()))
        flat = np.asarray(lists)[idx].astype(np.int64)
        if m is not None:
            flat = m[flat]
        return ln.tobytes(), flat.tobytes()

    pos_t = np.full(g.nboxes, -1, np.int64)
    pos_t[np.asarray(full.target_boxes, np.int64)] = np.arange(len(full.target_boxes))
    pos_p = np.full(g.nboxes, -1, np.int64)
    pos_p[np.asarray(full.target_or_target_parent_boxes, np.int64)] = np.arange(
        len(full.target_or_target_parent_boxes))
    deep_cover = np.zeros(g.nboxes, np.int64)
    tgt_cover = np.zeros(g.nboxes, np.int64)
    for r in results:
        t, tr, gid = r["let"], r["trav"], r["gid"]
        # mask 1: all lists of the box; 2 (extents): the lists it has as a parent of target boxes,
        # its own targets are another rank's
        hm, hm1 = r["mask"] != 0, r["mask"] == 1
        nb = t.nboxes
        assert r["nglobal"] == g.nboxes and nb <= g.nboxes
        if world > 2 and expect_partial:
            assert nb < g.nboxes                   # a halo, not the whole tree
        assert len(np.unique(gid)) == nb
        # the LET is the global tree restricted to its boxes
        assert np.array_equal(t.box_levels, g.box_levels[gid])
        if not np.array_equal(t.box_flags, g.box_flags[gid]):
            bad = np.nonzero(t.box_flags != g.box_flags[gid])[0]
            print("flags differ at LET boxes", bad[:10], "levels", t.box_levels[bad][:10], "got",
                  t.box_flags[bad][:10], "want", g.box_flags[gid][bad][:10], "global ids", gid[bad][:10],
                  "tgt nonchild/cumul", g.box_target_counts_nonchild[gid[bad]][:10],
                  g.box_target_counts_cumul[gid[bad]][:10], "src cumul", g.box_source_counts_cumul[gid[bad]][:10],
                  "children", g.box_child_ids[:, gid[bad[0]]])
        assert np.array_equal(t.box_flags, g.box_flags[gid])
        assert np.array_equal(t.box_centers[:, :nb], g.box_centers[:, gid])
        assert np.array_equal(gid[t.box_parent_ids], g.box_parent_ids[gid])
        ch = t.box_child_ids[:, :nb]
        mapped = np.where(ch != 0, gid[ch], 0)
        assert np.all((mapped == 0) | (mapped == g.box_child_ids[:, gid]))
        mine_deep = hm & (t.box_levels > top_level)
        assert np.array_equal(mapped[:, mine_deep], g.box_child_ids[:, gid[mine_deep]])
        deep_cover[gid[mine_deep]] += 1
        # every list array ends where its starts say (no slack behind the last row)
        for name in ("same_level_non_well_sep_boxes", "neighbor_source_boxes", "from_sep_siblings",
                     "from_sep_bigger", "from_sep_close_smaller", "from_sep_close_bigger"):
            st_, li_ = getattr(tr, name + "_starts", None), getattr(tr, name + "_lists", None)
            if st_ is not None:
                assert len(li_) == st_[-1], name
        # lists, mapped to global numbers
        gt_boxes = gid[tr.target_boxes]
        assert np.all(hm1[tr.target_boxes])
        tgt_cover[gt_boxes] += 1
        sel_t = pos_t[gt_boxes]
        assert np.all(sel_t >= 0)
        gp_boxes = gid[tr.target_or_target_parent_boxes]
        sel_p = pos_p[gp_boxes]
        assert np.all(sel_p >= 0)
        assert rows(tr.neighbor_source_boxes_starts, tr.neighbor_source_boxes_lists,
                    range(len(sel_t)), gid) == rows(full.neighbor_source_boxes_starts,
                                                    full.neighbor_source_boxes_lists, sel_t)
        for name in ("from_sep_siblings", "from_sep_bigger"):
            got = rows(getattr(tr, name + "_starts"), getattr(tr, name + "_lists"),
                       range(len(sel_p)), gid)
            want = rows(getattr(full, name + "_starts"), getattr(full, name + "_lists"), sel_p)
            assert got == want, name
        act = np.nonzero(hm)[0]
        assert rows(tr.same_level_non_well_sep_boxes_starts, tr.same_level_non_well_sep_boxes_lists,
                    act, gid) == rows(full.same_level_non_well_sep_boxes_starts,
                                      full.same_level_non_well_sep_boxes_lists, gid[act])
        if rchunks is not None:
            for name in ("from_sep_close_smaller", "from_sep_close_bigger"):
                got = rows(getattr(tr, name + "_starts"), getattr(tr, name + "_lists"),
                           range(len(sel_t)), gid)
                want = rows(getattr(full, name + "_starts"), getattr(full, name + "_lists"), sel_t)
                assert got == want, name
            # what the traversal read of the LET's boxes: the global tree's values, every box
            ex = r["ext"]
            assert np.array_equal(ex["box_target_bounding_box_min"][:, :nb],
                                  g.box_target_bounding_box_min[:, gid])
            assert np.array_equal(ex["box_target_bounding_box_max"][:, :nb],
                                  g.box_target_bounding_box_max[:, gid])
            assert np.array_equal(ex["box_source_counts_cumul"], g.box_source_counts_cumul[gid])
        hm_global = np.zeros(g.nboxes, bool)
        hm_global[gid[hm1]] = True
        for lev in range(g.nlevels):
            a, b = tr.from_sep_smaller_by_level[lev], full.from_sep_smaller_by_level[lev]
            # (LET boxes are the global tree's in the same order: both key lists ascend)
            tb_got = gid[np.asarray(tr.target_boxes_sep_smaller_by_source_level[lev], np.int64)]
            tb_full = np.asarray(full.target_boxes_sep_smaller_by_source_level[lev], np.int64)
            keep = np.nonzero(hm_global[tb_full])[0]
            same = (np.array_equal(tb_got, tb_full[keep])
                    and rows(a.starts, a.lists, np.arange(len(tb_got)), gid) == rows(b.starts, b.lists, keep))
            if not same:            # what differs, for the first few boxes
                got = {int(gid[tb]): gid[a.lists[a.starts[i]:a.starts[i + 1]]].tolist()
                       for i, tb in enumerate(tr.target_boxes_sep_smaller_by_source_level[lev])}
                want = {int(tb): b.lists[b.starts[i]:b.starts[i + 1]].tolist()
                        for i, tb in enumerate(full.target_boxes_sep_smaller_by_source_level[lev])
                        if hm_global[tb]}
                for tb in sorted(set(got) | set(want))[:400]:
                    gl, wl = got.get(tb), want.get(tb)
                    if gl != wl:
                        gs, ws = set(gl or []), set(wl or [])
                        print("list 3, source level", lev, "target box", tb, "level", g.box_levels[tb],
                              "got", None if gl is None else len(gl), "want", None if wl is None else len(wl),
                              "missing", [(int(x), int(g.box_levels[x])) for x in sorted(ws - gs)][:8],
                              "extra", [(int(x), int(g.box_levels[x])) for x in sorted(gs - ws)][:8],
                              "in LET", [bool(np.isin(x, gid)) for x in sorted(ws - gs)][:8])
            assert same, f"list 3, source level {lev}"
    # every box below the top levels is some rank's own, exactly once
    assert np.all(deep_cover[g.box_levels > top_level] == 1)
    # the lists of every target box are built by exactly one rank
    assert np.all(tgt_cover[full.target_boxes] == 1) and tgt_cover.sum() == len(full.target_boxes)


@pytest.mark.gpu
def test_list_beyond_int32_csr_limit_raises(actx):
    """2*10^8 uniform 3D points, mpb 64: list 2 would hold 3.1*10^9 entries -- past the
    int32 CSR starts of the reference.  An error, not a wrapped count."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    pts = [torch.rand(2 * 10**8, generator=g, dtype=torch.float32, device="cuda") for _ in range(3)]
    tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=64)
    with pytest.raises(NotImplementedError, match="int32 CSR limit"):
        FMMTraversalBuilder(actx)(actx, tree)


@pytest.mark.gpu
def test_tree_of_boxes_without_level_starts(actx, oracle):
    """A TreeOfBoxes with level_start_box_nrs=None (allowed by boxtree/tree.py:236) is
    accepted by the traversal and the peer-list builders."""
    from boxtree_amd import FMMTraversalBuilder, PeerListFinder, TreeBuilder, TreeOfBoxes
    p = normal_particles(20000, 3, np.float64, seed=4)
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(x) for x in p], max_particles_in_box=30)
    tob = TreeOfBoxes(
        root_extent=tree.root_extent, box_centers=tree.box_centers,
        box_parent_ids=tree.box_parent_ids, box_child_ids=tree.box_child_ids,
        box_levels=tree.box_levels, box_flags=tree.box_flags, level_start_box_nrs=None,
        box_id_dtype=np.dtype(np.int32), box_level_dtype=np.dtype(np.uint8),
        coord_dtype=np.dtype(np.float64), sources_have_extent=False,
        targets_have_extent=False, extent_norm=None, stick_out_factor=0.0, _is_pruned=True)
    assert tob.nlevels == tree.nlevels
    otree = oracle.build_tree(p, max_particles_in_box=30)
    trav, _ = FMMTraversalBuilder(actx)(actx, tob)
    assert_same_traversal(actx.to_numpy(trav), oracle.build_traversal(otree))
    pl, _ = PeerListFinder(actx)(actx, tob)
    opl = oracle.peer_lists(otree)
    assert np.array_equal(actx.to_numpy(pl.peer_list_starts), opl.peer_list_starts)
    assert np.array_equal(actx.to_numpy(pl.peer_lists), opl.peer_lists)


@pytest.mark.gpu
def test_tree_of_boxes_children_not_consecutive(actx):
    """The lattice list kernels keep a box's children as "first child + masks", which needs
    the children of a box numbered consecutively in slot order; the structure check must send
    any other tree to the kernels that read the child table as it is.  The boxes of every
    level of a built tree are renumbered in reverse (still level by level, parents before
    children): the lists of the renumbered tree are the original lists, renumbered."""
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder, TreeOfBoxes
    p = normal_particles(30000, 3, np.float64, seed=11)
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(x) for x in p], max_particles_in_box=25)
    h = actx.to_numpy(tree)
    nb, al = h.nboxes, h.box_child_ids.shape[-1]
    ls = np.asarray(h.level_start_box_nrs)
    new_of_old = np.arange(nb)
    for lev in range(len(ls) - 1):
        a, b = int(ls[lev]), int(ls[lev + 1])
        new_of_old[a:b] = np.arange(b - 1, a - 1, -1)
    old_of_new = np.empty(nb, np.int64)
    old_of_new[new_of_old] = np.arange(nb)

    def per_box(arr):                       # [..., aligned] -> boxes permuted, padding kept
        out = arr.copy()
        out[..., :nb] = arr[..., :nb][..., old_of_new]
        return out

    child = per_box(h.box_child_ids)
    kids = child[:, :nb]
    kids[kids != 0] = new_of_old[kids[kids != 0]]
    parent = per_box(h.box_parent_ids)
    parent[:nb] = new_of_old[parent[:nb]]
    tob = TreeOfBoxes(
        root_extent=h.root_extent, box_centers=actx.from_numpy(per_box(h.box_centers)),
        box_parent_ids=actx.from_numpy(parent), box_child_ids=actx.from_numpy(child),
        box_levels=actx.from_numpy(per_box(h.box_levels)), box_flags=actx.from_numpy(per_box(h.box_flags)),
        level_start_box_nrs=h.level_start_box_nrs,
        box_id_dtype=np.dtype(np.int32), box_level_dtype=np.dtype(np.uint8),
        coord_dtype=np.dtype(np.float64), sources_have_extent=False,
        targets_have_extent=False, extent_norm=None, stick_out_factor=0.0, _is_pruned=True)
    assert child.shape[-1] == al
    ref = actx.to_numpy(FMMTraversalBuilder(actx)(actx, tree)[0])
    got = actx.to_numpy(FMMTraversalBuilder(actx)(actx, tob)[0])

    def rows(t, starts, lists, boxes, to_old):
        s, l = np.asarray(getattr(t, starts)), np.asarray(getattr(t, lists))
        return {int(to_old[int(b)]): sorted(to_old[l[s[i]:s[i + 1]]].tolist())
                for i, b in enumerate(np.asarray(boxes))}

    ident = np.arange(nb)
    for name, boxes in (("same_level_non_well_sep_boxes", lambda t: np.arange(nb)),
                        ("neighbor_source_boxes", lambda t: t.target_boxes),
                        ("from_sep_siblings", lambda t: t.target_or_target_parent_boxes),
                        ("from_sep_bigger", lambda t: t.target_or_target_parent_boxes)):
        want = rows(ref, name + "_starts", name + "_lists", boxes(ref), ident)
        have = rows(got, name + "_starts", name + "_lists", boxes(got), old_of_new)
        assert have == want, name
    for lev in range(h.nlevels):
        a, b = got.from_sep_smaller_by_level[lev], ref.from_sep_smaller_by_level[lev]
        have = {int(old_of_new[tb]): sorted(old_of_new[a.lists[a.starts[i]:a.starts[i + 1]]].tolist())
                for i, tb in enumerate(got.target_boxes_sep_smaller_by_source_level[lev])}
        want = {int(tb): sorted(b.lists[b.starts[i]:b.starts[i + 1]].tolist())
                for i, tb in enumerate(ref.target_boxes_sep_smaller_by_source_level[lev])}
        assert have == want, lev
