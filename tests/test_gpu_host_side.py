"""The host side of a call: root box on the device, stream-ordered
results, the caller's stream, optional stage events.  Every variant must produce the
arrays of the plain, host-synchronous variant bit for bit."""

import numpy as np
import pytest

from compare import assert_same_traversal, assert_same_tree

pytestmark = pytest.mark.gpu


def points(n, d, dtype=np.float64, seed=3):
    rng = np.random.default_rng(seed)
    return [rng.random(n).astype(dtype) * (1 + 3 * i) - 0.5 * i for i in range(d)]


def build(actx, pts, targets=None, **kw):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    dev = [actx.from_numpy(p) for p in pts]
    tdev = None if targets is None else [actx.from_numpy(t) for t in targets]
    tree, ev_t = TreeBuilder(actx)(actx, dev, targets=tdev, max_particles_in_box=20, **kw)
    trav, ev_v = FMMTraversalBuilder(actx)(actx, tree)
    return tree, trav, ev_t, ev_v


@pytest.mark.parametrize("d,dtype,n", [(2, np.float64, 30000), (3, np.float64, 40000),
                                       (3, np.float32, 20000), (1, np.float64, 500)])
@pytest.mark.parametrize("with_targets", [False, True])
def test_device_root_box_is_the_host_root_box(monkeypatch, d, dtype, n, with_targets):
    """bt_tree_params.compute_root_box reproduces the numpy arithmetic of
    boxtree_amd.tree_build._root_box (tree_build.py:456-476 upstream) in the coordinate
    type: same bounding box, root extent and therefore the same tree."""
    from boxtree_amd import HIPArrayContext
    actx = HIPArrayContext(0)
    pts = points(n, d, dtype)
    tg = points(n // 3, d, dtype, seed=11) if with_targets else None
    if tg is not None:
        tg[0] = tg[0] * dtype(1.5) - dtype(0.7)          # targets stick out of the sources' box
    tree_dev, trav_dev, _, _ = build(actx, pts, tg)
    monkeypatch.setenv("BOXTREE_HIP_HOST_ROOT_BOX", "1")
    tree_host, trav_host, _, _ = build(actx, pts, tg)
    monkeypatch.delenv("BOXTREE_HIP_HOST_ROOT_BOX")
    h_dev, h_host = actx.to_numpy(tree_dev), actx.to_numpy(tree_host)
    assert np.dtype(h_dev.coord_dtype) == np.dtype(dtype)
    assert type(h_dev.root_extent) is type(h_host.root_extent)
    assert_same_tree(h_dev, h_host)
    assert_same_traversal(actx.to_numpy(trav_dev), actx.to_numpy(trav_host))


def test_stream_ordered_results_equal_synchronous_results(monkeypatch):
    """A context with stream-ordered results (the default) and one that waits at the end of
    every call give the same arrays; the events of the former wait for the stream."""
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.tools import DoneEvent, StreamEvent
    pts = points(60000, 3)
    actx = HIPArrayContext(0)
    assert actx.stream_ordered
    tree, trav, ev_t, ev_v = build(actx, pts)
    assert isinstance(ev_t, StreamEvent) and isinstance(ev_v, StreamEvent)
    ev_t.wait()
    ev_v.wait()
    monkeypatch.setenv("BOXTREE_HIP_STREAM_ORDERED", "0")
    sync_actx = HIPArrayContext(0)
    monkeypatch.delenv("BOXTREE_HIP_STREAM_ORDERED")
    assert not sync_actx.stream_ordered
    tree_s, trav_s, ev_ts, _ = build(sync_actx, pts)
    assert isinstance(ev_ts, DoneEvent)
    assert_same_tree(actx.to_numpy(tree), sync_actx.to_numpy(tree_s))
    assert_same_traversal(actx.to_numpy(trav), sync_actx.to_numpy(trav_s))


def test_results_are_ordered_by_the_callers_stream():
    """The library queues its kernels on torch's current stream: inputs produced by torch
    ops on a side stream, the build, and torch ops on the outputs need no events."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    pts = points(50000, 3)
    ref_tree, ref_trav, _, _ = build(actx, pts)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        base = [actx.from_numpy(p) for p in pts]
        dev = [b * 1.0 for b in base]                    # produced on the side stream
        tree, _ = TreeBuilder(actx)(actx, dev, max_particles_in_box=20)
        trav, _ = FMMTraversalBuilder(actx)(actx, tree)
        total = int(tree.box_source_counts_cumul[0])     # a torch op on an output
        n_l2 = int(trav.from_sep_siblings_starts[-1])
    side.synchronize()
    assert total == 50000
    assert n_l2 == int(ref_trav.from_sep_siblings_starts[-1])
    assert_same_tree(actx.to_numpy(tree), actx.to_numpy(ref_tree))
    assert_same_traversal(actx.to_numpy(trav), actx.to_numpy(ref_trav))
    # and back on the default stream
    tree2, trav2, _, _ = build(actx, pts)
    assert_same_tree(actx.to_numpy(tree2), actx.to_numpy(ref_tree))


def test_stage_events_are_optional():
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    actx = HIPArrayContext(0)
    pts = [actx.from_numpy(p) for p in points(30000, 3)]
    tb = TreeBuilder(actx)
    tree, _ = tb(actx, pts, max_particles_in_box=20)
    FMMTraversalBuilder(actx)(actx, tree)
    times = tb.last_stage_times
    assert {"keygen", "sort", "boxes", "gather"} <= set(times)
    assert any(k.startswith("trav:") for k in times)
    assert all(v >= 0 for v in times.values())
    actx.set_stage_timing(False)
    tree2, _ = tb(actx, pts, max_particles_in_box=20)
    FMMTraversalBuilder(actx)(actx, tree2)
    assert tb.last_stage_times == {}
    assert_same_tree(actx.to_numpy(tree2), actx.to_numpy(tree))
    actx.set_stage_timing(True)


def _subtree_sizes_by_hand(h_tree):
    child = np.asarray(h_tree.box_child_ids)[:, :h_tree.nboxes]
    levels = np.asarray(h_tree.box_levels)
    sizes = np.ones(h_tree.nboxes, np.int64)
    for b in np.argsort(-levels.astype(np.int64), kind="stable"):    # deepest first
        kids = child[:, b]
        sizes[b] += sizes[kids[kids != 0]].sum()
    return sizes


@pytest.mark.parametrize("d,n,kw", [
    (3, 60000, {}),
    (2, 30000, {}),
    (1, 700, {}),
    (3, 20000, {"kind": "adaptive-level-restricted"}),
    (2, 20000, {"kind": "non-adaptive"}),
    (3, 30000, {"separate_targets": True}),
    (2, 20000, {"separate_targets": True, "target_radii": True}),
])
def test_subtree_sizes_of_the_export(monkeypatch, d, n, kw):
    """bt_tree_arrays.box_subtree_sizes counts the boxes under every box, and a traversal
    that starts from it (bt_trav_params.box_subtree_sizes) equals one that counts itself."""
    from boxtree_amd import HIPArrayContext
    actx = HIPArrayContext(0)
    kw = dict(kw)
    pts = points(n, d)
    targets = points(n // 2, d, seed=5) if kw.pop("separate_targets", False) else None
    if kw.pop("target_radii", False):
        rng = np.random.default_rng(9)
        kw["target_radii"] = actx.from_numpy(rng.random(n // 2) * 1e-3)
        kw["stick_out_factor"] = 0.25
    tree, trav, _, _ = build(actx, pts, targets, **kw)
    h_tree = actx.to_numpy(tree)
    got = tree._subtree_sizes.cpu().numpy()
    assert got.dtype == np.int32 and got.shape == (h_tree.nboxes,)
    assert np.array_equal(got, _subtree_sizes_by_hand(h_tree))
    assert got[0] == h_tree.nboxes
    monkeypatch.setenv("BOXTREE_HIP_SUBTREE_SIZES", "0")
    from boxtree_amd import FMMTraversalBuilder
    trav_counted, _ = FMMTraversalBuilder(actx)(actx, tree)
    monkeypatch.delenv("BOXTREE_HIP_SUBTREE_SIZES")
    assert_same_traversal(actx.to_numpy(trav), actx.to_numpy(trav_counted))
