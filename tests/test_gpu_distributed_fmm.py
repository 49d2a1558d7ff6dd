"""Work partition, box masks, local trees and the distributed FMM driver
(boxtree/distributed/partition.py, local_tree.py, calculation.py) on the device,
against the Python restatement in oracle/oracle.py and, end to end, against the
reference's own check: constant-one potentials == nsources on rank 0
(test/test_distributed.py:182-290).  Ranks are threads of one process sharing the
GPU (tests/fake_dist.py); RCCL itself needs more than one GPU."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def actx():
    from boxtree_amd import HIPArrayContext
    return HIPArrayContext(0)


def _build(actx, dims, n, ntargets=None, extent=False, seed=3, mpb=30):
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    rng = np.random.default_rng(seed)
    src = [rng.standard_normal(n) for _ in range(dims)]
    kw = dict(max_particles_in_box=mpb)
    if ntargets:
        tgt = [rng.standard_normal(ntargets) + (2.0 if ax == 0 else 0.0) for ax in range(dims)]
        kw["targets"] = [actx.from_numpy(t) for t in tgt]
    if extent:
        radii = 2.0 ** rng.uniform(-10, 0, ntargets) * 0.05
        kw.update(target_radii=actx.from_numpy(radii), stick_out_factor=0.25)
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(s) for s in src], **kw)
    trav, _ = FMMTraversalBuilder(actx)(actx, tree)
    return tree, trav


CASES = [(2, 6000, None, False), (3, 20000, None, False), (3, 15000, 9000, False),
         (3, 15000, 6000, True), (2, 4000, 3000, True)]


@pytest.mark.parametrize("dims,n,ntargets,extent", CASES)
def test_dfs_order_and_partition(actx, oracle, dims, n, ntargets, extent):
    from fake_dist import FakeWorld
    from boxtree_amd.cost import FMMCostModel
    from boxtree_amd.distributed.partition import get_box_ids_dfs_order, partition_work
    tree, trav = _build(actx, dims, n, ntargets, extent)
    htree = actx.to_numpy(tree)
    order = get_box_ids_dfs_order(actx, tree).cpu().numpy()
    want = oracle.dfs_order(htree)
    assert np.array_equal(order, want)

    nlevels = int(tree.nlevels)
    cost = FMMCostModel().cost_per_box(actx, trav, np.ones(nlevels, np.int32),
                                       FMMCostModel.get_unit_calibration_params())
    hcost = cost.cpu().numpy()
    assert np.all(hcost == np.round(hcost))       # unit parameters: exactly summable
    for world in (1, 2, 3, 4, 7):
        seg = oracle.partition_work_segments(hcost, want, world)
        fw = FakeWorld(world)
        got = [None] * world
        errors = []

        def run(rank, world=world, fw=fw, got=got, errors=errors):
            try:
                from boxtree_amd import HIPArrayContext
                a = HIPArrayContext(0)
                got[rank] = partition_work(a, cost if rank == 0 else None, trav,
                                           fw.rank_view(rank)).cpu().numpy()
            except BaseException as e:      # noqa: BLE001
                errors.append((rank, repr(e)))
                fw.barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        assert not errors, errors
        for r in range(world):
            assert np.array_equal(got[r], want[seg[r, 0]:seg[r, 1]]), (world, r)
        assert sum(len(g) for g in got) == tree.nboxes
        # about balanced: no share above its ideal by more than the heaviest box
        shares = np.array([hcost[g].sum() for g in got])
        assert shares.max() <= hcost.sum() / world + hcost.max() + 1e-9


def test_partition_skewed_costs(actx, oracle):
    """All the cost in a handful of boxes: a box ends at most one segment, trailing
    ranks may come out empty -- same table as the reference's loop."""
    from boxtree_amd.distributed.partition import get_box_ids_dfs_order
    import ctypes as ct
    from boxtree_amd.array_context import ptr
    tree, _ = _build(actx, 2, 3000)
    order = get_box_ids_dfs_order(actx, tree)
    horder = order.cpu().numpy()
    nboxes = int(tree.nboxes)
    rng = np.random.default_rng(0)
    for trial in range(6):
        hcost = np.zeros(nboxes)
        hot = rng.choice(nboxes, size=3, replace=False)
        hcost[hot] = rng.integers(1, 1000, size=3)
        if trial == 0:
            hcost[:] = 0
            hcost[horder[-1]] = 5.0          # everything on the last box visited
        if trial == 1:
            hcost[:] = 0                      # nothing exceeds a zero threshold
        for world in (2, 5, 9):
            seg = np.zeros((world, 2), np.int32)
            code = actx.lib.bt_partition_work(
                actx.handle, nboxes, ptr(order), ptr(actx.from_numpy(hcost)), world,
                seg.ctypes.data_as(ct.POINTER(ct.c_int32)))
            assert code == 0
            assert np.array_equal(seg, oracle.partition_work_segments(hcost, horder, world))


@pytest.mark.parametrize("dims,n,ntargets,extent", CASES)
def test_box_masks_and_local_tree(actx, oracle, dims, n, ntargets, extent):
    from fake_dist import FakeWorld
    from boxtree_amd.distributed.local_tree import generate_local_tree
    from boxtree_amd.distributed.partition import get_box_ids_dfs_order, get_box_masks
    tree, trav = _build(actx, dims, n, ntargets, extent)
    htree, htrav = actx.to_numpy(tree), actx.to_numpy(trav)
    order = get_box_ids_dfs_order(actx, tree)
    nboxes = int(tree.nboxes)
    comm = FakeWorld(1).rank_view(0)
    for lo, hi in ((0, nboxes // 3), (nboxes // 3, 2 * nboxes // 3), (nboxes - 7, nboxes),
                   (5, 6)):
        resp = order[lo:hi].contiguous()
        hresp = resp.cpu().numpy()
        masks = get_box_masks(actx, trav, resp)
        want = oracle.box_masks(htree, htrav, hresp)
        for name, w in want.items():
            assert np.array_equal(getattr(masks, name).cpu().numpy(), w), (name, lo, hi)

        local, src_idx, tgt_idx = generate_local_tree(actx, trav, resp, comm)
        hl = actx.to_numpy(local)
        for kind, mask, idx in (("source", want["point_src_boxes"], src_idx),
                                ("target", want["responsible_boxes"], tgt_idx)):
            starts = getattr(htree, f"box_{kind}_starts")
            nonchild = getattr(htree, f"box_{kind}_counts_nonchild")
            cumul = getattr(htree, f"box_{kind}_counts_cumul")
            npart = htree.nsources if kind == "source" else htree.ntargets
            ws, wn, wc, widx = oracle.local_particles_and_lists(mask, starts, nonchild, cumul,
                                                                npart)
            assert np.array_equal(getattr(hl, f"box_{kind}_starts"), ws)
            assert np.array_equal(getattr(hl, f"box_{kind}_counts_nonchild"), wn)
            assert np.array_equal(getattr(hl, f"box_{kind}_counts_cumul"), wc)
            assert np.array_equal(idx.cpu().numpy(), widx)
            glob = htree.sources if kind == "source" else htree.targets
            loc = hl.sources if kind == "source" else hl.targets
            for ax in range(dims):
                assert np.array_equal(loc[ax], glob[ax][widx])
        if extent:
            assert np.array_equal(hl.target_radii, htree.target_radii[tgt_idx.cpu().numpy()])
        assert np.array_equal(hl.box_flags, oracle.modify_target_flags(
            htree.box_flags, hl.box_target_counts_nonchild, hl.box_target_counts_cumul))
        ustarts, ulists = oracle.box_to_user_ranks(want["multipole_src_boxes"][None, :])
        assert np.array_equal(hl.box_to_user_rank_starts, ustarts)
        assert np.array_equal(hl.box_to_user_rank_lists, ulists)
        assert np.array_equal(hl.responsible_boxes_mask, want["responsible_boxes"])
        assert np.array_equal(hl.ancestor_mask, want["ancestor_boxes"])


def test_box_to_user_ranks_and_subrange(actx, oracle):
    import ctypes as ct
    from boxtree_amd.array_context import ptr
    rng = np.random.default_rng(5)
    nranks, nboxes = 6, 5000
    masks = (rng.random((nranks, nboxes)) < 0.3).astype(np.int8)
    masks[:, 17] = 0
    masks[:, 18] = 1
    d_masks = actx.from_numpy(masks)
    starts = actx.empty(nboxes + 1, np.int32)
    n = ct.c_int64(0)
    assert actx.lib.bt_box_to_user_ranks(actx.handle, nranks, nboxes, ptr(d_masks), ptr(starts),
                                         None, ct.byref(n)) == 0
    lists = actx.empty(int(n.value), np.int32)
    assert actx.lib.bt_box_to_user_ranks(actx.handle, nranks, nboxes, ptr(d_masks), ptr(starts),
                                         ptr(lists), ct.byref(n)) == 0
    ws, wl = oracle.box_to_user_ranks(masks)
    assert np.array_equal(starts.cpu().numpy(), ws)
    assert np.array_equal(lists.cpu().numpy(), wl)

    contributing = (rng.random(nboxes) < 0.5).astype(np.int8)
    for lo, hi in ((0, 3), (3, 6), (2, 3), (0, 6), (4, 4)):
        boxes = actx.empty(nboxes, np.int32)
        assert actx.lib.bt_boxes_used_by_ranks(
            actx.handle, nboxes, ptr(actx.from_numpy(contributing)), lo, hi, ptr(starts),
            ptr(lists), ptr(boxes), ct.byref(n)) == 0
        want = [b for b in range(nboxes) if contributing[b] and masks[lo:hi, b].any()]
        assert np.array_equal(boxes[:int(n.value)].cpu().numpy(), want)


@pytest.mark.parametrize("world,dims,nsources,ntargets,extent,allreduce", [
    (4, 3, 10000, 10000, False, False),     # the reference's case, test_distributed.py:272-290
    (4, 3, 10000, 10000, False, True),
    (3, 2, 8000, None, False, False),
    (5, 3, 20000, 7000, True, False),
    (7, 3, 30000, None, False, False),
    (1, 3, 5000, 5000, False, False),
])
def test_constantone_distributed(world, dims, nsources, ntargets, extent, allreduce):
    import torch
    from fake_dist import FakeWorld
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.constant_one import ConstantOneTreeIndependentDataForWrangler
    from boxtree_amd.distributed import DistributedFMMRunner
    from boxtree_amd.distributed.calculation import DistributedConstantOneExpansionWrangler
    fw = FakeWorld(world)
    results = [None] * world
    errors = []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = fw.rank_view(rank)
            tree = None
            weights = torch.empty(0, dtype=torch.float64, device="cuda")
            if rank == 0:
                rng = np.random.default_rng(15)
                src = [actx.from_numpy(rng.standard_normal(nsources)) for _ in range(dims)]
                kw = dict(max_particles_in_box=30)
                if ntargets:
                    kw["targets"] = [actx.from_numpy(rng.standard_normal(ntargets)
                                                     + (2.0 if ax == 0 else 0.0))
                                     for ax in range(dims)]
                if extent:
                    kw.update(target_radii=actx.from_numpy(
                        2.0 ** rng.uniform(-10, 0, ntargets) * 0.05), stick_out_factor=0.25)
                tree, _ = TreeBuilder(actx)(actx, src, **kw)
                weights = torch.ones(nsources, dtype=torch.float64, device="cuda")
            tg = FMMTraversalBuilder(actx)
            tree_indep = ConstantOneTreeIndependentDataForWrangler()

            def wrangler_factory(local_traversal, global_traversal):
                return DistributedConstantOneExpansionWrangler(
                    comm, tree_indep, local_traversal, global_traversal,
                    communicate_mpoles_via_allreduce=allreduce)

            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runner = DistributedFMMRunner(actx, tree, tg, wrangler_factory, comm=comm)
            pot = runner.drive_dfmm(actx, [weights])
            ltree = runner.wrangler.traversal.tree
            results[rank] = dict(
                pot=None if pot is None else pot.cpu().numpy(),
                ntargets=int(ltree.ntargets), nsources=int(ltree.nsources),
                nresp=int(ltree.responsible_boxes_list.shape[0]))
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()[-2000:]))
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
    pot = results[0]["pot"]
    assert pot.shape == (ntargets or nsources,)
    assert np.all(pot == nsources)
    assert all(r["pot"] is None for r in results[1:])
    # every target is evaluated by exactly one rank
    assert sum(r["ntargets"] for r in results) == (ntargets or nsources)
    # sources are replicated where list 1 crosses an ownership boundary
    assert sum(r["nsources"] for r in results) >= nsources
