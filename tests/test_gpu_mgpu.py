"""bt_mgpu_exchange (the C-ABI entry of the sharded build) with a one-rank RCCL
communicator on the one GPU a test box has: root box, ownership cells, the packed
receive buffer and the top-tree prefix must reproduce what the torch.distributed
path computes, and the shard must build the same tree as the plain single-GPU call.
(More ranks need more GPUs; the host part -- who owns which cell -- is compared with
the Python plan for 1..8 ranks in tests/test_cabi.py.)"""

import ctypes as ct
import os

import numpy as np
import pytest

from compare import assert_same_tree  # noqa: I001

pytestmark = pytest.mark.gpu


def one_rank_comm():
    import torch
    rccl = ct.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ct.Structure):
        _fields_ = [("internal", ct.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ct.byref(uid)) == 0
    comm = ct.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, UniqueId, ct.c_int]
    assert rccl.ncclCommInitRank(ct.byref(comm), 1, uid, 0) == 0
    return rccl, comm


@pytest.mark.parametrize("dims,dtype", [(3, np.float64), (2, np.float64), (3, np.float32)])
def test_exchange_one_rank(dims, dtype):
    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder, _lib
    from boxtree_amd.distributed import morton_cells, top_tree_plan
    actx = HIPArrayContext(0)
    torch.cuda.set_device(0)
    rccl, comm = one_rank_comm()
    try:
        n, mpb = 200000, 30
        rng = np.random.default_rng(dims)
        pts = [torch.from_numpy(rng.standard_normal(n).astype(dtype)).cuda() for _ in range(dims)]
        par = _lib.MgpuParams()
        par.dims = dims
        par.coord_kind = _lib.BT_F64 if dtype == np.float64 else _lib.BT_F32
        par.n = n
        for ax in range(dims):
            par.coords[ax] = pts[ax].data_ptr()
        par.top_level = 0
        par.max_particles_in_box = mpb
        shard = _lib.MgpuShard()
        torch.cuda.synchronize()
        mc = ct.c_void_p()
        _lib.check(actx.lib.bt_mgpu_comm_rccl(comm, 0, 1, ct.byref(mc)))
        _lib.check(actx.lib.bt_mgpu_exchange(actx.handle, mc, ct.byref(par), ct.byref(shard)))
        actx.lib.bt_mgpu_comm_destroy(mc)
        assert shard.n_owned == n and shard.bytes_sent == 0 and shard.rounds == 1

        # the receive buffer: one rank owns everything, in the original order
        es = np.dtype(dtype).itemsize
        recv = torch.empty(n * dims, dtype=pts[0].dtype, device="cuda")
        hip = ct.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(ct.c_void_p(recv.data_ptr()), ct.c_void_p(shard.points),
                             ct.c_size_t(n * dims * es), 3) == 0
        for ax in range(dims):
            assert torch.equal(recv.view(n, dims)[:, ax], pts[ax])

        # root box: tree_build.py:462-476 on the global (= local) bounding box
        lo = np.array([float(p.min()) for p in pts], dtype=dtype)
        hi = np.array([float(p.max()) for p in pts], dtype=dtype)
        root_extent = (hi - lo).max() * (1 + 1e-4)
        assert np.array_equal(np.array(shard.bbox_min[:dims], dtype=dtype), lo)
        assert np.array_equal(np.array(shard.bbox_max[:dims], dtype=dtype), lo + root_extent)
        assert dtype(shard.root_extent) == root_extent

        # ownership cells and their prefix sums
        k = shard.top_level
        assert k == (5 if dims == 3 else 7)
        cells = morton_cells(pts, lo, lo + root_extent, k)
        hist = torch.bincount(cells, minlength=1 << (dims * k)).cpu().numpy()
        plan = top_tree_plan(hist, dims, k, mpb)
        prefix = torch.empty((1 << (dims * k)) + 1, dtype=torch.int64, device="cuda")
        assert hip.hipMemcpy(ct.c_void_p(prefix.data_ptr()), ct.c_void_p(shard.top_cell_prefix),
                             ct.c_size_t(prefix.numel() * 8), 3) == 0
        assert np.array_equal(prefix.cpu().numpy(), plan["cell_prefix"])

        # the shard builds the tree of the plain call
        views = [recv.view(n, dims)[:, ax] for ax in range(dims)]
        kw = dict(_root_box=(lo, lo + root_extent, root_extent), _top_tree=(k, prefix))
        if dims > 1:
            kw["_point_stride"] = dims
        else:
            views = [v.contiguous() for v in views]
        t_shard, _ = TreeBuilder(actx)(actx, views, max_particles_in_box=mpb, **kw)
        t_plain, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=mpb)
        assert_same_tree(actx.to_numpy(t_shard), actx.to_numpy(t_plain))
    finally:
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("dims,nway", [(3, 1), (2, 2)])
def test_steps_1_to_6_one_rank_rccl(dims, nway):
    """Exchange, build, numbering and the local essential tree through the C ABI on a real
    one-rank RCCL communicator (boxtree_amd.distributed.native): the numbering is the
    identity, the LET is the tree itself and its lists are the plain traversal's."""
    import torch
    import torch.distributed as dist
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    from compare import assert_same_traversal
    actx = HIPArrayContext(0)
    torch.cuda.set_device(0)

    comm = nat.rccl_comm(actx, _OneRank)
    try:
        rng = np.random.default_rng(3 + dims)
        n, mpb = 150000, 20
        pts = [torch.from_numpy(rng.standard_normal(n)).cuda() for _ in range(dims)]
        p2, kw, stats = nat.exchange_particles(actx, comm, pts, mpb)
        assert stats["bytes_sent"] == 0 and stats["planned"]
        tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
        plain, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=mpb)
        assert_same_tree(actx.to_numpy(tree), actx.to_numpy(plain))
        num = nat.number_sharded_tree(actx, comm, tree)
        assert num["nboxes"] == int(tree.nboxes) and num["nsources"] == n
        assert num["source_offset"] == 0
        assert np.array_equal(num["box_ids"].cpu().numpy(), np.arange(int(tree.nboxes)))
        assert np.array_equal(num["global_level_start_box_nrs"],
                              actx.to_numpy(tree.level_start_box_nrs))
        let, info = nat.build_local_essential_tree(actx, comm, tree, num, well_sep_is_n_away=nway)
        assert info["halo_boxes_received"] == 0 and info["nboxes"] == int(tree.nboxes)
        assert np.array_equal(info["global_box_ids"].cpu().numpy(), np.arange(int(tree.nboxes)))
        h = actx.to_numpy(let)
        g = actx.to_numpy(tree)
        nb = g.nboxes
        assert np.array_equal(h.box_levels, g.box_levels) and np.array_equal(h.box_flags, g.box_flags)
        assert np.array_equal(h.box_centers[:, :nb], g.box_centers[:, :nb])
        assert np.array_equal(h.box_parent_ids, g.box_parent_ids)
        assert np.array_equal(h.box_child_ids[:, :nb], g.box_child_ids[:, :nb])
        tb = FMMTraversalBuilder(actx, well_sep_is_n_away=nway)
        t_let, _ = tb(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                      _active_level_ranges=info["active_level_ranges"])
        t_plain, _ = tb(actx, tree)
        assert_same_traversal(actx.to_numpy(t_let), actx.to_numpy(t_plain))
    finally:
        comm.close()


class _OneRank:           # the unique id needs no broadcast with one rank
    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def get_world_size():
        return 1


@pytest.mark.parametrize("dims,sep", [(3, False), (2, False), (3, True)])
def test_self_loopback_runs_the_point_to_point_branch(dims, sep):
    """world = 1 is all a one-GPU box can give RCCL, and there a rank has no peer: with the
    self-loopback switch its own segment of the particle all-to-all-v and an echo of its halo
    records go through ncclSend / ncclRecv (grouped, in rounds) instead of device copies.  The
    shard, the tree, the LET and the lists must be what they are without the switch."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    from compare import assert_same_traversal
    actx = HIPArrayContext(0)
    torch.cuda.set_device(0)
    comm = nat.rccl_comm(actx, _OneRank, self_loopback=True)
    try:
        rng = np.random.default_rng(11 + dims)
        n, nt, mpb = 180000, 70000, 25
        pts = [torch.from_numpy(rng.standard_normal(n)).cuda() for _ in range(dims)]
        es = 8
        if sep:
            tgts = [torch.from_numpy(rng.standard_normal(nt)).cuda() for _ in range(dims)]
            p2, t2, kw, stats = nat.exchange_particles(actx, comm, pts, mpb, targets=tgts)
            assert stats["bytes_sent"] == (n + nt) * dims * es
            for ax in range(dims):
                assert torch.equal(p2[ax], pts[ax]) and torch.equal(t2[ax], tgts[ax])
            tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, max_particles_in_box=mpb, **kw)
            plain, _ = TreeBuilder(actx)(actx, pts, targets=tgts, max_particles_in_box=mpb)
        else:
            p2, kw, stats = nat.exchange_particles(actx, comm, pts, mpb)
            assert stats["bytes_sent"] == n * dims * es and stats["rounds"] == 1
            for ax in range(dims):
                assert torch.equal(p2[ax], pts[ax])
            tree, _ = TreeBuilder(actx)(actx, p2, max_particles_in_box=mpb, **kw)
            plain, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=mpb)
        assert_same_tree(actx.to_numpy(tree), actx.to_numpy(plain))
        # particle identity through ncclSend / ncclRecv as well: with one rank the received order
        # is the chunk's, so the ids are 0..n-1 and a routed array arrives as it left
        route = stats["route"]
        assert torch.equal(route.global_ids("sources"), torch.arange(n, dtype=torch.int32, device="cuda"))
        assert route.chunk_offset["sources"] == 0 and route.n_global["sources"] == n
        assert route.n_sent["sources"] == n             # (the own segment travels too, here)
        vals = torch.from_numpy(rng.standard_normal(n)).cuda()
        owned = route.to_owners(vals, "sources")
        assert torch.equal(owned, vals) and torch.equal(route.to_callers(owned, "sources"), vals)
        if sep:
            i32 = torch.arange(nt, dtype=torch.int32, device="cuda") * 3
            assert torch.equal(route.global_ids("targets", dtype=torch.int64),
                               torch.arange(nt, dtype=torch.int64, device="cuda"))
            assert torch.equal(route.to_callers(route.to_owners(i32, "targets"), "targets"), i32)
        num = nat.number_sharded_tree(actx, comm, tree)
        let, info = nat.build_local_essential_tree(actx, comm, tree, num)
        lsb = actx.to_numpy(tree.level_start_box_nrs)
        k = stats["top_level"]
        deep = int(tree.nboxes) - int(lsb[min(k + 1, len(lsb) - 1)])
        assert deep > 0 and info["loopback_records"] == deep and info["loopback_mismatches"] == 0
        assert info["halo_boxes_received"] == 0 and info["nboxes"] == int(tree.nboxes)
        tb = FMMTraversalBuilder(actx)
        t_let, _ = tb(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                      _active_level_ranges=info["active_level_ranges"])
        t_plain, _ = tb(actx, tree)
        assert_same_traversal(actx.to_numpy(t_let), actx.to_numpy(t_plain))
    finally:
        comm.close()


def test_self_loopback_messages_above_the_round_limit():
    """A 1.7 GB message to oneself: four grouped ncclSend / ncclRecv rounds of at most 512 MiB
    (an unchunked message above 1 GB once arrived corrupted, LAB_NOTES.md section 6); every byte must arrive."""
    import torch
    from boxtree_amd import HIPArrayContext
    from boxtree_amd.distributed import native as nat
    actx = HIPArrayContext(0)
    torch.cuda.set_device(0)
    comm = nat.rccl_comm(actx, _OneRank, self_loopback=True)
    try:
        n = 72_000_000
        g = torch.Generator(device="cuda").manual_seed(5)
        pts = [torch.rand(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
        p2, kw, stats = nat.exchange_particles(actx, comm, pts, 64)
        assert stats["bytes_sent"] == n * 24 and stats["rounds"] == 4, stats
        for ax in range(3):
            assert torch.equal(p2[ax], pts[ax])
        assert stats["a2a_ms"] > 0
        # a routed 8-byte array to oneself: 576 MB, two rounds
        owned = stats["route"].to_owners(pts[1], "sources")
        assert torch.equal(owned, pts[1])
    finally:
        comm.close()


def test_rank_without_targets_keeps_the_collectives_in_step():
    """Target presence is a property of the job, not of a rank's chunk: a rank whose chunk has
    no target takes part in the two-set exchange (same all-reduce lengths, same flags of the
    shared top boxes) and may end up owning targets of the others."""
    import threading

    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    world, n, mpb = 3, 30000, 20
    rng = np.random.default_rng(77)
    chunks = [[rng.random(n) for _ in range(3)] for _ in range(world)]
    tchunks = [[rng.random(m) for _ in range(3)] for m in (25000, 0, 4000)]
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            out = nat.sharded_tree_and_lists(
                actx, comm, [torch.from_numpy(a).cuda() for a in chunks[rank]], mpb,
                targets=[torch.from_numpy(a).cuda() for a in tchunks[rank]])
            res[rank] = (out["numbering"]["nboxes"], int(out["tree"].nsources), int(out["tree"].ntargets),
                         out["numbering"]["ntargets"])
            comm.close()
        except BaseException as e:      # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    assert not errors, errors
    actx = HIPArrayContext(0)
    pts = [torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda() for ax in range(3)]
    tg = [torch.from_numpy(np.concatenate([c[ax] for c in tchunks])).cuda() for ax in range(3)]
    g, _ = TreeBuilder(actx)(actx, pts, targets=tg, max_particles_in_box=mpb)
    assert all(r[0] == int(g.nboxes) and r[3] == 29000 for r in res)
    assert sum(r[1] for r in res) == world * n and sum(r[2] for r in res) == 29000


def test_sharded_tree_and_lists_one_call():
    """boxtree_amd.distributed.native.sharded_tree_and_lists with two thread-ranks: the
    global box count of both ranks is the single-GPU tree's, every deep box is owned once."""
    import threading

    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    world, n, mpb = 2, 50000, 30
    chunks = [[np.random.default_rng(40 + r).standard_normal(n) for _ in range(3)] for r in range(world)]
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            out = nat.sharded_tree_and_lists(actx, comm, [torch.from_numpy(a).cuda() for a in chunks[rank]], mpb)
            res[rank] = (out["numbering"]["nboxes"], int(out["tree"].nsources),
                         int(out["let_info"]["target_boxes_mask"].sum()),
                         int(out["traversal"].target_boxes.shape[0]))
            comm.close()
        except BaseException as e:      # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    group.close()
    assert not errors, errors
    actx = HIPArrayContext(0)
    pts = [torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda() for ax in range(3)]
    g, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=mpb)
    assert res[0][0] == res[1][0] == int(g.nboxes)
    assert res[0][1] + res[1][1] == world * n
    assert all(r[3] > 0 and r[2] >= r[3] for r in res)


def test_failing_rank_does_not_hang_its_peers():
    """A rank that leaves a collective entry with an error (here: no plan on its context for
    bt_mgpu_number) tells the local group; the peer, waiting in the same collective, returns an
    error too instead of waiting forever."""
    import threading

    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder, _lib
    from boxtree_amd.distributed import native as nat
    group = nat.LocalGroup(2)
    outcome = [None, None]

    def run(rank):
        actx = HIPArrayContext(0)
        comm = group.comm(rank)
        pts = [torch.from_numpy(np.random.default_rng(rank).random(20000)).cuda() for _ in range(3)]
        try:
            if rank == 0:
                # skips the exchange: its context has no plan, bt_mgpu_number refuses
                tree, _ = TreeBuilder(actx)(actx, pts, max_particles_in_box=30)
                nat.number_sharded_tree(actx, comm, tree)
            else:
                nat.exchange_particles(actx, comm, pts, 30)
            outcome[rank] = "ok"
        except _lib.BoxtreeHipError as e:
            outcome[rank] = e.msg
        finally:
            comm.close()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert all(not t.is_alive() for t in threads), "a rank hangs"
    group.close()
    assert "no top-tree plan" in outcome[0]
    assert "peer rank" in outcome[1]


@pytest.mark.parametrize("case", ["empty_chunk", "own_nothing", "empty_sources_with_targets"])
def test_ranks_with_nothing_to_give_or_to_own(case):
    """A rank whose chunk is empty still takes part in every collective, and so does a rank
    that ends up owning no particle (more ranks than occupied top cells: its tree is the root
    box alone): all ranks arrive at the global tree's box count, no particle is lost."""
    import threading

    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    n, mpb = 20000, 20
    full = lambda r: [np.random.default_rng(50 + r).standard_normal(n) for _ in range(3)]    # noqa: E731
    empty = [np.zeros(0) for _ in range(3)]
    tchunks = None
    if case == "empty_chunk":
        chunks = [full(0), empty, full(2)]
    elif case == "own_nothing":
        def tiny(r):
            g = np.random.default_rng(70 + r)
            c = [0.3 + 1e-4 * g.random(3000) for _ in range(3)]
            if r == 0:
                for ax in range(3):
                    c[ax][0] = 0.0
                    c[ax][1] = 1.0
            return c
        chunks = [tiny(r) for r in range(6)]
    else:
        chunks = [empty, full(1)]
        tchunks = [full(5), full(6)]
    world = len(chunks)
    group = nat.LocalGroup(world)
    res, errors = [None] * world, []

    def run(rank):
        try:
            actx = HIPArrayContext(0)
            comm = group.comm(rank)
            kw = {}
            if tchunks is not None:
                kw["targets"] = [torch.from_numpy(a).cuda() for a in tchunks[rank]]
            out = nat.sharded_tree_and_lists(
                actx, comm, [torch.from_numpy(a).cuda() for a in chunks[rank]], mpb, **kw)
            res[rank] = (out["numbering"]["nboxes"], int(out["tree"].nsources), int(out["tree"].ntargets))
            comm.close()
        except BaseException as e:      # noqa: BLE001
            errors.append((rank, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    group.close()
    assert not errors, errors
    assert all(r is not None for r in res)
    actx = HIPArrayContext(0)
    cat = lambda cs: [torch.from_numpy(np.concatenate([c[ax] for c in cs])).cuda() for ax in range(3)]   # noqa: E731
    g, _ = TreeBuilder(actx)(actx, cat(chunks), targets=None if tchunks is None else cat(tchunks),
                             max_particles_in_box=mpb)
    assert all(r[0] == int(g.nboxes) for r in res), (res, int(g.nboxes))
    assert sum(r[1] for r in res) == sum(len(c[0]) for c in chunks)
    if tchunks is not None:
        assert sum(r[2] for r in res) == sum(len(c[0]) for c in tchunks)
    if case == "own_nothing":
        assert sum(1 for r in res if r[1] == 0) >= 3
