"""N>1 path on CPU: world_size-2 gloo run of the particle exchange that precedes
the per-rank tree build (boxtree_amd/distributed/__init__.py).  No GPU needed."""

import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, dims, with_targets, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import boxtree_amd.distributed as bd
        from boxtree_amd.distributed import exchange_particles, morton_cells
        if with_targets:
            # force the multi-round all-to-all (messages capped at 20 kB here)
            bd.A2A_MESSAGE_LIMIT_BYTES = 20000
        rng = np.random.default_rng(15 + rank)
        n = 20000 + 1000 * rank
        pts = [torch.from_numpy(rng.standard_normal(n)) for _ in range(dims)]
        tgts = None
        kw = {}
        if with_targets:
            tgts = [torch.from_numpy(rng.random(5000)) for _ in range(dims)]
            kw = {"target_radii": torch.from_numpy(2.0 ** rng.uniform(-10, 0, 5000) * 2.0 ** -7),
                  "stick_out_factor": 0.25}
        newp, newt, nkw, st = exchange_particles(None, dist, pts, tgts, kw, return_plan=True)
        cells = morton_cells(newp, st["bbox_min"], st["bbox_max"], st["top_level"]).numpy()
        # particle identity: global ids of what arrived, per-particle arrays to
        # the owners and back over the exchange's own plan -- several rounds with the 20-kB limit
        route = st["route"]
        gids = route.global_ids("sources").numpy()
        back = route.to_callers(route.to_owners(pts[0], "sources"), "sources")
        ident = dict(gids=gids, x_owned=route.to_owners(pts[0], "sources").numpy(),
                     round_trip=bool(torch.equal(back, pts[0])), offset=route.chunk_offset("sources"))
        if with_targets:
            ident["tgids"] = route.global_ids("targets", dtype=torch.int64).numpy()
            ident["t_round_trip"] = bool(torch.equal(
                route.to_callers(route.to_owners(tgts[1], "targets"), "targets"), tgts[1]))
        res = dict(
            rank=rank, ident=ident,
            sent=np.stack([p.numpy() for p in pts]),
            got=np.stack([p.numpy() for p in newp]),
            owner=st["owner"],
            bbox=np.stack([nkw["_root_box"][0], nkw["_root_box"][1]], axis=1), cells=cells,
            tsent=None if tgts is None else np.stack([t.numpy() for t in tgts]
                                                     + [kw["target_radii"].numpy()]),
            tgot=None if newt is None else np.stack([t.numpy() for t in newt]
                                                    + [nkw["target_radii"].numpy()]),
        )
        q.put(res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,with_targets", [(3, False), (2, True)])
def test_exchange_world2(dims, with_targets):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, with_targets, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda r: r["rank"])

    # identical plan on all ranks
    assert np.array_equal(results[0]["owner"], results[1]["owner"])
    assert np.array_equal(results[0]["bbox"], results[1]["bbox"])
    owner = results[0]["owner"]
    assert np.all(np.diff(owner) >= 0)          # contiguous Morton ranges
    assert set(owner.tolist()) <= {0, 1}

    # nothing lost, nothing duplicated
    def rows(a):
        a = np.ascontiguousarray(a.T)
        return a[np.lexsort(a.T[::-1])]
    sent = np.concatenate([r["sent"] for r in results], axis=1)
    got = np.concatenate([r["got"] for r in results], axis=1)
    assert np.array_equal(rows(sent), rows(got))
    if with_targets:
        tsent = np.concatenate([r["tsent"] for r in results], axis=1)
        tgot = np.concatenate([r["tgot"] for r in results], axis=1)
        assert np.array_equal(rows(tsent), rows(tgot))

    # every rank received exactly the particles of the cells it owns
    for r in results:
        assert np.all(owner[r["cells"]] == r["rank"])
    # particle identity: received particle j is particle gids[j] of the concatenated chunks;
    # ids ascend within what one sender sent (stable exchange), arrays come home unchanged
    off = 0
    for r in results:
        idn = r["ident"]
        assert idn["gids"].dtype == np.int32 and idn["round_trip"]
        assert idn["offset"] == (off, sent.shape[1])
        off += r["sent"].shape[1]
        assert np.array_equal(sent[:, idn["gids"]], r["got"])
        assert np.array_equal(sent[0, idn["gids"]], idn["x_owned"])
        if with_targets:
            assert idn["tgids"].dtype == np.int64 and idn["t_round_trip"]
            assert np.array_equal(tsent[:, idn["tgids"]], r["tgot"])
    assert np.array_equal(np.sort(np.concatenate([r["ident"]["gids"] for r in results])),
                          np.arange(sent.shape[1]))
    # root box covers everything and is square
    bbox = results[0]["bbox"]
    assert np.all(bbox[:, 0] <= sent.min(axis=1)) and np.all(bbox[:, 1] > sent.max(axis=1))
    ext = bbox[:, 1] - bbox[:, 0]
    assert np.all(np.abs(ext - ext[0]) < 1e-14)
    # balance: within 25% of even
    sizes = [r["got"].shape[1] for r in results]
    assert max(sizes) < 1.25 * sum(sizes) / world + 1


def test_partition_cells_is_deterministic():
    from boxtree_amd.distributed import partition_cells
    hist = np.random.default_rng(0).integers(0, 1000, 4096)
    o1 = partition_cells(hist, 8)
    o2 = partition_cells(hist.copy(), 8)
    assert np.array_equal(o1, o2)
    assert np.all(np.diff(o1) >= 0) and o1.min() == 0 and o1.max() == 7
    loads = np.bincount(o1, weights=hist, minlength=8)
    assert loads.max() < 1.2 * hist.sum() / 8


def _reference_top_tree(hist, dims, k, mpb):
    """Plain recursive statement of the top of the adaptive tree."""
    C = 1 << dims
    boxes = {0: [0]}        # level -> paths of existing boxes

    def count(level, path):
        sh = dims * (k - level)
        return int(hist[path << sh:(path + 1) << sh].sum())

    leaves = []             # (level, path) of frontier boxes
    for lev in range(k):
        nxt = []
        for path in boxes[lev]:
            if count(lev, path) > mpb:
                for m in range(C):
                    child = path * C + m
                    if count(lev + 1, child) > 0:
                        nxt.append(child)
            else:
                leaves.append((lev, path))
        boxes[lev + 1] = nxt
    leaves += [(k, p) for p in boxes[k]]
    return boxes, leaves


@pytest.mark.parametrize("dims,k", [(2, 4), (3, 3)])
def test_top_tree_plan_and_numbering(dims, k):
    from boxtree_amd.distributed import global_box_numbering, partition_cells, top_tree_plan
    rng = np.random.default_rng(3)
    C = 1 << dims
    hist = rng.integers(0, 40, C ** k)
    hist[rng.random(C ** k) < 0.5] = 0
    hist[: C ** k // 4] = rng.integers(0, 2, C ** k // 4)       # a sparse quarter
    mpb = 30
    plan = top_tree_plan(hist, dims, k, mpb)
    boxes, leaves = _reference_top_tree(hist, dims, k, mpb)
    for lev in range(k + 1):
        assert np.nonzero(plan["exists"][lev])[0].tolist() == boxes[lev]
        assert plan["nboxes"][lev] == len(boxes[lev])
        assert plan["index"][lev][boxes[lev]].tolist() == list(range(len(boxes[lev])))
    # every frontier box is one ownership unit
    world = 3
    owner = partition_cells(hist, world, plan["unit_start"])
    assert np.all(np.diff(owner) >= 0)
    for lev, path in leaves:
        sh = dims * (k - lev)
        assert len(set(owner[path << sh:(path + 1) << sh].tolist())) == 1
        assert np.all(plan["unit_start"][path << sh:(path + 1) << sh] == path << sh)
    assert plan["cell_prefix"][-1] == hist.sum() and plan["cell_prefix"][0] == 0

    # numbering: shared top levels by plan, deep levels rank-major
    lc = np.zeros((world, 64), np.int64)
    for r in range(world):
        lc[r, :k + 1] = 1                      # (content of the top levels is not used)
        lc[r, k + 1:k + 4] = rng.integers(1, 50, 3)
    lc[2, k + 3] = 0                           # rank 2's tree is one level shallower
    for r in range(world):
        starts, deep = global_box_numbering(plan, lc, r)
        assert len(starts) == k + 5
        for lev in range(k + 1):
            assert starts[lev + 1] - starts[lev] == len(boxes[lev])
        for lev in range(k + 1, k + 4):
            assert starts[lev + 1] - starts[lev] == lc[:, lev].sum()
            assert deep[lev] == starts[lev] + lc[:r, lev].sum()


def _plan_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from boxtree_amd.distributed import exchange_particles, morton_cells
        rng = np.random.default_rng(40 + rank)
        n = 30000
        # a dense blob and a thin background: light top cells exist
        pts = [torch.from_numpy(np.concatenate([0.05 * rng.standard_normal(n - 300) + 0.6,
                                                rng.random(300)])) for _ in range(3)]
        newp, _, nkw, st = exchange_particles(None, dist, pts, None, {}, top_level=3,
                                              return_plan=True, max_particles_in_box=30)
        cells = morton_cells(newp, st["bbox_min"], st["bbox_max"], 3).numpy()
        plan = st["plan"]
        q.put(dict(rank=rank, owner=st["owner"], cells=cells, unit_start=plan["unit_start"],
                   total=int(plan["cell_prefix"][-1]), n=len(newp[0]),
                   has_top=("_top_tree" in nkw)))
    finally:
        dist.destroy_process_group()


def test_exchange_with_top_tree_plan_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = results
    assert np.array_equal(a["owner"], b["owner"]) and np.array_equal(a["unit_start"], b["unit_start"])
    assert a["total"] == b["total"] == 60000 == a["n"] + b["n"]
    owner, unit = a["owner"], a["unit_start"]
    assert np.all(np.diff(owner) >= 0)
    assert np.array_equal(owner, owner[unit])       # frontier boxes are not cut
    for r in results:
        assert np.all(owner[r["cells"]] == r["rank"])
        assert not r["has_top"]                     # device prefix only on the GPU path


def _gather_worker(rank, world, port, q):
    """Steps 5+6 over gloo with CPU-built local trees (the oracle stands in for the
    device build here; the device build itself is checked in the gpu tests)."""
    import sys
    from types import SimpleNamespace

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import oracle
        from boxtree_amd.distributed import (gather_global_box_tree, morton_cells,
                                             number_sharded_tree, partition_cells,
                                             top_tree_plan)
        dims, k, mpb = 2, 2, 30
        rng = np.random.default_rng(77)
        allpts = [rng.random(20000) * 1.9 + 0.05 for _ in range(dims)]     # same on all ranks
        bbox = np.array([[0.0, 2.0]] * dims)
        tp = [torch.from_numpy(a) for a in allpts]
        cells = morton_cells(tp, bbox[:, 0], bbox[:, 1], k).numpy()
        hist = np.bincount(cells, minlength=4 ** k)
        assert hist.min() > mpb            # every top cell is heavy: no top-tree hint needed
        plan = top_tree_plan(hist, dims, k, mpb)
        owner = partition_cells(hist, world, plan["unit_start"])
        mine = owner[cells] == rank
        ot = oracle.build_tree([a[mine] for a in allpts], max_particles_in_box=mpb, bbox=bbox)
        tree = SimpleNamespace(
            dimensions=dims, nboxes=ot.nboxes, nsources=ot.nsources, ntargets=ot.ntargets,
            level_start_box_nrs=ot.level_start_box_nrs, root_extent=ot.root_extent,
            stick_out_factor=ot.stick_out_factor,
            box_centers=torch.from_numpy(ot.box_centers), box_levels=torch.from_numpy(ot.box_levels),
            box_flags=torch.from_numpy(ot.box_flags),
            box_parent_ids=torch.from_numpy(ot.box_parent_ids),
            box_child_ids=torch.from_numpy(ot.box_child_ids))
        stats = dict(plan=plan, bbox_min=bbox[:, 0], root_extent=ot.root_extent)
        num = number_sharded_tree(dist, tree, stats)
        gt = gather_global_box_tree(None, dist, tree, num)
        res = dict(rank=rank, nboxes=num["nboxes"], starts=num["global_level_start_box_nrs"],
                   centers=gt.box_centers.numpy(), levels=gt.box_levels.numpy(),
                   flags=gt.box_flags.numpy(), parents=gt.box_parent_ids.numpy(),
                   children=gt.box_child_ids.numpy(), src_off=num["source_offset"],
                   ranges=num["active_level_ranges"])
        if rank == 0:
            g = oracle.build_tree(allpts, max_particles_in_box=mpb, bbox=bbox)
            res["g"] = dict(nboxes=g.nboxes, starts=g.level_start_box_nrs,
                            centers=g.box_centers, levels=g.box_levels, flags=g.box_flags,
                            parents=g.box_parent_ids, children=g.box_child_ids)
        q.put(res)
    finally:
        dist.destroy_process_group()


def test_number_and_gather_world2(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = results[0]["g"]
    for r in results:
        # every rank holds the box arrays of the tree one process builds from all points
        assert r["nboxes"] == g["nboxes"]
        assert np.array_equal(r["starts"], g["starts"])
        for name in ("centers", "levels", "flags", "parents", "children"):
            assert np.array_equal(r[name], g[name]), name
    assert results[0]["src_off"] == 0 and results[1]["src_off"] > 0
    # the ranks' deep ranges tile every level
    r0, r1 = results[0]["ranges"], results[1]["ranges"]
    for lev in range(3, len(r0)):
        assert r0[lev][1] == r1[lev][0]
        assert r0[lev][0] == g["starts"][lev] and r1[lev][1] == g["starts"][lev + 1]


def _chunk_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from boxtree_amd.distributed import all_to_all_chunked
        rng = np.random.default_rng(rank)
        s_split = [[700, 1301, 0], [5, 2048, 999], [1234, 1, 77]][rank]
        send = torch.from_numpy(rng.random(sum(s_split)))
        counts = torch.tensor(s_split)
        rc = torch.empty_like(counts)
        dist.all_to_all_single(rc, counts)
        r_split = rc.tolist()
        want = torch.empty(sum(r_split), dtype=send.dtype)
        dist.all_to_all_single(want, send, r_split, s_split)
        res = {}
        for limit in (8 * 100, 8 * 333, 8 * 5000):
            got = torch.full_like(want, -1.0)
            rounds = all_to_all_chunked(dist, got, send, r_split, s_split, limit_bytes=limit)
            res[limit] = (rounds, bool(torch.equal(got, want)))
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_all_to_all_chunked_world3():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results.items():
        for limit, (rounds, ok) in res.items():
            assert ok, (rank, limit)
        assert res[8 * 100][0] >= 13 and res[8 * 5000][0] == 1


def test_native_particle_route_refuses_to_outlive_its_exchange():
    """The library keeps the plan of a context's LATEST exchange only and takes no array lengths
    (bt_mgpu_route): a ParticleRoute kept across a second exchange must raise before it reaches the
    library, not move the new plan's counts through buffers sized for the old one.  Also: the id
    width follows the global particle count (int32 = the reference's particle_id_t while it fits)."""
    import types

    import torch
    from boxtree_amd.distributed import native as nat

    class Ctx:
        _mgpu_exchange_serial = 3
        device_index = 0

        def sync_in(self):
            raise AssertionError("reached the library")

    def shard(total):
        return types.SimpleNamespace(n_owned=10, n_owned_targets=0, source_chunk_offset=0, target_chunk_offset=0,
                                     n_global_sources=total, n_global_targets=0, n_sent_sources=4,
                                     n_sent_targets=0)

    actx = Ctx()
    route = nat.ParticleRoute(actx, None, shard(100), 12, None)
    assert route.serial == 3
    actx._mgpu_exchange_serial = 4          # another exchange on the same context
    with pytest.raises(RuntimeError, match="another exchange"):
        route.to_owners(torch.zeros(12, dtype=torch.float64))
    with pytest.raises(RuntimeError, match="another exchange"):
        route.to_callers(torch.zeros(10, dtype=torch.float64))
    with pytest.raises(RuntimeError, match="another exchange"):
        route.global_ids("sources")
    # a current route gets as far as the library (the stub's sync_in) -- with the id width chosen
    # by the global count
    for total, want in ((2**31 - 1, torch.int32), (2**31, torch.int64)):
        route = nat.ParticleRoute(actx, None, shard(total), 12, None)
        seen = {}
        real_empty = torch.empty

        def spy(n, dtype=None, device=None):
            seen["dtype"] = dtype
            return real_empty(n, dtype=dtype)

        torch.empty = spy
        try:
            with pytest.raises(AssertionError, match="reached the library"):
                route.global_ids("sources")
        finally:
            torch.empty = real_empty
        assert seen["dtype"] == want
