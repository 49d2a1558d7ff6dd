"""CPU side of the distributed FMM evaluation (boxtree_amd/distributed/calculation.py,
partition rules restated in oracle/oracle.py): the stage structure of the sparse
multipole all-reduce, the same exchange on a real ``gloo`` process group (world 3
and 5, odd sizes exercise the unpaired rank), and properties of the partition."""

import os
import socket
from types import SimpleNamespace

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("size", list(range(1, 24)) + [32, 33, 64])
def test_reduce_scatter_stage_structure(size):
    from boxtree_amd.distributed.calculation import (reduce_scatter_num_stages,
                                                     reduce_scatter_stage)
    rng = np.random.default_rng(size)
    nboxes = 200
    users = rng.random((size, nboxes)) < 0.3            # users[r, b]: r reads box b
    contrib = rng.random((size, nboxes)) < 0.4
    vals = np.where(contrib, rng.integers(1, 100, (size, nboxes)), 0).astype(np.int64)
    want = vals.sum(axis=0)
    have = vals.copy()
    holds = contrib.copy()
    ranges = [(0, size)] * size
    nstages = reduce_scatter_num_stages(size)
    assert nstages == int(np.ceil(np.log2(size))) if size > 1 else nstages == 0
    for _ in range(nstages):
        msgs = []
        nxt = list(ranges)
        plan = {}
        for r in range(size):
            left, right = ranges[r]
            if right - left > 1:
                sinks, sources, (ulo, uhi), nxt[r] = reduce_scatter_stage(r, left, right)
                plan[r] = sources
                assert len(sinks) == 1 and sinks[0] != r
                sel = holds[r] & users[ulo:uhi].any(axis=0)
                msgs.append((r, sinks[0], np.nonzero(sel)[0], have[r][sel].copy()))
                # what is sent is used on the other side of the cut only
                lo, hi = nxt[r]
                assert (ulo, uhi) == ((left + right) // 2, right) or (ulo, uhi) == (left, (left + right) // 2)
                assert not (lo <= sinks[0] < hi)
        for r, sources in plan.items():
            assert sorted(m[0] for m in msgs if m[1] == r) == sorted(sources)
        for _src, dst, boxes, v in msgs:
            have[dst][boxes] += v
            holds[dst][boxes] = True
        ranges = nxt
    assert all(right - left == 1 for left, right in ranges)
    for r in range(size):
        assert np.array_equal(have[r][users[r]], want[users[r]]), r


def _mpole_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from boxtree_amd.distributed.calculation import DistributedExpansionWranglerMixin
        from oracle import oracle as orc
        actx = SimpleNamespace(torch=torch, device=torch.device("cpu"))
        rng = np.random.default_rng(77)            # same stream on every rank
        nboxes, ncoeff = 3000, 3
        users = (rng.random((world, nboxes)) < 0.25).astype(np.int8)
        contrib = rng.random((world, nboxes)) < 0.3
        vals = rng.integers(1, 1000, (world, nboxes, ncoeff)).astype(np.float64)
        vals[~contrib] = 0
        ustarts, ulists = orc.box_to_user_ranks(users)

        class W(DistributedExpansionWranglerMixin):
            def _boxes_used_by(self, actx, contributing, subrange):
                c = contributing.numpy().astype(bool)
                sel = c & users[subrange[0]:subrange[1]].any(axis=0)
                return torch.from_numpy(np.nonzero(sel)[0].astype(np.int32))

        w = W()
        w.comm = dist
        resp = np.nonzero(contrib[rank])[0].astype(np.int32)
        tree = SimpleNamespace(
            nboxes=nboxes, ancestor_mask=torch.zeros(nboxes, dtype=torch.int8),
            responsible_boxes_list=torch.from_numpy(resp),
            box_to_user_rank_starts=torch.from_numpy(ustarts),
            box_to_user_rank_lists=torch.from_numpy(ulists), nsources=100 + rank,
            ntargets=50 + 2 * rank)
        w.traversal = SimpleNamespace(tree=tree)
        w.global_traversal = SimpleNamespace(
            tree=SimpleNamespace(ntargets=sum(50 + 2 * r for r in range(world))))

        mp_sparse = torch.from_numpy(vals[rank].copy())
        stats = w.communicate_mpoles(actx, mp_sparse, return_stats=True)
        w.communicate_mpoles_via_allreduce = True
        mp_all = torch.from_numpy(vals[rank].copy())
        w.communicate_mpoles(actx, mp_all)
        want = vals.sum(axis=0)
        mine = users[rank].astype(bool)
        ok_sparse = bool(np.array_equal(mp_sparse.numpy()[mine], want[mine]))
        ok_all = bool(np.array_equal(mp_all.numpy(), want))

        # weights out, potentials back
        nsrc = [100 + r for r in range(world)]
        ntgt = [50 + 2 * r for r in range(world)]
        gw = np.arange(1000, dtype=np.float64)
        src_idx = [torch.from_numpy(np.random.default_rng(r).choice(1000, nsrc[r], replace=False)
                                    .astype(np.int32)) for r in range(world)]
        got_w = w.distribute_source_weights(
            actx, [torch.from_numpy(gw)] if rank == 0 else [None],
            src_idx if rank == 0 else None)
        ok_w = len(got_w) == 1 and bool(np.array_equal(got_w[0].numpy(),
                                                       gw[src_idx[rank].numpy()]))
        off = np.concatenate([[0], np.cumsum(ntgt)])
        tgt_idx = [torch.arange(off[r], off[r + 1], dtype=torch.int32) for r in range(world)]
        pot = torch.full((ntgt[rank],), float(rank + 1), dtype=torch.float64)
        gathered = w.gather_potential_results(actx, pot, tgt_idx if rank == 0 else None)
        if rank == 0:
            ok_p = bool(np.array_equal(gathered.numpy(),
                                       np.repeat(np.arange(1, world + 1.0), ntgt)))
        else:
            ok_p = gathered is None
        q.put((rank, dict(ok_sparse=ok_sparse, ok_all=ok_all, ok_w=ok_w, ok_p=ok_p,
                          sent=sum(stats["bytes_sent_by_stage"]),
                          dense=nboxes * (ncoeff * 8 + 4) * len(stats["bytes_sent_by_stage"]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_communicate_mpoles_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mpole_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results.items():
        assert res["ok_sparse"] and res["ok_all"] and res["ok_w"] and res["ok_p"], (rank, res)
        assert res["sent"] < res["dense"]         # the sparse exchange moves fewer boxes


@pytest.mark.parametrize("dims", [2, 3])
def test_partition_rules_on_oracle_trees(oracle, dims):
    rng = np.random.default_rng(4)
    pts = [rng.standard_normal(4000) for _ in range(dims)]
    tree = oracle.build_tree(pts, max_particles_in_box=20)
    trav = oracle.build_traversal(tree)
    order = oracle.dfs_order(tree)
    assert sorted(order) == list(range(tree.nboxes)) and order[0] == 0
    pos = np.empty(tree.nboxes, np.int64)
    pos[order] = np.arange(tree.nboxes)
    # preorder: a parent comes first, siblings in descending child number
    for b in range(1, tree.nboxes):
        assert pos[tree.box_parent_ids[b]] < pos[b]
    kids = [c for c in tree.box_child_ids[:, 0] if c > 0]
    assert all(pos[a] > pos[b] for a, b in zip(kids, kids[1:]))

    cost = rng.integers(0, 50, tree.nboxes).astype(np.float64)
    for nranks in (1, 2, 5, 16):
        seg = oracle.partition_work_segments(cost, order, nranks)
        assert seg[0, 0] == 0 and seg[-1, 1] == tree.nboxes
        assert np.all(seg[1:, 0] == seg[:-1, 1])
    with pytest.raises(RuntimeError):
        oracle.partition_work_segments(cost[:3], order[:3], 4)

    resp = order[tree.nboxes // 4: tree.nboxes // 2]
    m = oracle.box_masks(tree, trav, resp)
    assert m["responsible_boxes"].sum() == len(resp)
    # ancestors are exactly the union of the parent chains
    anc = set()
    for b in resp:
        while b != 0:
            b = int(tree.box_parent_ids[b])
            anc.add(b)
    assert set(np.nonzero(m["ancestor_boxes"])[0]) == anc
    assert np.all(m["point_src_boxes"][resp] == 1)
    ls, ln, lc, idx = oracle.local_particles_and_lists(
        m["point_src_boxes"], tree.box_source_starts, tree.box_source_counts_nonchild,
        tree.box_source_counts_cumul, tree.nsources)
    assert lc[0] == len(idx) == ln.sum()
