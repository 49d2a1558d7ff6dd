"""Particle identity across the sharded build: the received records
carry no ids; the library keeps the exchange's send plan and routes any per-particle array over it
(``bt_mgpu_route`` / ``bt_mgpu_global_ids``).  What the reference keeps for the same purpose:
``src_idx`` / ``tgt_idx`` (boxtree/distributed/__init__.py:238-248), used to hand out source
weights and to collect potentials (distributed/calculation.py:86-142).

Everything compared here is the LIBRARY's output: the tests keep no id bookkeeping of their own.
Ranks are threads over a local communicator (RCCL refuses two ranks on one GPU)."""

import numpy as np
import pytest

from test_gpu_mgpu_extents import make_chunks, run_ranks

pytestmark = pytest.mark.gpu


def csr_rows_sum(starts, lists, values, name="?"):
    """out[i] = sum of values[lists[starts[i]:starts[i+1]]] (exact for integers in float64)."""
    starts = np.asarray(starts, np.int64)
    lists = np.asarray(lists, np.int64)
    out = np.zeros(len(starts) - 1, np.float64)
    assert len(lists) == starts[-1], f"{name}: {len(lists)} entries, starts end at {starts[-1]}"
    assert len(lists) == 0 or (lists.min() >= 0 and lists.max() < len(values)), \
        f"{name}: entries outside [0, {len(values)}): {lists.min()} .. {lists.max()}"
    if len(lists):
        seg = np.repeat(np.arange(len(starts) - 1), np.diff(starts))
        np.add.at(out, seg, values[np.asarray(lists, np.int64)])
    return out


def constant_one_on_boxes(trav, box_levels, box_parent_ids, own_w, cum_w, nlevels):
    """The constant-one FMM (boxtree/constant_one.py:49-237 driven by fmm.py:380-532) on box
    sums: *own_w* = weight of a box's own sources, *cum_w* = of its whole subtree.  Returns
    (direct part per target box [len(target_boxes)], local expansion per box [nboxes]); a target
    in target box b gets direct[b] + local[b]."""
    nb = len(box_levels)
    direct = csr_rows_sum(trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists, own_w, 'list 1')
    local = np.zeros(nb, np.float64)
    tp = np.asarray(trav.target_or_target_parent_boxes, np.int64)
    local[tp] += csr_rows_sum(trav.from_sep_siblings_starts, trav.from_sep_siblings_lists, cum_w, 'list 2')
    local[tp] += csr_rows_sum(trav.from_sep_bigger_starts, trav.from_sep_bigger_lists, own_w, 'list 4')
    tb = np.asarray(trav.target_boxes, np.int64)
    pos = np.full(nb, -1, np.int64)
    pos[tb] = np.arange(len(tb))
    for lev in range(nlevels):
        bl = trav.from_sep_smaller_by_level[lev]
        tbl = np.asarray(trav.target_boxes_sep_smaller_by_source_level[lev], np.int64)
        if len(tbl):
            direct[pos[tbl]] += csr_rows_sum(bl.starts, bl.lists, cum_w, f'list 3 level {lev}')
    for name in ("from_sep_close_smaller", "from_sep_close_bigger"):
        st = getattr(trav, name + "_starts", None)
        if st is not None:
            direct += csr_rows_sum(st, getattr(trav, name + "_lists"), own_w, name)
    # downward pass over the boxes that have lists here
    have = np.zeros(nb, bool)
    have[tp] = True
    for lev in range(1, nlevels):
        idx = np.nonzero((box_levels == lev) & have)[0]
        local[idx] += local[box_parent_ids[idx]]
    return direct, local


@pytest.mark.parametrize("dims,world,dist_kind,mode", [
    (3, 2, "uniform", "points"), (3, 3, "blob", "points"), (2, 5, "normal", "points"),
    (3, 8, "uniform", "points"), (2, 8, "blob", "points"),
    # (more than 8 owners: the partition ranks with LDS counters instead of one ballot per owner)
    (3, 12, "uniform", "points"), (2, 19, "normal", "targets"),
    (3, 3, "uniform", "targets"), (2, 4, "blob", "targets"), (3, 8, "normal", "targets"),
    (3, 2, "uniform", "extents"), (3, 5, "blob", "extents"), (2, 8, "normal", "extents")])
def test_identity_ids_and_sharded_fmm(dims, world, dist_kind, mode):
    check_identity(dims, world, dist_kind, mode)


def check_identity(dims, world, dist_kind, mode, n_src=24000, n_tgt=5000, mpb=20, seed=900, sof=0.25,
                   norm="linf"):
    """(1) bt_mgpu_global_ids + the rank's user_source_ids reproduce the single-GPU tree's
    user_source_ids slice; sorted_target_ids likewise through bt_mgpu_route; (2) arrays routed
    to the owners and back are unchanged and land beside their particles; (3) a constant-one FMM
    on the ranks' local essential trees, with random integer source weights handed in in the
    CALLER's order and potentials collected in the caller's order, equals the single-GPU run --
    and so does its List-1 part alone, which (unlike the full sum) depends on which source
    carries which weight."""
    import torch
    from boxtree_amd import FMMTraversalBuilder, HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    src, tgt, rad = make_chunks(world, dims, n_src, n_tgt, seed + world, dist_kind,
                                0.4 if dist_kind == "normal" else 0.05)
    # ragged chunks: rank r gives away a different number of particles, one rank (of > 2) none
    for r in range(world):
        keep = max(n_src - (n_src // 25) * r, 1) if not (world > 2 and r == 1) else 0
        src[r] = [a[:keep] for a in src[r]]
    sep = mode != "points"
    ext = mode == "extents"
    wrng = np.random.default_rng(5)
    weights = [wrng.integers(1, 1000, len(src[r][0])).astype(np.float64) for r in range(world)]
    group = nat.LocalGroup(world)

    def rank_fn(rank):
        actx = HIPArrayContext(0)
        comm = group.comm(rank)
        p = [torch.from_numpy(a).cuda() for a in src[rank]]
        kw_x = {}
        if sep:
            kw_x["targets"] = [torch.from_numpy(a).cuda() for a in tgt[rank]]
        if ext:
            kw_x.update(target_radii=torch.from_numpy(rad[rank]).cuda(), stick_out_factor=sof,
                        extent_norm=norm)
        res = nat.sharded_tree_and_lists(actx, comm, p, mpb, **kw_x)
        tree, num, let, info, trav, route = (res[k] for k in ("tree", "numbering", "let", "let_info",
                                                              "traversal", "route"))
        out = dict(num={k: num[k] for k in ("source_offset", "target_offset", "nboxes")})
        # (1) the two index arrays of the global tree, this rank's share
        out["user_source_ids"] = route.global_user_source_ids(tree).cpu().numpy()
        out["sorted_target_ids"] = route.global_sorted_target_ids(tree, num["target_offset"]).cpu().numpy()
        out["gids64"] = route.global_ids("sources", dtype=torch.int64).cpu().numpy()
        out["chunk_offset"] = dict(route.chunk_offset)
        out["n_global"] = dict(route.n_global)
        # (2) a coordinate routed like any other array arrives beside its particle; round trips
        x_owned = route.to_owners(p[0], "sources")
        out["x_owned"], out["x_recv"] = x_owned.cpu().numpy(), tree.sources[0].cpu().numpy()
        out["x_usid"] = tree.user_source_ids.cpu().numpy()
        back = route.to_callers(x_owned, "sources")
        out["round_trip_f64"] = bool(torch.equal(back, p[0]))
        i32 = torch.arange(len(p[0]), dtype=torch.int32, device="cuda") * 7 + rank
        out["round_trip_i32"] = bool(torch.equal(route.to_callers(route.to_owners(i32, "sources"), "sources"), i32))
        if sep:
            t0 = kw_x["targets"][0]
            out["round_trip_tgt"] = bool(torch.equal(route.to_callers(route.to_owners(t0, "targets"), "targets"), t0))
        # (3) source weights: caller's order -> owners -> tree order
        w_tree = route.to_owners(torch.from_numpy(weights[rank]).cuda(), "sources")[tree.user_source_ids.long()]
        h = actx.to_numpy(tree)
        wt = w_tree.cpu().numpy()
        csum = np.concatenate([[0.0], np.cumsum(wt)])
        own_w = csum[h.box_source_starts + h.box_source_counts_nonchild] - csum[h.box_source_starts]
        cum_w = csum[h.box_source_starts + h.box_source_counts_cumul] - csum[h.box_source_starts]
        out.update(own_w=own_w, cum_w=cum_w, local_gid=num["box_ids"].cpu().numpy().astype(np.int64),
                   local_levels=h.box_levels, tree=h, let=actx.to_numpy(let), trav=actx.to_numpy(trav),
                   let_gid=info["global_box_ids"].cpu().numpy().astype(np.int64),
                   mask=info["target_boxes_mask"].cpu().numpy(), route=route, actx=actx, comm=comm,
                   dev_tree=tree)
        return out

    results = run_ranks(world, rank_fn)

    actx = HIPArrayContext(0)
    cat = lambda chunks, ax: torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()  # noqa: E731
    allsrc = [cat(src, ax) for ax in range(dims)]
    gkw = {}
    if sep:
        gkw["targets"] = [cat(tgt, ax) for ax in range(dims)]
    if ext:
        gkw.update(target_radii=torch.from_numpy(np.concatenate(rad)).cuda(), stick_out_factor=sof,
                   extent_norm=norm)
    gt, _ = TreeBuilder(actx)(actx, allsrc, max_particles_in_box=mpb, **gkw)
    g = actx.to_numpy(gt)
    gtrav = actx.to_numpy(FMMTraversalBuilder(actx)(actx, gt)[0])
    n_all = sum(len(s[0]) for s in src)
    nt_all = sum(len(t[0]) for t in tgt) if sep else n_all

    # ---- (1) ids ---------------------------------------------------------------------------------
    off = 0
    sorted_tid = []
    for r, res in enumerate(results):
        ns = len(res["user_source_ids"])
        so = res["num"]["source_offset"]
        assert np.array_equal(g.user_source_ids[so:so + ns], res["user_source_ids"]), f"rank {r}"
        assert res["user_source_ids"].dtype == np.int32
        assert res["chunk_offset"]["sources"] == off and res["n_global"]["sources"] == n_all
        if sep:
            assert res["n_global"]["targets"] == nt_all
        off += len(src[r][0])
        sorted_tid.append(res["sorted_target_ids"])
        assert res["gids64"].dtype == np.int64
        # received particle j IS global particle gids[j]
        assert np.array_equal(np.concatenate([c[0] for c in src])[res["gids64"]], res["x_owned"])
        assert np.array_equal(res["x_owned"][res["x_usid"]], res["x_recv"])
        assert res["round_trip_f64"] and res["round_trip_i32"] and res.get("round_trip_tgt", True)
    assert np.array_equal(np.concatenate(sorted_tid), g.sorted_target_ids)

    # ---- (3) constant-one FMM with weights in the caller's order ------------------------------------
    w_all = np.concatenate(weights)
    gw = w_all[g.user_source_ids]
    gsum = np.concatenate([[0.0], np.cumsum(gw)])
    g_own = gsum[g.box_source_starts + g.box_source_counts_nonchild] - gsum[g.box_source_starts]
    g_cum = gsum[g.box_source_starts + g.box_source_counts_cumul] - gsum[g.box_source_starts]
    # box sums by global box number, assembled from the ranks (the "multipole exchange": deep boxes
    # have one owner, shared top boxes are summed) -- they must be the single-GPU tree's
    own_glob = np.zeros(g.nboxes)
    cum_glob = np.zeros(g.nboxes)
    for res in results:
        np.add.at(own_glob, res["local_gid"], res["own_w"])
        np.add.at(cum_glob, res["local_gid"], res["cum_w"])
    assert np.array_equal(own_glob, g_own) and np.array_equal(cum_glob, g_cum)

    def per_target(trav, tree, direct, local, near_only):
        """potential per target in TREE order"""
        out = np.zeros(tree.ntargets)
        tb = np.asarray(trav.target_boxes, np.int64)
        val = direct if near_only else direct + local[tb]
        st, cn = tree.box_target_starts[tb], tree.box_target_counts_nonchild[tb]
        for s, c, v in zip(st, cn, val):
            out[s:s + c] = v
        return out

    g_direct, g_local = constant_one_on_boxes(gtrav, g.box_levels, g.box_parent_ids, g_own, g_cum, g.nlevels)
    g_near_only = csr_rows_sum(gtrav.neighbor_source_boxes_starts, gtrav.neighbor_source_boxes_lists, g_own)
    want_full = per_target(gtrav, g, g_direct, g_local, False)[g.sorted_target_ids]
    want_near = per_target(gtrav, g, g_near_only, None, True)[g.sorted_target_ids]
    assert np.all(want_full == w_all.sum())        # the completeness test, with weights

    # every rank evaluates on its LET (box sums by global number), potentials go home through
    # the library's route: collective again, so the ranks run as threads once more
    def eval_fn(rank):
        res = results[rank]
        let, trav, lg = res["let"], res["trav"], res["let_gid"]
        own_w, cum_w = own_glob[lg], cum_glob[lg]
        direct, local = constant_one_on_boxes(trav, let.box_levels, let.box_parent_ids, own_w, cum_w,
                                              let.nlevels)
        near = csr_rows_sum(trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists, own_w)
        # LET box -> the rank's local tree box (its own boxes), for the particle ranges
        h = res["tree"]
        loc_of = {int(gb): i for i, gb in enumerate(res["local_gid"])}
        tb = np.asarray(trav.target_boxes, np.int64)
        pots = []
        for val in (direct + local[tb], near):
            pot = np.full(h.ntargets, np.nan)
            for b, v in zip(tb, val):
                lb = loc_of[int(lg[b])]
                s, c = h.box_target_starts[lb], h.box_target_counts_nonchild[lb]
                pot[s:s + c] = v
            assert not np.isnan(pot).any(), "a target of this rank got no potential"
            # tree order -> received order -> the caller's order
            recv = torch.from_numpy(pot).cuda()[res["dev_tree"].sorted_target_ids.long()]
            pots.append(res["route"].to_callers(recv, "targets" if sep else "sources").cpu().numpy())
        res["comm"].close()
        return pots

    pots = run_ranks(world, eval_fn)
    group.close()
    got_full = np.concatenate([p[0] for p in pots])
    got_near = np.concatenate([p[1] for p in pots])
    assert np.array_equal(got_full, want_full)
    assert np.array_equal(got_near, want_near)
    if n_all >= 20000 * 2:
        assert len(np.unique(want_near)) > 10      # (it does tell the sources apart)


def test_route_errors():
    """Routes need an exchange on the context, the exchange's communicator and its sizes."""
    import torch
    from boxtree_amd import HIPArrayContext, _lib
    from boxtree_amd.distributed import native as nat
    actx = HIPArrayContext(0)
    group = nat.LocalGroup(1)
    comm = group.comm(0)
    a = torch.zeros(10, dtype=torch.int32, device="cuda")
    import ctypes as ct
    with pytest.raises(_lib.BoxtreeHipError, match="no exchange"):
        _lib.check(actx.lib.bt_mgpu_route(actx.handle, comm.handle, 0, 0, 4, ct.c_void_p(a.data_ptr()),
                                          ct.c_void_p(a.data_ptr())))
    # (a rank that leaves a collective with an error fails its local group for good: the peers
    # must not wait for it)
    comm.close()
    group.close()
    group = nat.LocalGroup(1)
    comm = group.comm(0)
    pts = [torch.rand(5000, dtype=torch.float64, device="cuda") for _ in range(3)]
    p2, kw, stats = nat.exchange_particles(actx, comm, pts, 30)
    route = stats["route"]
    with pytest.raises(ValueError, match="no separate targets"):
        route.to_owners(a, "targets")
    with pytest.raises(ValueError, match="values for"):
        route.to_owners(a, "sources")
    with pytest.raises(TypeError):
        route.to_owners(torch.zeros(5000, dtype=torch.int16, device="cuda"))
    ids = route.global_ids("sources")
    assert torch.equal(ids.sort().values, torch.arange(5000, dtype=torch.int32, device="cuda"))
    with pytest.raises(_lib.BoxtreeHipError, match="invalid argument"):
        _lib.check(actx.lib.bt_mgpu_route(actx.handle, comm.handle, 0, 7, 4, ct.c_void_p(a.data_ptr()),
                                          ct.c_void_p(a.data_ptr())))
    comm.close()
    group.close()
