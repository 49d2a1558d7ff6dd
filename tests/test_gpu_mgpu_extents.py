"""The sharded build with targets that have extents (BASELINE configs[3] on N ranks): steps 1-5
through the library's multi-GPU entries, the ranks being threads over a local communicator.

Where a particle with an extent stops is not a function of the cell histogram
(tree_build_kernels.py:388-428), so the exchange counts, per box of the shared top levels, the
targets that stay in it; a stayer travels to the owner of the box's first cell.  The per-rank
trees, numbered globally, must be slices of the tree one GPU builds from everything."""

import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_chunks(world, dims, n_src, n_tgt, seed, dist_kind, radius_scale):
    src, tgt, rad = [], [], []
    for r in range(world):
        rng = np.random.default_rng(seed + r)
        if dist_kind == "uniform":
            s = [rng.random(n_src) for _ in range(dims)]
            t = [rng.random(n_tgt) for _ in range(dims)]
        elif dist_kind == "normal":
            s = [rng.standard_normal(n_src) for _ in range(dims)]
            t = [rng.standard_normal(n_tgt) for _ in range(dims)]
        else:   # a dense blob and a thin background
            s = [np.where(rng.random(n_src) < 0.7, 0.6 + 2e-2 * rng.standard_normal(n_src), rng.random(n_src))
                 for _ in range(dims)]
            t = [rng.random(n_tgt) for _ in range(dims)]
        # radii over four decades: most targets go deep, some stay in the top boxes
        rr = radius_scale * 10.0 ** rng.uniform(-4, 0, n_tgt)
        src.append(s); tgt.append(t); rad.append(rr)
    return src, tgt, rad


def run_ranks(world, fn):
    results, errors = [None] * world, []

    def body(rank):
        try:
            results[rank] = fn(rank)
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors.append((rank, repr(e), traceback.format_exc()[-2000:]))

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads), "a rank is stuck in a collective"
    return results


@pytest.mark.parametrize("dims,world,dist_kind,norm", [(3, 2, "uniform", "linf"), (3, 3, "blob", "linf"),
                                                       (2, 4, "normal", "l2"), (3, 5, "uniform", "l2"),
                                                       (2, 2, "blob", "linf")])
def test_sharded_tree_with_target_extents(dims, world, dist_kind, norm):
    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    n_src, n_tgt, mpb, top_level, sof = 30000, 6000, 20, 3 if dims == 3 else 4, 0.25
    src, tgt, rad = make_chunks(world, dims, n_src, n_tgt, 400, dist_kind,
                                0.4 if dist_kind == "normal" else 0.05)
    group = nat.LocalGroup(world)

    def rank_fn(rank):
        actx = HIPArrayContext(0)
        comm = group.comm(rank)
        p = [torch.from_numpy(a).cuda() for a in src[rank]]
        t = [torch.from_numpy(a).cuda() for a in tgt[rank]]
        r = torch.from_numpy(rad[rank]).cuda()
        p2, t2, r2, kw, stats = nat.exchange_particles(actx, comm, p, mpb, top_level=top_level, targets=t,
                                                       target_radii=r, stick_out_factor=sof, extent_norm=norm)
        tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, target_radii=r2, max_particles_in_box=mpb, **kw)
        num = nat.number_sharded_tree(actx, comm, tree)
        comm.close()
        return dict(tree=actx.to_numpy(tree), gid=num["box_ids"].cpu().numpy().astype(np.int64),
                    num={k: num[k] for k in ("nboxes", "nlevels", "source_offset", "target_offset",
                                             "nsources", "ntargets")},
                    lsb=num["global_level_start_box_nrs"])

    results = run_ranks(world, rank_fn)
    group.close()

    actx = HIPArrayContext(0)
    cat = lambda chunks, ax: torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()  # noqa: E731
    allsrc = [cat(src, ax) for ax in range(dims)]
    alltgt = [cat(tgt, ax) for ax in range(dims)]
    allrad = torch.from_numpy(np.concatenate(rad)).cuda()
    gt, _ = TreeBuilder(actx)(actx, allsrc, targets=alltgt, target_radii=allrad, max_particles_in_box=mpb,
                              stick_out_factor=sof, extent_norm=norm)
    g = actx.to_numpy(gt)
    assert np.any((g.box_target_counts_nonchild > 0) & (g.box_levels <= top_level)
                  & (g.box_target_counts_nonchild < g.box_target_counts_cumul)), \
        "the case has no target that stays in an internal top box"

    hits = np.zeros(g.nboxes, np.int64)
    tcum = np.zeros(g.nboxes, np.int64)
    scum = np.zeros(g.nboxes, np.int64)
    tnon = np.zeros(g.nboxes, np.int64)
    for res in results:
        h, m, num = res["tree"], res["gid"], res["num"]
        nb = h.nboxes
        assert num["nboxes"] == g.nboxes and num["nlevels"] == g.nlevels
        assert np.array_equal(res["lsb"], g.level_start_box_nrs)
        assert num["nsources"] == g.nsources and num["ntargets"] == g.ntargets
        assert len(set(m.tolist())) == nb
        hits[m] += 1
        assert np.array_equal(g.box_levels[m], h.box_levels)
        assert np.array_equal(g.box_centers[:, m], h.box_centers[:, :nb])
        assert np.array_equal(g.box_parent_ids[m], m[h.box_parent_ids])
        ch = h.box_child_ids[:, :nb]
        mapped = np.where(ch != 0, m[ch], 0)
        gch = g.box_child_ids[:, m]
        deep = h.box_levels > top_level
        assert np.array_equal(mapped[:, deep], gch[:, deep])
        assert np.all((mapped == 0) | (mapped == gch))
        for name in ("box_source_counts_cumul", "box_source_counts_nonchild", "box_target_counts_cumul",
                     "box_target_counts_nonchild", "box_flags"):
            assert np.array_equal(getattr(g, name)[m][deep], getattr(h, name)[deep]), name
        so, to = num["source_offset"], num["target_offset"]
        assert np.array_equal(g.box_source_starts[m][deep], h.box_source_starts[deep] + so)
        assert np.array_equal(g.box_target_starts[m][deep], h.box_target_starts[deep] + to)
        scum[m] += h.box_source_counts_cumul
        tcum[m] += h.box_target_counts_cumul
        tnon[m] += h.box_target_counts_nonchild
        # the rank's particles are one slice of the global tree order
        for ax in range(dims):
            assert np.array_equal(g.sources[ax][so:so + h.nsources], h.sources[ax])
            assert np.array_equal(g.targets[ax][to:to + h.ntargets], h.targets[ax])
        assert np.array_equal(g.target_radii[to:to + h.ntargets], h.target_radii)
    assert np.all(hits >= 1) and np.all(hits[g.box_levels > top_level] == 1)
    assert np.array_equal(scum, g.box_source_counts_cumul)
    assert np.array_equal(tcum, g.box_target_counts_cumul)
    # the targets that stay in a shared top box are all on ONE rank
    top = g.box_levels <= top_level
    assert np.array_equal(tnon[top & (g.box_target_counts_nonchild < g.box_target_counts_cumul)],
                          g.box_target_counts_nonchild[top & (g.box_target_counts_nonchild
                                                              < g.box_target_counts_cumul)])


@pytest.mark.parametrize("dims,world,dist_kind,mode", [(3, 3, "uniform", "sources"), (2, 4, "blob", "sources"),
                                                       (3, 2, "blob", "targets"), (2, 5, "normal", "targets"),
                                                       (3, 4, "uniform", "extents"), (2, 3, "blob", "extents")])
def test_sharded_tree_with_refine_weights(dims, world, dist_kind, mode):
    """Refine weights (tree_build.py:395-446, tree_build_kernels.py:569-591: a box splits iff the
    weight bound for its children exceeds max_leaf_refine_weight): the exchange sums weights per
    cell next to the counts, the weights travel with the particles, and the per-rank trees are
    slices of the single-GPU tree built with the same weights -- sources only, with separate
    targets, and with targets that have extents."""
    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    n_src, n_tgt, max_w, top_level, sof = 30000, 6000, 40, 3 if dims == 3 else 4, 0.25
    src, tgt, rad = make_chunks(world, dims, n_src, n_tgt, 900, dist_kind,
                                0.4 if dist_kind == "normal" else 0.05)
    wsrc = [np.random.default_rng(70 + r).integers(0, 6, n_src).astype(np.int32) for r in range(world)]
    wtgt = [np.random.default_rng(170 + r).integers(0, 4, n_tgt).astype(np.int32) for r in range(world)]
    group = nat.LocalGroup(world)

    def rank_fn(rank):
        actx = HIPArrayContext(0)
        comm = group.comm(rank)
        p = [torch.from_numpy(a).cuda() for a in src[rank]]
        kw_x = dict(top_level=top_level, refine_weights=torch.from_numpy(wsrc[rank]).cuda(),
                    max_leaf_refine_weight=max_w)
        if mode == "sources":
            p2, kw, stats = nat.exchange_particles(actx, comm, p, None, **kw_x)
            tree, _ = TreeBuilder(actx)(actx, p2, **kw)
        else:
            t = [torch.from_numpy(a).cuda() for a in tgt[rank]]
            kw_x["targets"] = t
            # (one rank passes no target weights: ones)
            kw_x["target_refine_weights"] = None if rank == 1 else torch.from_numpy(wtgt[rank]).cuda()
            if mode == "extents":
                p2, t2, r2, kw, stats = nat.exchange_particles(
                    actx, comm, p, None, target_radii=torch.from_numpy(rad[rank]).cuda(),
                    stick_out_factor=sof, extent_norm="linf", **kw_x)
                tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, target_radii=r2, **kw)
            else:
                p2, t2, kw, stats = nat.exchange_particles(actx, comm, p, None, **kw_x)
                tree, _ = TreeBuilder(actx)(actx, p2, targets=t2, **kw)
        num = nat.number_sharded_tree(actx, comm, tree)
        comm.close()
        return dict(tree=actx.to_numpy(tree), gid=num["box_ids"].cpu().numpy().astype(np.int64),
                    num={k: num[k] for k in ("nboxes", "nlevels", "source_offset", "target_offset")})

    results = run_ranks(world, rank_fn)
    group.close()

    actx = HIPArrayContext(0)
    cat = lambda chunks, ax: torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()  # noqa: E731
    allsrc = [cat(src, ax) for ax in range(dims)]
    weights = np.concatenate(wsrc)
    gkw = {}
    if mode != "sources":
        gkw["targets"] = [cat(tgt, ax) for ax in range(dims)]
        wt = [np.ones(n_tgt, np.int32) if r == 1 else wtgt[r] for r in range(world)]
        weights = np.concatenate([weights] + wt)
        if mode == "extents":
            gkw.update(target_radii=torch.from_numpy(np.concatenate(rad)).cuda(), stick_out_factor=sof,
                       extent_norm="linf")
    gt, _ = TreeBuilder(actx)(actx, allsrc, refine_weights=torch.from_numpy(weights).cuda(),
                              max_leaf_refine_weight=max_w, **gkw)
    g = actx.to_numpy(gt)
    # the weights matter: the same particles with unit weights make another tree
    gu = TreeBuilder(actx)(actx, allsrc, max_particles_in_box=max_w, **gkw)[0]
    assert int(gu.nboxes) != g.nboxes
    hits = np.zeros(g.nboxes, np.int64)
    scum = np.zeros(g.nboxes, np.int64)
    for res in results:
        h, m, num = res["tree"], res["gid"], res["num"]
        nb = h.nboxes
        assert num["nboxes"] == g.nboxes and num["nlevels"] == g.nlevels
        assert len(set(m.tolist())) == nb
        hits[m] += 1
        assert np.array_equal(g.box_levels[m], h.box_levels)
        assert np.array_equal(g.box_centers[:, m], h.box_centers[:, :nb])
        assert np.array_equal(g.box_parent_ids[m], m[h.box_parent_ids])
        ch = h.box_child_ids[:, :nb]
        mapped = np.where(ch != 0, m[ch], 0)
        gch = g.box_child_ids[:, m]
        deep = h.box_levels > top_level
        assert np.array_equal(mapped[:, deep], gch[:, deep])
        assert np.all((mapped == 0) | (mapped == gch))
        for name in ("box_source_counts_cumul", "box_source_counts_nonchild", "box_flags"):
            assert np.array_equal(getattr(g, name)[m][deep], getattr(h, name)[deep]), name
        so = num["source_offset"]
        assert np.array_equal(g.box_source_starts[m][deep], h.box_source_starts[deep] + so)
        scum[m] += h.box_source_counts_cumul
        for ax in range(dims):
            assert np.array_equal(g.sources[ax][so:so + h.nsources], h.sources[ax])
        if mode != "sources":
            to = num["target_offset"]
            for ax in range(dims):
                assert np.array_equal(g.targets[ax][to:to + h.ntargets], h.targets[ax])
    assert np.all(hits >= 1) and np.all(hits[g.box_levels > top_level] == 1)
    assert np.array_equal(scum, g.box_source_counts_cumul)


def test_one_call_entry_with_extents_and_weights():
    """boxtree_amd.distributed.native.sharded_tree_and_lists with TreeBuilder's keyword arguments:
    targets with radii, and refine weights on both sets, three thread ranks."""
    import torch
    from boxtree_amd import HIPArrayContext, TreeBuilder
    from boxtree_amd.distributed import native as nat
    world, dims, n_src, n_tgt, max_w, sof = 3, 3, 20000, 4000, 30, 0.25
    src, tgt, rad = make_chunks(world, dims, n_src, n_tgt, 1500, "uniform", 0.05)
    wsrc = [np.random.default_rng(7 + r).integers(1, 4, n_src).astype(np.int32) for r in range(world)]
    wtgt = [np.random.default_rng(17 + r).integers(1, 3, n_tgt).astype(np.int32) for r in range(world)]
    group = nat.LocalGroup(world)

    def rank_fn(rank):
        actx = HIPArrayContext(0)
        comm = group.comm(rank)
        dev = lambda a: torch.from_numpy(a).cuda()      # noqa: E731
        out = nat.sharded_tree_and_lists(
            actx, comm, [dev(a) for a in src[rank]], targets=[dev(a) for a in tgt[rank]],
            target_radii=dev(rad[rank]), stick_out_factor=sof, refine_weights=dev(wsrc[rank]),
            target_refine_weights=dev(wtgt[rank]), max_leaf_refine_weight=max_w)
        comm.close()
        return dict(nboxes=int(out["numbering"]["nboxes"]), nlevels=int(out["numbering"]["nlevels"]),
                    nl1=int(out["traversal"].neighbor_source_boxes_lists.shape[0]))

    results = run_ranks(world, rank_fn)
    group.close()
    actx = HIPArrayContext(0)
    cat = lambda chunks, ax: torch.from_numpy(np.concatenate([c[ax] for c in chunks])).cuda()  # noqa: E731
    g, _ = TreeBuilder(actx)(
        actx, [cat(src, ax) for ax in range(dims)], targets=[cat(tgt, ax) for ax in range(dims)],
        target_radii=torch.from_numpy(np.concatenate(rad)).cuda(), stick_out_factor=sof,
        refine_weights=torch.from_numpy(np.concatenate(wsrc + wtgt)).cuda(), max_leaf_refine_weight=max_w)
    for r in results:
        assert r["nboxes"] == int(g.nboxes) and r["nlevels"] == int(g.nlevels) and r["nl1"] > 0
