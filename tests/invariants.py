"""Property checks restated from the reference's own test-suite.

These are the assertions that pin the hot path in inducer/boxtree's tests
(there are no golden vectors there):

* test/test_tree.py:88-220   (run_build_test)
* test/test_tree.py:341-437  (test_source_target_tree)
* test/test_tree.py:445-629  (test_extent_tree)
* test/test_traversal.py:58-267 (test_tree_connectivity)
* test/test_fmm.py:141-391 + boxtree/constant_one.py:50-237 + boxtree/fmm.py:342-532
  (interaction completeness with the constant-one "FMM")

They operate on any object with the reference's ``Tree`` / ``FMMTraversalInfo``
attribute names holding numpy arrays (oracle output or product output moved to
the host).
"""

import numpy as np

IS_SOURCE_BOX = 1
IS_TARGET_BOX = 2
HAS_SOURCE_CHILD_BOXES = 4
HAS_TARGET_CHILD_BOXES = 8


def _box_extents(tree):
    lev = tree.box_levels.astype(np.int64)
    box_size = tree.root_extent / (1 << lev)
    centers = tree.box_centers[:, :tree.nboxes]
    lo = centers - 0.5 * box_size
    hi = lo + box_size
    return lo, hi


def check_tree(tree, particles, targets=None, max_particles_in_box=None,
               refine_weights=None, max_leaf_refine_weight=None,
               source_radii=None, target_radii=None, extent_norm=None):
    """test_tree.py run_build_test / test_source_target_tree / test_extent_tree."""
    dtype = np.dtype(tree.coord_dtype)
    tol = 1e-4 if dtype == np.float32 else 1e-12
    scaled_tol = tol * tree.root_extent
    nb = tree.nboxes
    dims = tree.dimensions

    unsorted_sources = np.array(particles)
    sorted_sources = np.array(list(tree.sources))
    assert np.all(sorted_sources == unsorted_sources[:, tree.user_source_ids])
    assert sorted(tree.user_source_ids.tolist()) == list(range(tree.nsources))

    if targets is not None:
        unsorted_targets = np.array(targets)
        sorted_targets = np.array(list(tree.targets))
        user_target_ids = np.empty(tree.ntargets, dtype=np.intp)
        user_target_ids[tree.sorted_target_ids] = np.arange(tree.ntargets)
        assert np.all(sorted_targets == unsorted_targets[:, user_target_ids])
        if target_radii is not None:
            assert np.all(tree.target_radii == target_radii[user_target_ids])
        if source_radii is not None:
            assert np.all(tree.source_radii == source_radii[tree.user_source_ids])
    else:
        sorted_targets = sorted_sources
        # sorted_target_ids is the inverse permutation (tree_build.py:1467)
        assert np.all(tree.user_source_ids[tree.sorted_target_ids]
                      == np.arange(tree.nsources))

    lo, hi = _box_extents(tree)
    occupied = (tree.box_flags & (IS_SOURCE_BOX | IS_TARGET_BOX)) != 0
    if tree._is_pruned and not (tree.sources_have_extent or tree.targets_have_extent):
        # every pruned box contains at least one particle
        assert np.all(tree.box_source_counts_cumul + tree.box_target_counts_cumul > 0)

    bb_lo, bb_hi = tree.bounding_box
    assert np.all(lo[:, occupied] >= bb_lo[:, None] - scaled_tol)
    assert np.all(hi[:, occupied] <= bb_hi[:, None] + scaled_tol)

    centers = tree.box_centers[:, :nb]
    have_ext = tree.sources_have_extent or tree.targets_have_extent
    if not have_ext:
        for bmin, bmax in [
                (tree.box_source_bounding_box_min, tree.box_source_bounding_box_max),
                (tree.box_target_bounding_box_min, tree.box_target_bounding_box_max)]:
            bmin = bmin[:, :nb][:, occupied]
            bmax = bmax[:, :nb][:, occupied]
            assert np.all(lo[:, occupied] - scaled_tol <= bmin)
            assert np.all(bmin - scaled_tol <= centers[:, occupied])
            assert np.all(bmax - scaled_tol <= hi[:, occupied])
            assert np.all(centers[:, occupied] - scaled_tol <= bmax)

    # nonchild + sum(children cumul) == cumul  (test_tree.py:182-184, 404-409)
    child = tree.box_child_ids[:, :nb]
    for cumul, nonchild in [
            (tree.box_source_counts_cumul, tree.box_source_counts_nonchild),
            (tree.box_target_counts_cumul, tree.box_target_counts_nonchild)]:
        kid_sum = np.where(child != 0, cumul[child], 0).sum(axis=0)
        assert np.all(nonchild + kid_sum == cumul)

    if have_ext:
        assert np.sum(tree.box_source_counts_nonchild) == tree.nsources
        assert np.sum(tree.box_target_counts_nonchild) == tree.ntargets

    # all particles of a box lie in [low, high)   (test_tree.py:186-190, 411-420)
    for starts, counts, pts, radii, is_src in [
            (tree.box_source_starts, tree.box_source_counts_cumul, sorted_sources,
             tree.source_radii, True),
            (tree.box_target_starts, tree.box_target_counts_cumul, sorted_targets,
             tree.target_radii, False)]:
        npts = pts.shape[1]
        # per-level check: within a level the boxes' particle ranges are disjoint
        for lev in range(tree.nlevels):
            b0, b1 = tree.level_start_box_nrs[lev:lev + 2]
            ids = np.arange(b0, b1)
            cnt = counts[b0:b1]
            owner = np.repeat(ids, cnt)
            idx = np.concatenate([
                np.arange(s, s + c) for s, c in zip(starts[b0:b1], cnt)]
                or [np.zeros(0, np.int64)]).astype(np.int64)
            assert idx.size == 0 or idx.max() < npts
            p = pts[:, idx]
            if not have_ext:
                assert np.all(p < hi[:, owner] + scaled_tol)
                assert np.all(lo[:, owner] - scaled_tol <= p)
            else:
                # stick-out criterion, test_tree.py:580-629
                r = radii[idx] if radii is not None else 0.0
                box_radius = 0.5 * tree.root_extent / (
                    1 << tree.box_levels[owner].astype(np.int64))
                c = centers[:, owner]
                if (extent_norm or "linf") == "linf":
                    so = tree.stick_out_factor * box_radius
                    assert np.all(p + r < c + box_radius + so)
                    assert np.all(c - box_radius - so <= p - r)
                else:
                    rws = (1 + tree.stick_out_factor) * box_radius
                    cd = np.sqrt(np.sum((p - c) ** 2, axis=0))
                    assert np.all((cd + r) ** 2 < dims * rws ** 2)

    # leaf occupancy  (test_tree.py:203-218)
    leaf = (tree.box_flags & (HAS_SOURCE_CHILD_BOXES | HAS_TARGET_CHILD_BOXES)) == 0
    if max_particles_in_box is not None and not have_ext:
        tot = tree.box_source_counts_cumul.astype(np.int64)
        if not tree.sources_are_targets:
            tot = tot + tree.box_target_counts_cumul
        assert np.all(tot[leaf] <= max_particles_in_box)
    if refine_weights is not None and tree.sources_are_targets:
        w = refine_weights[tree.user_source_ids].astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(w)])
        s = tree.box_source_starts
        bw = cs[s + tree.box_source_counts_cumul] - cs[s]
        assert np.all(bw[leaf] <= max_leaf_refine_weight)

    # structural consistency (test_traversal.py:85-90)
    par = tree.box_parent_ids
    lv = tree.box_levels.astype(np.int64)
    ids = np.arange(1, nb)
    assert np.all(lv[par[ids]] + 1 == lv[ids])
    assert np.all((child[:, par[ids]] == ids[None, :]).sum(axis=0) == 1)
    assert np.all(np.diff(lv) >= 0)
    ls = tree.level_start_box_nrs
    assert ls[0] == 0 and ls[-1] == nb
    for lev in range(tree.nlevels):
        assert np.all(lv[ls[lev]:ls[lev + 1]] == lev)
    assert tree.aligned_nboxes == (nb + 31) // 32 * 32
    assert np.all(tree.box_child_ids[:, nb:] == 0)


def _csr(starts, lists, i):
    return lists[starts[i]:starts[i + 1]]


def check_traversal(tree, trav):
    """test_traversal.py:58-267 (test_tree_connectivity)."""
    nb = tree.nboxes
    levels = tree.box_levels.astype(np.int64)
    children = tree.box_child_ids[:, :nb]
    centers = tree.box_centers[:, :nb]
    sat = tree.sources_are_targets
    have_ext = tree.sources_have_extent or tree.targets_have_extent

    # list 1 consists of source boxes (leaves, when there are no extents)
    st = trav.neighbor_source_boxes_starts
    ls = trav.neighbor_source_boxes_lists
    assert len(st) == len(trav.target_boxes) + 1
    if not have_ext:
        assert np.all(children[:, ls] == 0)
    assert np.all(tree.box_flags[ls] & IS_SOURCE_BOX)
    if sat:
        owner = np.repeat(np.arange(len(trav.target_boxes)), np.diff(st))
        has_self = np.zeros(len(trav.target_boxes), bool)
        has_self[owner[ls == trav.target_boxes[owner]]] = True
        assert np.all(has_self)

    # list 2: same level and separated
    st = trav.from_sep_siblings_starts
    ls = trav.from_sep_siblings_lists
    tb = trav.target_or_target_parent_boxes
    assert len(st) == len(tb) + 1
    owner = tb[np.repeat(np.arange(len(tb)), np.diff(st))]
    assert np.all(levels[ls] == levels[owner])
    mindist = 2.5 * 0.5 * 2.0 ** -levels[owner] * tree.root_extent
    dist = np.sqrt(np.sum((centers[:, ls] - centers[:, owner]) ** 2, axis=0))
    assert np.all(dist > mindist)

    # list 3 / list 4 level assumptions + duality
    l3pairs = set()
    for level, ssn in enumerate(trav.from_sep_smaller_by_level):
        tboxes = trav.target_boxes_sep_smaller_by_source_level[level]
        assert len(tboxes) == ssn.num_nonempty_lists
        assert len(ssn.starts) == len(tboxes) + 1
        assert np.all(np.diff(ssn.starts) > 0)
        owner = tboxes[np.repeat(np.arange(len(tboxes)), np.diff(ssn.starts))]
        assert np.all(levels[owner] < levels[ssn.lists])
        assert np.all(levels[ssn.lists] == level)
        if sat and not have_ext:
            l3pairs.update(zip(owner.tolist(), ssn.lists.tolist()))

    st = trav.from_sep_bigger_starts
    ls = trav.from_sep_bigger_lists
    owner = tb[np.repeat(np.arange(len(tb)), np.diff(st))]
    assert np.all(levels[owner] > levels[ls])
    if sat and not have_ext:
        assert np.all(trav.target_or_target_parent_boxes == np.arange(nb))
        assert np.all(trav.source_boxes == trav.target_boxes)
        # test_traversal.py:147-214: lists 3 and 4 are duals of each other
        l4pairs = set(zip(owner.tolist(), ls.tolist()))   # (target, bigger source)
        assert l4pairs == {(s, t) for (t, s) in l3pairs}

    # level starts
    for name, ref in [
            ("level_start_source_box_nrs", trav.source_boxes),
            ("level_start_source_parent_box_nrs", trav.source_parent_boxes),
            ("level_start_target_box_nrs", trav.target_boxes),
            ("level_start_target_or_target_parent_box_nrs",
             trav.target_or_target_parent_boxes)]:
        lst = getattr(trav, name)
        assert len(lst) == tree.nlevels + 1
        for lev in range(tree.nlevels):
            assert np.all(levels[ref[lst[lev]:lst[lev + 1]]] == lev), name
        assert lst[-1] == len(ref)


def constant_one_potentials(tree, trav, filtered_user=None, filtered_tree=None):
    """Restatement of drive_fmm (boxtree/fmm.py:342-532) with the constant-one
    wrangler (boxtree/constant_one.py:50-237) and unit source weights.  Every
    target must end up with potential == nsources (test_fmm.py:141-391).
    """
    nb = tree.nboxes
    src_cnt = tree.box_source_counts_nonchild.astype(np.int64)

    def seg_sum(starts, lists, values):
        starts = np.asarray(starts, np.int64)
        out = np.zeros(len(starts) - 1, np.int64)
        if len(lists):
            owner = np.repeat(np.arange(len(starts) - 1), np.diff(starts))
            np.add.at(out, owner, values[lists])
        return out

    # form_multipoles + coarsen_multipoles
    mpoles = np.zeros(nb, np.int64)
    mpoles[trav.source_boxes] += src_cnt[trav.source_boxes]
    lsp = trav.level_start_source_parent_box_nrs
    for source_level in range(tree.nlevels - 1, 2, -1):
        target_level = source_level - 1
        start, stop = lsp[target_level:target_level + 2]
        for ibox in trav.source_parent_boxes[start:stop]:
            ch = tree.box_child_ids[:, ibox]
            mpoles[ibox] += mpoles[ch[ch != 0]].sum()

    ntb = len(trav.target_boxes)
    tgt_start = tree.box_target_starts
    tgt_cnt = tree.box_target_counts_nonchild
    pot_box = np.zeros(ntb, np.int64)     # per target box (all its targets equal)

    # eval_direct (list 1) and close lists
    pot_box += seg_sum(trav.neighbor_source_boxes_starts,
                       trav.neighbor_source_boxes_lists, src_cnt)
    if trav.from_sep_close_smaller_starts is not None:
        pot_box += seg_sum(trav.from_sep_close_smaller_starts,
                           trav.from_sep_close_smaller_lists, src_cnt)
    if trav.from_sep_close_bigger_starts is not None:
        pot_box += seg_sum(trav.from_sep_close_bigger_starts,
                           trav.from_sep_close_bigger_lists, src_cnt)

    # multipole_to_local (list 2)
    ttp = trav.target_or_target_parent_boxes
    local = np.zeros(nb, np.int64)
    local[ttp] += seg_sum(trav.from_sep_siblings_starts,
                          trav.from_sep_siblings_lists, mpoles)

    # eval_multipoles (list 3)
    tb_index = np.full(nb, -1, np.int64)
    tb_index[trav.target_boxes] = np.arange(ntb)
    for level, ssn in enumerate(trav.from_sep_smaller_by_level):
        tboxes = trav.target_boxes_sep_smaller_by_source_level[level]
        contrib = seg_sum(ssn.starts, ssn.lists, mpoles)
        pot_box[tb_index[tboxes]] += contrib

    # form_locals (list 4)
    local[ttp] += seg_sum(trav.from_sep_bigger_starts,
                          trav.from_sep_bigger_lists, src_cnt)

    # refine_locals
    ltt = trav.level_start_target_or_target_parent_box_nrs
    for target_lev in range(1, tree.nlevels):
        start, stop = ltt[target_lev:target_lev + 2]
        boxes = ttp[start:stop]
        local[boxes] += local[tree.box_parent_ids[boxes]]

    # eval_locals
    pot_box += local[trav.target_boxes]

    if filtered_user is not None:
        # test_fmm.py:128-138: the lists hold user target numbers; potentials come
        # back in user order (constant_one.py:78)
        pot = np.zeros(tree.ntargets, np.int64)
        for itb, ibox in enumerate(trav.target_boxes):
            ids = filtered_user.target_lists[
                filtered_user.target_starts[ibox]:filtered_user.target_starts[ibox + 1]]
            pot[tree.sorted_target_ids[ids]] += pot_box[itb]
        return pot[tree.sorted_target_ids]
    if filtered_tree is not None:
        # test_fmm.py:103-125
        potf = np.zeros(filtered_tree.nfiltered_targets, np.int64)
        for itb, ibox in enumerate(trav.target_boxes):
            s = filtered_tree.box_target_starts[ibox]
            potf[s:s + filtered_tree.box_target_counts_nonchild[ibox]] += pot_box[itb]
        allpot = np.zeros(tree.ntargets, np.int64)
        allpot[filtered_tree.unfiltered_from_filtered_target_indices] = potf
        return allpot[tree.sorted_target_ids]

    pot = np.zeros(tree.ntargets, np.int64)
    filled = np.zeros(tree.ntargets, bool)
    for itb, ibox in enumerate(trav.target_boxes):
        s = tgt_start[ibox]
        e = s + tgt_cnt[ibox]
        pot[s:e] += pot_box[itb]
        filled[s:e] = True
    assert filled.all()
    return pot


def constant_one_stages(tree, trav, weights):
    """drive_fmm (boxtree/fmm.py:342-532) with the constant-one wrangler
    (boxtree/constant_one.py:50-237), *weights* in user source order, returning what
    every stage returns, keyed like tests/golden/make_reference_vectors.py keys the
    reference's own run: ``<stage>_<call number>``, plus ``weights``/``potentials``."""
    nb = tree.nboxes
    w = np.asarray(weights, np.float64)[tree.user_source_ids]
    out = {"weights": np.asarray(weights, np.float64)}
    starts, cnt = tree.box_source_starts, tree.box_source_counts_nonchild
    box_w = np.array([w[starts[b]:starts[b] + cnt[b]].sum() for b in range(nb)])

    def rows(st, li, values):
        return np.array([values[li[st[i]:st[i + 1]]].sum() for i in range(len(st) - 1)])

    def to_targets(boxes, per_box):
        pot = np.zeros(tree.ntargets)
        for v, b in zip(per_box, boxes):
            s = tree.box_target_starts[b]
            pot[s:s + tree.box_target_counts_nonchild[b]] += v
        return pot

    mpoles = np.zeros(nb)
    mpoles[trav.source_boxes] += box_w[trav.source_boxes]
    out["form_multipoles_0"] = mpoles.copy()
    lsp = trav.level_start_source_parent_box_nrs
    for source_level in range(tree.nlevels - 1, 2, -1):
        start, stop = lsp[source_level - 1:source_level + 1]
        for ibox in trav.source_parent_boxes[start:stop]:
            ch = tree.box_child_ids[:, ibox]
            mpoles[ibox] += mpoles[ch[ch != 0]].sum()
    out["coarsen_multipoles_0"] = mpoles.copy()

    ndirect = 0

    def direct(st, li):
        nonlocal ndirect
        out[f"eval_direct_{ndirect}"] = to_targets(trav.target_boxes, rows(st, li, box_w))
        ndirect += 1
        return out[f"eval_direct_{ndirect - 1}"]

    pot = direct(trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists)
    ttp = trav.target_or_target_parent_boxes
    local = np.zeros(nb)
    local[ttp] += rows(trav.from_sep_siblings_starts, trav.from_sep_siblings_lists, mpoles)
    out["multipole_to_local_0"] = local.copy()
    m2p = np.zeros(tree.ntargets)
    for level, ssn in enumerate(trav.from_sep_smaller_by_level):
        m2p += to_targets(trav.target_boxes_sep_smaller_by_source_level[level],
                          rows(ssn.starts, ssn.lists, mpoles))
    out["eval_multipoles_0"] = m2p
    pot = pot + m2p
    if trav.from_sep_close_smaller_starts is not None:
        pot = pot + direct(trav.from_sep_close_smaller_starts, trav.from_sep_close_smaller_lists)
    p2l = np.zeros(nb)
    p2l[ttp] += rows(trav.from_sep_bigger_starts, trav.from_sep_bigger_lists, box_w)
    out["form_locals_0"] = p2l
    local = local + p2l
    if trav.from_sep_close_bigger_starts is not None:
        pot = pot + direct(trav.from_sep_close_bigger_starts, trav.from_sep_close_bigger_lists)
    ltt = trav.level_start_target_or_target_parent_box_nrs
    for target_lev in range(1, tree.nlevels):
        start, stop = ltt[target_lev:target_lev + 2]
        boxes = ttp[start:stop]
        local[boxes] += local[tree.box_parent_ids[boxes]]
    out["refine_locals_0"] = local.copy()
    out["eval_locals_0"] = to_targets(trav.target_boxes, local[trav.target_boxes])
    pot = pot + out["eval_locals_0"]
    out["potentials"] = pot[tree.sorted_target_ids]
    return out
