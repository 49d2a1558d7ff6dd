"""What ``debug=True`` runs (boxtree/tree_build.py:1039-1052, 1087-1098, 1545-1559;
boxtree/traversal.py:2035-2039): the reference's host assertions, restated on the finished
containers as device operations (no array comes to the host, so they also run at 10^9
particles).  The reference asserts while it builds, level by level; a sort-first build has no
such intermediate states, so the same facts are checked on the result:

* every box of a level's chunk carries that level (``:1087-1095``), the chunks are
  ``level_start_box_nrs``;
* the number of leaves of a level is its boxes minus those with children (``:1039-1052``) --
  here: parent and child tables agree, every box but the root is the child of its parent;
* particle starts lie inside the particle arrays (``:1097-1098``);
* ``user_source_ids`` / ``sorted_target_ids`` are in range (``:1545-1553``) -- here also:
  permutations;
* source and target counts add up (``:1555-1559``) -- here: ``cumul = nonchild + sum of the
  children's cumul`` on both sides, the root holds everything.

``tests/device_invariants.py`` adds what the reference's *tests* assert on top (coordinates,
leaf occupancy, containment)."""

from __future__ import annotations


def check_tree(torch, tree, chunk=1 << 27):
    """Raises ``AssertionError`` with the failed fact; returns a dict of what was checked."""
    nb = int(tree.nboxes)
    dev = tree.box_centers.device
    i64 = torch.int64
    out = {"nboxes": nb, "nsources": int(tree.nsources), "ntargets": int(tree.ntargets)}

    levels = tree.box_levels.to(i64)
    lsb = torch.as_tensor(tree.level_start_box_nrs).to(dev).to(i64)
    nlev = int(tree.nlevels)
    # level-major numbering: the chunk of level l holds boxes of level l and nothing else
    assert int(lsb[0]) == 0 and int(lsb[nlev]) == nb, "level_start_box_nrs does not span the boxes"
    assert bool((levels[1:] >= levels[:-1]).all()), "box levels do not ascend with the box number"
    assert torch.equal(torch.bincount(levels, minlength=nlev)[:nlev], lsb[1:nlev + 1] - lsb[:nlev]), \
        "a level's chunk holds boxes of another level"

    child = tree.box_child_ids[:, :nb].to(i64)              # [C, nb]
    parent = tree.box_parent_ids.to(i64)
    has = child != 0
    box = torch.arange(nb, device=dev, dtype=i64)
    for m in range(child.shape[0]):
        sel = has[m]
        c = child[m][sel]
        assert bool(((c > 0) & (c < nb)).all()), "child id out of range"
        assert torch.equal(parent[c], box[sel]), "a child does not point back at its parent"
        assert torch.equal(levels[c], levels[sel] + 1), "a child is not one level below its parent"
    nchildren = has.sum(0)
    assert int(nchildren.sum()) == nb - 1, "every box but the root is the child of exactly one box"
    assert int(parent[0]) == 0
    out["nleaves"] = int((nchildren == 0).sum())

    def side(name, n, ids, inverse):
        cumul = getattr(tree, f"box_{name}_counts_cumul").to(i64)
        nonchild = getattr(tree, f"box_{name}_counts_nonchild").to(i64)
        starts = getattr(tree, f"box_{name}_starts").to(i64)
        assert bool((nonchild >= 0).all()) and bool((nonchild <= cumul).all())
        kid_sum = torch.zeros(nb, dtype=i64, device=dev)
        for m in range(child.shape[0]):
            kid_sum += torch.where(has[m], cumul[child[m]], torch.zeros((), dtype=i64, device=dev))
        assert torch.equal(nonchild + kid_sum, cumul), f"{name}: nonchild + children's cumul != cumul"
        assert int(cumul[0]) == n, f"the root does not hold all {name}s"
        # (a box without particles of this side may start at n)
        assert bool(((starts >= 0) & (starts + cumul <= n)).all()), f"{name} starts outside the array"
        if n == 0:
            return
        lo_, hi_ = int(ids.min()), int(ids.max())
        assert lo_ == 0 and hi_ == n - 1, f"{name} ids outside [0, {n}): {lo_} .. {hi_}"
        hits = torch.zeros(n, dtype=torch.int8, device=dev)
        hits.index_fill_(0, ids.to(i64), 1)
        assert bool(hits.all()), f"the {name} ids are not a permutation"
        del hits
        if inverse is not None:
            for lo in range(0, n, chunk):
                hi = min(n, lo + chunk)
                assert torch.equal(ids[inverse[lo:hi].to(i64)].to(i64),
                                   torch.arange(lo, hi, device=dev, dtype=i64)), \
                    "sorted_target_ids is not the inverse of user_source_ids"

    same = bool(getattr(tree, "sources_are_targets", False))
    side("source", int(tree.nsources), tree.user_source_ids, tree.sorted_target_ids if same else None)
    if not same:
        side("target", int(tree.ntargets), tree.sorted_target_ids, None)
    if getattr(tree, "_is_pruned", True):
        total = tree.box_source_counts_cumul.to(i64)
        if not same:
            total = total + tree.box_target_counts_cumul.to(i64)
        assert bool((total > 0).all()), "a pruned tree has an empty box"
    out["checked"] = "levels, parent/child tables, counts, particle ids"
    return out


def check_traversal(torch, trav, nboxes):
    """Every list of an ``FMMTraversalInfo`` is a well-formed CSR over box numbers: starts begin
    at 0 and ascend, the list ends where they say, entries are boxes."""
    i64 = torch.int64
    nb = int(nboxes)

    def csr(name, starts, lists, nrows=None):
        if starts is None:
            return
        s = starts.to(i64)
        assert int(s[0]) == 0 and bool((s[1:] >= s[:-1]).all()), f"{name}: starts do not ascend from 0"
        assert int(s[-1]) == int(lists.shape[0]), f"{name}: {int(lists.shape[0])} entries, starts end at {int(s[-1])}"
        if nrows is not None:
            assert int(s.shape[0]) == nrows + 1, f"{name}: {int(s.shape[0]) - 1} rows for {nrows} boxes"
        if int(lists.shape[0]):
            assert int(lists.min()) >= 0 and int(lists.max()) < nb, f"{name}: entry is not a box"

    def boxes(name, a):
        if int(a.shape[0]):
            assert int(a.min()) >= 0 and int(a.max()) < nb, f"{name}: entry is not a box"
            assert bool((a[1:] > a[:-1]).all()), f"{name}: not ascending"

    for name in ("source_boxes", "target_boxes", "source_parent_boxes", "target_or_target_parent_boxes"):
        boxes(name, getattr(trav, name))
    ntb, nttp = int(trav.target_boxes.shape[0]), int(trav.target_or_target_parent_boxes.shape[0])
    csr("same_level_non_well_sep_boxes", trav.same_level_non_well_sep_boxes_starts,
        trav.same_level_non_well_sep_boxes_lists, nb)
    csr("neighbor_source_boxes", trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists, ntb)
    csr("from_sep_siblings", trav.from_sep_siblings_starts, trav.from_sep_siblings_lists, nttp)
    csr("from_sep_bigger", trav.from_sep_bigger_starts, trav.from_sep_bigger_lists, nttp)
    csr("from_sep_close_smaller", trav.from_sep_close_smaller_starts, trav.from_sep_close_smaller_lists, ntb)
    csr("from_sep_close_bigger", trav.from_sep_close_bigger_starts, trav.from_sep_close_bigger_lists, ntb)
    for lev, (lst, tb) in enumerate(zip(trav.from_sep_smaller_by_level,
                                        trav.target_boxes_sep_smaller_by_source_level)):
        csr(f"from_sep_smaller_by_level[{lev}]", lst.starts, lst.lists, int(tb.shape[0]))
        boxes(f"target_boxes_sep_smaller_by_source_level[{lev}]", tb)
    return {"checked": "box lists ascend, CSR lists well-formed"}
