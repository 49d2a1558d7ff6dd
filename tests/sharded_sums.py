"""Sums of a traversal's lists that a SHARDED build can reproduce rank by rank (TEST
INFRASTRUCTURE, on tests/fullsize_sums.py): every list row belongs to a box of the GLOBAL tree, a
row's value is order-sensitive, and the sum over rows is linear -- so the rows the ranks build
(local box numbers mapped to global ones) add up to the single tree's sum.  Rows that several
ranks build (boxes of the shared top levels) must carry the same value; a row nobody builds
counts as empty.

    single tree (oracle or one GPU):  single_tree_sums(torch, tree, trav)
    a rank:                           rank_rows(torch, trav, gid, mask) -> RowMerger.add(...)
    all ranks merged:                 RowMerger.sums()
"""

import fullsize_sums as fs


def list_names(nlevels):
    return ["colleagues", "list1", "list2", "list4"] + [f"list3[{lev}]" for lev in range(nlevels)]


def single_tree_sums(torch, tree, trav):
    """{list name: sum} + entry counts of one tree's traversal (box numbers are global already)."""
    out = {
        "colleagues": fs.csr_rows_sum(torch, trav.same_level_non_well_sep_boxes_starts,
                                      trav.same_level_non_well_sep_boxes_lists),
        "list1": fs.csr_rows_sum(torch, trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists,
                                 row_gid=trav.target_boxes),
        "list2": fs.csr_rows_sum(torch, trav.from_sep_siblings_starts, trav.from_sep_siblings_lists,
                                 row_gid=trav.target_or_target_parent_boxes),
        "list4": fs.csr_rows_sum(torch, trav.from_sep_bigger_starts, trav.from_sep_bigger_lists,
                                 row_gid=trav.target_or_target_parent_boxes),
    }
    for lev, bl in enumerate(trav.from_sep_smaller_by_level):
        out[f"list3[{lev}]"] = fs.csr_rows_sum(
            torch, bl.starts, bl.lists, row_gid=trav.target_boxes_sep_smaller_by_source_level[lev])
    return out


def rank_rows(torch, trav, gid, mask):
    """{list name: (global box of every row this rank built, the rows' values)} for the traversal
    of a rank's local essential tree; gid: LET box -> global box, mask: target_boxes_mask."""
    gid = gid.to(torch.int64)
    rows = {}
    act = torch.nonzero(mask != 0).reshape(-1)
    v = fs.csr_row_values(torch, trav.same_level_non_well_sep_boxes_starts,
                          trav.same_level_non_well_sep_boxes_lists, entry_gid=gid)
    rows["colleagues"] = (gid[act], v[act])
    rows["list1"] = (gid[trav.target_boxes.long()], fs.csr_row_values(
        torch, trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists, entry_gid=gid))
    ttp = gid[trav.target_or_target_parent_boxes.long()]
    rows["list2"] = (ttp, fs.csr_row_values(torch, trav.from_sep_siblings_starts,
                                            trav.from_sep_siblings_lists, entry_gid=gid))
    rows["list4"] = (ttp, fs.csr_row_values(torch, trav.from_sep_bigger_starts,
                                            trav.from_sep_bigger_lists, entry_gid=gid))
    for lev, bl in enumerate(trav.from_sep_smaller_by_level):
        tb = trav.target_boxes_sep_smaller_by_source_level[lev]
        rows[f"list3[{lev}]"] = (gid[tb.long()], fs.csr_row_values(torch, bl.starts, bl.lists, entry_gid=gid))
    return rows


class RowMerger:
    """Rows of all ranks by global box number; overlapping rows must agree."""

    def __init__(self, torch, nglobal, nlevels, device):
        self.torch = torch
        self.names = list_names(nlevels)
        self.acc = {k: torch.zeros(nglobal, dtype=torch.int64, device=device) for k in self.names}
        self.seen = {k: torch.zeros(nglobal, dtype=torch.bool, device=device) for k in self.names}
        self.disagreements = []
        self.nglobal = nglobal

    def add(self, rows):
        for name, (g, vals) in rows.items():
            g = g.to(self.acc[name].device).to(self.torch.int64)
            vals = vals.to(self.acc[name].device)
            old = self.seen[name][g]
            if bool((self.acc[name][g][old] != vals[old]).any()):
                self.disagreements.append(name)
            self.acc[name][g] = vals
            self.seen[name][g] = True

    def sums(self):
        allg = self.torch.arange(self.nglobal, device=self.acc[self.names[0]].device, dtype=self.torch.int64)
        return {k: fs.rows_sum(self.torch, allg, self.acc[k]) for k in self.names}
