"""Checksums of a whole Tree / FMMTraversalInfo, for comparing the HIP path with the CPU oracle at
BASELINE.json's FULL sizes, where shipping the oracle's arrays to the GPU box is not an option
(tests/golden/fullsize_oracle_sums.json holds a few hundred integers instead of ~10 GB).

TEST INFRASTRUCTURE.  One implementation, on torch tensors, for both sides: the oracle's numpy
arrays enter through torch.from_numpy (CPU), the product's arrays are device tensors.

* ``array_sum(a)`` = sum_p (p + 1) * bits(a.flat[p]) in wrapping int64 arithmetic: sensitive to the
  position of every element; floats enter by their bit pattern (f64 -> int64, f32 -> int32), so
  "equal sums" claims the same bit-for-bit equality tests/compare.py asserts array by array.
* ``csr_rows_sum`` = sum over rows r of w(g(r)) * sum_k (k + 1) * (g(entry_k) + 1): sensitive to the
  order within a row, LINEAR over rows -- the rows (target boxes) of a sharded traversal are built by
  exactly one rank each, so the ranks' sums over their own rows, with local box numbers mapped to
  global ones, add up to the single tree's value.

Semantics of the arrays: /root/reference/boxtree/tree.py:298-686, traversal.py:1353-1705, 2299-2343.
"""

from __future__ import annotations

import numpy as np

_CHUNK = 1 << 26
_MULT = 2654435761


def _t(torch, a):
    if isinstance(a, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(a))
    return a


def _bits(torch, a):
    if a.dtype == torch.float64:
        return a.view(torch.int64)
    if a.dtype == torch.float32:
        return a.view(torch.int32).to(torch.int64)
    if a.dtype == torch.bool:
        return a.to(torch.int64)
    return a.to(torch.int64)


def wrap(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def array_sum(torch, a):
    """sum_p (p + 1) * bits(a.flat[p]), wrapping int64; a: numpy array or torch tensor."""
    a = _t(torch, a).contiguous().reshape(-1)
    total = 0
    for lo in range(0, a.numel(), _CHUNK):
        part = _bits(torch, a[lo:lo + _CHUNK])
        pos = torch.arange(lo + 1, lo + 1 + part.numel(), device=part.device, dtype=torch.int64)
        total += int((pos * part).sum().item())
    return wrap(total)


def row_weights(torch, g):
    g = g.to(torch.int64)
    return ((g * _MULT) & 0xFFFFFFFF) | 1


def csr_row_values(torch, starts, lists, entry_gid=None):
    """int64[nrows]: value(r) = sum_k (k + 1) * (g(lists[starts[r] + k]) + 1), wrapping.  entry_gid:
    map from a list entry to its global id (default identity)."""
    starts = _t(torch, starts).to(torch.int64)
    lists = _t(torch, lists)
    nrows = starts.numel() - 1
    dev = starts.device
    if entry_gid is not None:
        entry_gid = _t(torch, entry_gid).to(dev).to(torch.int64)
    out = torch.zeros(max(nrows, 0), dtype=torch.int64, device=dev)
    r0 = 0
    while r0 < nrows:            # chunks of rows whose entries number about _CHUNK
        target = int(starts[r0].item()) + _CHUNK
        r1 = int(torch.searchsorted(starts, torch.tensor([target], device=dev), right=True).item()) - 1
        r1 = min(max(r1, r0 + 1), nrows)
        e0, e1 = int(starts[r0].item()), int(starts[r1].item())
        if e1 > e0:
            cnt = starts[r0 + 1:r1 + 1] - starts[r0:r1]
            rows = torch.repeat_interleave(torch.arange(r0, r1, device=dev), cnt)
            k = torch.arange(e0, e1, device=dev, dtype=torch.int64) - starts[rows]
            ent = lists[e0:e1].to(torch.int64)
            if entry_gid is not None:
                ent = entry_gid[ent]
            out.index_add_(0, rows, (k + 1) * (ent + 1))
        r0 = r1
    return out


def rows_sum(torch, row_gid, values):
    """sum_r w(g(r)) * value(r), wrapping int64: LINEAR over rows -- partial sums over disjoint row
    sets (the ranks of a sharded traversal) add up."""
    return wrap(int((row_weights(torch, _t(torch, row_gid).to(values.device)) * values).sum().item()))


def csr_rows_sum(torch, starts, lists, row_gid=None, entry_gid=None):
    """rows_sum of csr_row_values; row_gid: global id of the object row r belongs to (default r)."""
    vals = csr_row_values(torch, starts, lists, entry_gid)
    if row_gid is None:
        row_gid = torch.arange(vals.numel(), device=vals.device, dtype=torch.int64)
    return rows_sum(torch, row_gid, vals)


TREE_ARRAYS = [
    "level_start_box_nrs", "user_source_ids", "sorted_target_ids",
    "box_source_starts", "box_source_counts_nonchild", "box_source_counts_cumul",
    "box_target_starts", "box_target_counts_nonchild", "box_target_counts_cumul",
    "box_parent_ids", "box_child_ids", "box_centers", "box_levels", "box_flags",
    "box_source_bounding_box_min", "box_source_bounding_box_max",
    "box_target_bounding_box_min", "box_target_bounding_box_max",
]
TRAV_ARRAYS = [
    "source_boxes", "target_boxes", "source_parent_boxes", "target_or_target_parent_boxes",
    "level_start_source_box_nrs", "level_start_target_box_nrs",
    "level_start_source_parent_box_nrs", "level_start_target_or_target_parent_box_nrs",
    "same_level_non_well_sep_boxes_starts", "same_level_non_well_sep_boxes_lists",
    "neighbor_source_boxes_starts", "neighbor_source_boxes_lists",
    "from_sep_siblings_starts", "from_sep_siblings_lists",
    "from_sep_bigger_starts", "from_sep_bigger_lists",
    "from_sep_close_smaller_starts", "from_sep_close_smaller_lists",
    "from_sep_close_bigger_starts", "from_sep_close_bigger_lists",
]


def tree_sums(torch, tree):
    """{field: checksum or scalar} of every array tests/compare.py::assert_same_tree compares."""
    out = {
        "nboxes": int(tree.nboxes), "nlevels": int(tree.nlevels),
        "aligned_nboxes": int(tree.aligned_nboxes),
        "nsources": int(tree.nsources), "ntargets": int(tree.ntargets),
        "root_extent_hex": float(tree.root_extent).hex(),
        "bounding_box_min_hex": [float(v).hex() for v in np.asarray(tree.bounding_box[0])],
    }
    for name in TREE_ARRAYS:
        a = getattr(tree, name)
        out[name] = array_sum(torch, a)
        out[name + ".len"] = int(np.prod(a.shape))
    for d in range(int(tree.dimensions)):
        out[f"sources[{d}]"] = array_sum(torch, tree.sources[d])
        out[f"targets[{d}]"] = array_sum(torch, tree.targets[d])
    for name in ("source_radii", "target_radii"):
        a = getattr(tree, name, None)
        out[name] = None if a is None else array_sum(torch, a)
    return out


def traversal_sums(torch, trav):
    """{field: checksum} of every array tests/compare.py::assert_same_traversal compares."""
    out = {}
    for name in TRAV_ARRAYS:
        a = getattr(trav, name, None)
        out[name] = None if a is None else array_sum(torch, a)
        out[name + ".len"] = None if a is None else int(a.shape[0])
    for lev, bl in enumerate(trav.from_sep_smaller_by_level):
        pre = f"from_sep_smaller_by_level[{lev}]."
        out[pre + "count"] = int(bl.count)
        out[pre + "num_nonempty_lists"] = int(bl.num_nonempty_lists)
        for name in ("starts", "lists", "nonempty_indices", "compressed_indices"):
            out[pre + name] = array_sum(torch, getattr(bl, name))
        out[f"target_boxes_sep_smaller_by_source_level[{lev}]"] = array_sum(
            torch, trav.target_boxes_sep_smaller_by_source_level[lev])
    return out


def diff(got, want):
    """Names whose values differ (for the assertion message)."""
    return sorted(k for k in set(got) | set(want) if got.get(k, "<missing>") != want.get(k, "<missing>"))
