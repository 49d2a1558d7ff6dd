"""Work partition of a global traversal over the ranks, and the box masks each
rank derives from its share -- the interface of boxtree/distributed/partition.py
(:39-357) on the device.

The reference walks the tree on the host (a Python stack and a per-box loop on
the root rank) and ships the answer with ``MPI_Scatter``; here the depth-first
order is two level sweeps, the cut is a prefix sum plus one binary search per
rank, and the masks are one kernel per interaction list.  *comm* is
``torch.distributed`` (one process per GPU) or any object with its collective
calls.
"""

from __future__ import annotations

import ctypes as ct
from dataclasses import dataclass
from typing import Any

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import ptr
from boxtree_amd.tree import level_start_box_nrs_of

__all__ = ["get_box_ids_dfs_order", "partition_work", "BoxMasks", "get_box_masks",
           "get_ancestor_boxes_mask", "get_point_src_boxes_mask",
           "get_multipole_src_boxes_mask"]


def _call(actx, code):
    if code == _lib.BT_ERR_INVALID:
        raise ValueError(actx.lib.bt_last_error_string().decode())
    _lib.check(code)


def get_box_ids_dfs_order(actx, tree):
    """Box ids in the depth-first order of partition.py:39-57 (children of a box are
    visited in descending child number, as the reference's stack pops them)."""
    lev = level_start_box_nrs_of(actx, tree)
    out = actx.empty(int(tree.nboxes), np.int32)
    actx.sync_in()
    _call(actx, actx.lib.bt_dfs_order(
        actx.handle, int(tree.box_child_ids.shape[0]), len(lev) - 1,
        lev.ctypes.data_as(ct.POINTER(ct.c_int32)), int(tree.nboxes), int(tree.aligned_nboxes),
        ptr(tree.box_child_ids), ptr(out)))
    return out


def partition_work(actx, cost_per_box, traversal, comm):
    """Assigns every rank a consecutive run of the depth-first box order with about
    1/size of the total cost (partition.py:60-121) and returns this rank's boxes.

    :arg cost_per_box: device float array [nboxes]; like upstream only rank 0's copy
        decides, its segment table is broadcast so that all ranks agree bit for bit.
    """
    tree = traversal.tree
    size, rank = comm.get_world_size(), comm.get_rank()
    nboxes = int(tree.nboxes)
    if size > nboxes:
        raise RuntimeError("Fail to partition work because the number of boxes is "
                           "less than the number of processes.")
    dfs_order = get_box_ids_dfs_order(actx, tree)
    seg = np.zeros((size, 2), dtype=np.int32)
    if rank == 0:
        cost = cost_per_box.to(actx.torch.float64).contiguous()
        actx.sync_in()
        _call(actx, actx.lib.bt_partition_work(
            actx.handle, nboxes, ptr(dfs_order), ptr(cost), size,
            seg.ctypes.data_as(ct.POINTER(ct.c_int32))))
    if size > 1:
        seg_dev = actx.from_numpy(seg)
        comm.broadcast(seg_dev, src=0)
        seg = actx.to_numpy(seg_dev)
    start, end = int(seg[rank, 0]), int(seg[rank, 1])
    return dfs_order[start:end].contiguous()


@dataclass(frozen=True)
class BoxMasks:
    """int8 masks over box numbers (partition.py:301-330): the boxes this rank
    evaluates, their ancestors, the boxes whose sources and the boxes whose
    multipole expansions it needs."""
    responsible_boxes: Any
    ancestor_boxes: Any
    point_src_boxes: Any
    multipole_src_boxes: Any


def _mark(actx, box_list, mask_a, mask_b, starts, lists, out):
    nrows = int(box_list.shape[0])
    if nrows == 0 or lists is None or int(lists.shape[0]) == 0:
        return
    actx.sync_in()
    _call(actx, actx.lib.bt_mark_list_boxes(
        actx.handle, nrows, ptr(box_list), ptr(mask_a), ptr(mask_b), ptr(starts),
        ptr(lists.contiguous()), ptr(out)))


def get_ancestor_boxes_mask(actx, traversal, responsible_boxes_mask):
    """partition.py:167-188: the proper ancestors of the boxes in the mask."""
    tree = traversal.tree
    out = actx.empty(int(tree.nboxes), np.int8)
    actx.sync_in()
    _call(actx, actx.lib.bt_ancestor_mask(
        actx.handle, int(tree.nboxes), ptr(tree.box_parent_ids), ptr(responsible_boxes_mask),
        ptr(out)))
    return out


def get_point_src_boxes_mask(actx, traversal, responsible_boxes_mask, ancestor_boxes_mask):
    """partition.py:191-245: own boxes, their list 1, list 4 of own boxes and ancestors,
    and the 'close' lists when targets have extent."""
    trav = traversal
    out = responsible_boxes_mask.clone()
    _mark(actx, trav.target_boxes, responsible_boxes_mask, None,
          trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists, out)
    _mark(actx, trav.target_or_target_parent_boxes, responsible_boxes_mask, ancestor_boxes_mask,
          trav.from_sep_bigger_starts, trav.from_sep_bigger_lists, out)
    if trav.tree.targets_have_extent:
        if trav.from_sep_close_smaller_starts is not None:
            _mark(actx, trav.target_boxes, responsible_boxes_mask, None,
                  trav.from_sep_close_smaller_starts, trav.from_sep_close_smaller_lists, out)
        if trav.from_sep_close_bigger_starts is not None:
            _mark(actx, trav.target_boxes, responsible_boxes_mask, ancestor_boxes_mask,
                  trav.from_sep_close_bigger_starts, trav.from_sep_close_bigger_lists, out)
    return out


def get_multipole_src_boxes_mask(actx, traversal, responsible_boxes_mask, ancestor_boxes_mask):
    """partition.py:248-298: list 2 of own boxes and ancestors, list 3 of own boxes."""
    trav = traversal
    out = actx.zeros(int(trav.tree.nboxes), np.int8)
    _mark(actx, trav.target_or_target_parent_boxes, responsible_boxes_mask, ancestor_boxes_mask,
          trav.from_sep_siblings_starts, trav.from_sep_siblings_lists, out)
    for ilevel in range(int(trav.tree.nlevels)):
        ssn = trav.from_sep_smaller_by_level[ilevel]
        _mark(actx, trav.target_boxes_sep_smaller_by_source_level[ilevel],
              responsible_boxes_mask, None, ssn.starts, ssn.lists, out)
    return out


def get_box_masks(actx, traversal, responsible_boxes_list):
    """partition.py:333-357."""
    responsible = actx.zeros(int(traversal.tree.nboxes), np.int8)
    responsible[responsible_boxes_list.long()] = 1
    ancestors = get_ancestor_boxes_mask(actx, traversal, responsible)
    point_src = get_point_src_boxes_mask(actx, traversal, responsible, ancestors)
    mpole_src = get_multipole_src_boxes_mask(actx, traversal, responsible, ancestors)
    return BoxMasks(responsible, ancestors, point_src, mpole_src)
