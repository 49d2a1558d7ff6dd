"""Area queries: ``PeerListFinder``, ``AreaQueryBuilder``,
``LeavesToBallsLookupBuilder``, ``SpaceInvaderQueryBuilder`` with the reference's
call surface (boxtree/area_query.py:65-170, :660-1192) in front of the gfx950
kernels of ``csrc/bt_area_query.hip``.
"""

from __future__ import annotations

import ctypes as ct
import logging
from dataclasses import dataclass
from typing import Any

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import HIPArrayContext, np_dtype_of, ptr
from boxtree_amd.tools import DoneEvent
from boxtree_amd.tree import _Container

logger = logging.getLogger(__name__)

__all__ = [
    "AreaQueryBuilder", "AreaQueryResult", "LeavesToBallsLookup",
    "LeavesToBallsLookupBuilder", "PeerListFinder", "PeerListLookup",
    "SpaceInvaderQueryBuilder",
]


# {{{ output containers (area_query.py:65-170)

@dataclass(frozen=True)
class PeerListLookup(_Container):
    """``peer_lists[peer_list_starts[box_id]:peer_list_starts[box_id+1]]`` are the
    peer boxes of *box_id* (area_query.py:65-93)."""
    tree: Any
    peer_list_starts: Any
    peer_lists: Any


@dataclass(frozen=True)
class AreaQueryResult(_Container):
    """``leaves_near_ball_lists[leaves_near_ball_starts[i]:...[i+1]]`` are the leaf
    boxes that overlap ball *i* (area_query.py:96-126)."""
    tree: Any
    leaves_near_ball_starts: Any
    leaves_near_ball_lists: Any


@dataclass(frozen=True)
class LeavesToBallsLookup(_Container):
    """``balls_near_box_lists[balls_near_box_starts[ibox]:...[ibox+1]]`` are the
    balls that overlap leaf box *ibox* (area_query.py:129-160)."""
    tree: Any
    balls_near_box_starts: Any
    balls_near_box_lists: Any

# }}}


def _aq_tree(actx, tree, need_levels):
    """bt_aq_tree for *tree*; returns (struct, keep-alive list)."""
    coord_dtype = np.dtype(tree.coord_dtype)
    keep = []

    def dev(a, dtype=None):
        t = actx.from_numpy(a) if isinstance(a, np.ndarray) else a
        if dtype is not None:
            # a TreeOfBoxes from boxtree.tree_of_boxes carries int32 levels
            t = t.to(dtype)
        t = t.contiguous()
        keep.append(t)
        return t

    t = _lib.AqTree()
    t.dims = int(tree.dimensions)
    t.coord_kind = _lib.BT_F64 if coord_dtype == np.float64 else _lib.BT_F32
    t.nlevels = int(tree.nlevels)
    t.nboxes = int(tree.nboxes)
    t.aligned_nboxes = int(tree.aligned_nboxes)
    t.root_extent = float(coord_dtype.type(tree.root_extent))
    bbox_min = tree.bounding_box[0]
    for d in range(t.dims):
        t.bbox_min[d] = float(coord_dtype.type(bbox_min[d]))
    box_centers = dev(tree.box_centers)
    if np_dtype_of(box_centers) != coord_dtype:
        raise TypeError("tree.box_centers dtype must match tree.coord_dtype")
    t.box_centers = ptr(box_centers)
    torch = actx.torch
    t.box_levels = ptr(dev(tree.box_levels, torch.uint8))
    t.box_child_ids = ptr(dev(tree.box_child_ids, torch.int32))
    t.box_flags = ptr(dev(tree.box_flags, torch.uint8))
    if need_levels:
        parents = dev(tree.box_parent_ids, torch.int32)
        if int(parents[0]) != 0:            # root parent -1 in a TreeOfBoxes
            parents = parents.clone()
            parents[0] = 0
            keep.append(parents)
        t.box_parent_ids = ptr(parents)
        from boxtree_amd.tree import level_start_box_nrs_of
        lsb = level_start_box_nrs_of(actx, tree)
        keep.append(lsb)
        t.level_start_box_nrs = lsb.ctypes.data_as(ct.POINTER(ct.c_int32))
    return t, keep


def _call(actx, code):
    lib = actx.lib
    if code == _lib.BT_ERR_UNSUPPORTED:
        raise NotImplementedError(lib.bt_last_error_string().decode())
    if code == _lib.BT_ERR_INVALID:
        raise ValueError(lib.bt_last_error_string().decode())
    _lib.check(code)


def _export_csr(actx, nrows, nentries):
    starts = actx.empty(nrows + 1, np.int32)
    lists = actx.empty(int(nentries), np.int32)
    _call(actx, actx.lib.bt_csr_export(actx.handle, ptr(starts), ptr(lists)))
    return starts, lists


def _check_balls(actx, tree, ball_centers, ball_radii):
    """area_query.py:764-768 (same checks in all three builders)."""
    coord_dtype = np.dtype(tree.coord_dtype)
    dtypes = {np_dtype_of(bc) for bc in ball_centers}
    if len(dtypes) != 1 or dtypes.pop() != coord_dtype:
        raise TypeError("ball_centers dtype must match tree.coord_dtype")
    if np_dtype_of(ball_radii) != coord_dtype:
        raise TypeError("ball_radii dtype must match tree.coord_dtype")
    if len(ball_centers) != tree.dimensions:
        raise ValueError("ball_centers must have one array per dimension")
    centers = [(actx.from_numpy(bc) if isinstance(bc, np.ndarray) else bc).contiguous()
               for bc in ball_centers]
    radii = (actx.from_numpy(ball_radii)
             if isinstance(ball_radii, np.ndarray) else ball_radii).contiguous()
    nballs = int(radii.shape[0])
    if any(int(c.shape[0]) != nballs for c in centers):
        raise ValueError("ball_centers and ball_radii must have the same length")
    arr = (ct.c_void_p * len(centers))(*[c.data_ptr() for c in centers])
    return centers, radii, nballs, arr


def _peer_arrays(actx, tree, peer_lists):
    if len(peer_lists.peer_list_starts) != tree.nboxes + 1:          # :781-782
        raise ValueError("size of peer lists must match with number of boxes")
    starts = peer_lists.peer_list_starts
    lists = peer_lists.peer_lists
    starts = (actx.from_numpy(starts) if isinstance(starts, np.ndarray) else starts).contiguous()
    lists = (actx.from_numpy(lists) if isinstance(lists, np.ndarray) else lists).contiguous()
    if np_dtype_of(starts) != np.int32 or np_dtype_of(lists) != np.int32:
        raise TypeError("peer lists must be int32")
    return starts, lists


# {{{ peer list build (area_query.py:1063-1192)

class PeerListFinder:
    """Builds the look-up table from box numbers to peer boxes.  A peer of box *b*
    is adjacent to or overlaps *b*, is at least as large as *b* or a leaf, and has
    no child with those properties (area_query.py:1067-1096)."""

    def __init__(self, array_context: HIPArrayContext) -> None:
        self._setup_actx = array_context

    def __call__(self, actx: HIPArrayContext, tree, wait_for=None):
        """:returns: a tuple *(pl, event)*, *pl* a :class:`PeerListLookup`."""
        assert isinstance(actx, HIPArrayContext)
        t, keep = _aq_tree(actx, tree, need_levels=True)
        n = ct.c_int64(0)
        actx.sync_in()
        _call(actx, actx.lib.bt_peer_lists_build(actx.handle, ct.byref(t), ct.byref(n)))
        starts, lists = _export_csr(actx, int(tree.nboxes), n.value)
        del keep
        lookup = PeerListLookup(tree=tree, peer_list_starts=starts, peer_lists=lists)
        return actx.freeze(lookup), DoneEvent()

# }}}


# {{{ area query build (area_query.py:660-812)

class AreaQueryBuilder:
    r"""Given a set of :math:`l^\infty` "balls", finds for each ball the leaf boxes
    that intersect it (area_query.py:660-680)."""

    def __init__(self, array_context: HIPArrayContext) -> None:
        self._setup_actx = array_context
        self.peer_list_finder = PeerListFinder(array_context)

    def __call__(self, actx: HIPArrayContext, tree, ball_centers, ball_radii,
                 peer_lists=None, wait_for=None):
        """:returns: a tuple *(aq, event)*, *aq* an :class:`AreaQueryResult`."""
        assert isinstance(actx, HIPArrayContext)
        centers, radii, nballs, arr = _check_balls(actx, tree, ball_centers, ball_radii)
        if peer_lists is None:
            peer_lists, _ = self.peer_list_finder(actx, tree, wait_for=wait_for)
        pl_starts, pl_lists = _peer_arrays(actx, tree, peer_lists)
        t, keep = _aq_tree(actx, tree, need_levels=False)
        n = ct.c_int64(0)
        actx.sync_in()
        _call(actx, actx.lib.bt_area_query_build(
            actx.handle, ct.byref(t), ptr(pl_starts), ptr(pl_lists), nballs, arr,
            ptr(radii), ct.byref(n)))
        starts, lists = _export_csr(actx, nballs, n.value)
        del keep, centers
        result = AreaQueryResult(tree=tree, leaves_near_ball_starts=starts,
                                 leaves_near_ball_lists=lists)
        return actx.freeze(result), DoneEvent()

# }}}


# {{{ area query transpose (leaves-to-balls) lookup build (area_query.py:817-924)

class LeavesToBallsLookupBuilder:
    r"""Given a set of :math:`l^\infty` "balls", builds the look-up table from leaf
    boxes to the balls that overlap them (area_query.py:819-826)."""

    def __init__(self, array_context: HIPArrayContext) -> None:
        self._setup_actx = array_context
        self.area_query_builder = AreaQueryBuilder(array_context)

    def __call__(self, actx: HIPArrayContext, tree, ball_centers, ball_radii,
                 peer_lists=None, wait_for=None):
        """:returns: a tuple *(lbl, event)*, *lbl* a :class:`LeavesToBallsLookup`."""
        assert isinstance(actx, HIPArrayContext)
        _check_balls(actx, tree, ball_centers, ball_radii)
        area_query, _ = self.area_query_builder(
            actx, tree, ball_centers, ball_radii, peer_lists, wait_for)
        nballs = len(area_query.leaves_near_ball_starts) - 1
        nboxes = int(tree.nboxes)
        nentries = int(area_query.leaves_near_ball_lists.shape[0])
        starts = actx.empty(nboxes + 1, np.int32)
        lists = actx.empty(nentries, np.int32)
        actx.sync_in()
        _call(actx, actx.lib.bt_leaves_to_balls(
            actx.handle, nballs, nboxes, ptr(area_query.leaves_near_ball_starts),
            ptr(area_query.leaves_near_ball_lists), nentries, ptr(starts), ptr(lists)))
        lookup = LeavesToBallsLookup(tree=tree, balls_near_box_starts=starts,
                                     balls_near_box_lists=lists)
        return actx.freeze(lookup), DoneEvent()

# }}}


# {{{ space invader query build (area_query.py:929-1056)

class SpaceInvaderQueryBuilder:
    r"""Given a set of :math:`l^\infty` "balls", maps every leaf box to its *outer
    space invader distance*: the largest centre-to-centre :math:`l^\infty` distance
    to a ball that intersects it, or 0 (area_query.py:931-949)."""

    def __init__(self, array_context: HIPArrayContext) -> None:
        self._setup_actx = array_context
        self.peer_list_finder = PeerListFinder(array_context)

    def __call__(self, actx: HIPArrayContext, tree, ball_centers, ball_radii,
                 peer_lists=None, wait_for=None):
        """:returns: a tuple *(sqi, event)*; *sqi* has *tree.coord_dtype* and shape
            *(tree.nboxes,)*: 0 for non-leaf boxes, the outer space invader
            distance for leaves."""
        assert isinstance(actx, HIPArrayContext)
        centers, radii, nballs, arr = _check_balls(actx, tree, ball_centers, ball_radii)
        if peer_lists is None:
            peer_lists, _ = self.peer_list_finder(actx, tree, wait_for=wait_for)
        pl_starts, pl_lists = _peer_arrays(actx, tree, peer_lists)
        t, keep = _aq_tree(actx, tree, need_levels=False)
        out = actx.empty(int(tree.nboxes), np.float32)
        actx.sync_in()
        _call(actx, actx.lib.bt_space_invader_query(
            actx.handle, ct.byref(t), ptr(pl_starts), ptr(pl_lists), nballs, arr,
            ptr(radii), ptr(out)))
        del keep, centers
        if np.dtype(tree.coord_dtype) != np.float32:
            # the kernel result is float32 like the reference's (float atomics);
            # cast to the coordinate type afterwards (area_query.py:1044-1051)
            import torch
            out = out.to(torch.float64)
        return out, DoneEvent()

# }}}

# vim: foldmethod=marker
