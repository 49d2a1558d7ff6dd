"""``FMMTraversalBuilder`` / ``FMMTraversalInfo``: the reference's call surface
(boxtree/traversal.py:1353-2345) in front of the gfx950 list-building kernels.
"""

from __future__ import annotations

import ctypes as ct
import dataclasses
import logging
import os
from dataclasses import dataclass
from typing import Any

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import HIPArrayContext, make_obj_array, np_dtype_of, ptr
from boxtree_amd.tools import DoneEvent, StreamEvent
from boxtree_amd.tree import _Container

logger = logging.getLogger(__name__)


@dataclass(frozen=True)
class BuiltList(_Container):
    """CSR list with optional empty-list elimination; fields as produced by
    ``pyopencl.algorithm.ListOfListsBuilder`` (boxtree/array_context.py:222-238)."""
    count: Any
    starts: Any
    lists: Any
    num_nonempty_lists: Any = None
    nonempty_indices: Any = None
    compressed_indices: Any = None


@dataclass(frozen=True)
class FMMTraversalInfo(_Container):
    """Interaction lists; field-for-field boxtree/traversal.py:1595-1630."""
    tree: Any
    well_sep_is_n_away: int

    source_boxes: Any
    target_boxes: Any
    level_start_source_box_nrs: Any
    level_start_target_box_nrs: Any
    source_parent_boxes: Any
    level_start_source_parent_box_nrs: Any
    target_or_target_parent_boxes: Any
    level_start_target_or_target_parent_box_nrs: Any

    same_level_non_well_sep_boxes_starts: Any
    same_level_non_well_sep_boxes_lists: Any

    neighbor_source_boxes_starts: Any
    neighbor_source_boxes_lists: Any

    from_sep_siblings_starts: Any
    from_sep_siblings_lists: Any

    from_sep_smaller_by_level: Any
    target_boxes_sep_smaller_by_source_level: Any
    from_sep_close_smaller_starts: Any
    from_sep_close_smaller_lists: Any

    from_sep_bigger_starts: Any
    from_sep_bigger_lists: Any
    from_sep_close_bigger_starts: Any
    from_sep_close_bigger_lists: Any

    @property
    def nboxes(self):
        return self.tree.nboxes

    @property
    def nlevels(self):
        return self.tree.nlevels

    @property
    def ntarget_boxes(self):
        return len(self.target_boxes)

    @property
    def ntarget_or_target_parent_boxes(self):
        return len(self.target_or_target_parent_boxes)

    def merge_close_lists(self, actx, debug=False):
        """Return a new :class:`FMMTraversalInfo` with the "close" lists merged
        into list 1 and set to *None* (traversal.py:1650-1693)."""
        import torch
        starts = [self.neighbor_source_boxes_starts, self.from_sep_close_smaller_starts,
                  self.from_sep_close_bigger_starts]
        lists = [self.neighbor_source_boxes_lists, self.from_sep_close_smaller_lists,
                 self.from_sep_close_bigger_lists]
        on_host = not isinstance(starts[0], torch.Tensor)
        dstarts = [actx.from_numpy(np.ascontiguousarray(a)) if on_host else a.contiguous()
                   for a in starts]
        dlists = [actx.from_numpy(np.ascontiguousarray(a)) if on_host else a.contiguous()
                  for a in lists]
        nrows = len(dstarts[0]) - 1
        total = sum(len(x) for x in dlists)
        new_starts = actx.empty(nrows + 1, np.int32)
        new_lists = actx.empty(total, np.int32)
        sp = (ct.c_void_p * 3)(*[x.data_ptr() for x in dstarts])
        lp = (ct.c_void_p * 3)(*[x.data_ptr() for x in dlists])
        actx.sync_in()
        _lib.check(actx.lib.bt_merge_csr_lists(
            actx.handle, 3, sp, lp, nrows, ptr(new_starts), ptr(new_lists)))
        if on_host:
            new_starts, new_lists = actx.to_numpy(new_starts), actx.to_numpy(new_lists)
        return dataclasses.replace(
            self,
            neighbor_source_boxes_starts=new_starts,
            neighbor_source_boxes_lists=new_lists,
            from_sep_close_smaller_starts=None,
            from_sep_close_smaller_lists=None,
            from_sep_close_bigger_starts=None,
            from_sep_close_bigger_lists=None)

    def get_box_list(self, what, index):
        starts = getattr(self, f"{what}_starts")
        lists = getattr(self, f"{what}_lists")
        start, stop = starts[index:index + 2]
        return lists[start:stop]


_SPAN_NAMES = (
    "source_boxes", "target_boxes", "source_parent_boxes", "target_or_target_parent_boxes",
    "same_level_non_well_sep_boxes_starts", "same_level_non_well_sep_boxes_lists",
    "neighbor_source_boxes_starts", "neighbor_source_boxes_lists",
    "from_sep_siblings_starts", "from_sep_siblings_lists",
    "from_sep_bigger_starts", "from_sep_bigger_lists",
    "from_sep_close_smaller_starts", "from_sep_close_smaller_lists",
    "from_sep_close_bigger_starts", "from_sep_close_bigger_lists")
_LEVEL_START_NAMES = (
    "level_start_source_box_nrs", "level_start_target_box_nrs",
    "level_start_source_parent_box_nrs", "level_start_target_or_target_parent_box_nrs")


def _info_from_packed(tree, well_sep_is_n_away, packed, buf, nlevels, share_target_boxes):
    """FMMTraversalInfo whose arrays are views into *buf*, the one int32 block the library
    filled; *packed* (bt_trav_packed) says where each array lives.  The struct is read
    through numpy in three pieces (ctypes attribute access costs about a microsecond per
    field and there are some 60 spans)."""
    P = _lib.TravPacked
    nspan = len(_SPAN_NAMES)
    L = _lib.BT_MAX_LEVELS
    spans = np.frombuffer(packed, dtype=np.int64, count=2 * (nspan + 5 * L),
                          offset=P.source_boxes.offset).reshape(-1, 2).tolist()
    lev = np.frombuffer(packed, dtype=np.int32, count=4 * (L + 1),
                        offset=P.level_start_source_box_nrs.offset).reshape(4, L + 1)
    S = _lib.TravSizes
    per_level = np.frombuffer(packed, dtype=np.int64, count=2 * L,
                              offset=P.sizes.offset + S.n_from_sep_smaller.offset
                              ).reshape(2, L)[:, :nlevels].tolist()

    def view(i):
        off, count = spans[i]
        return buf[off:off + count]

    kw = {name: view(i) for i, name in enumerate(_SPAN_NAMES)}
    if share_target_boxes:
        kw["target_boxes"] = kw["source_boxes"]
    if not (tree.sources_have_extent or tree.targets_have_extent):
        for name in _SPAN_NAMES[12:]:
            kw[name] = None
    # host arrays, as in the reference (traversal.py:2091 actx.to_numpy(result))
    for i, name in enumerate(_LEVEL_START_NAMES):
        kw[name] = lev[i, :nlevels + 1].copy()

    base = nspan
    by_level = np.empty(nlevels, dtype=object)
    tboxes = np.empty(nlevels, dtype=object)
    counts, nonempty = per_level
    for ilev in range(nlevels):
        by_level[ilev] = BuiltList(
            count=counts[ilev], starts=view(base + ilev), lists=view(base + L + ilev),
            num_nonempty_lists=nonempty[ilev], nonempty_indices=view(base + 2 * L + ilev),
            compressed_indices=view(base + 3 * L + ilev))
        tboxes[ilev] = view(base + 4 * L + ilev)

    return FMMTraversalInfo(
        tree=tree, well_sep_is_n_away=well_sep_is_n_away,
        from_sep_smaller_by_level=by_level,
        target_boxes_sep_smaller_by_source_level=tboxes, **kw)


class FMMTraversalBuilder:
    def __init__(self, array_context, *, well_sep_is_n_away=1,
                 from_sep_smaller_crit=None):
        """
        :arg well_sep_is_n_away: an integer 1 or greater (traversal.py:1731-1745).
        """
        assert isinstance(array_context, HIPArrayContext)
        self._setup_actx = array_context
        self.well_sep_is_n_away = well_sep_is_n_away
        self.from_sep_smaller_crit = from_sep_smaller_crit

    def __call__(self, actx, tree, wait_for=None, debug=False,
                 _from_sep_smaller_min_nsources_cumul=None,
                 source_boxes_mask=None, source_parent_boxes_mask=None,
                 _force_generic=None, _target_boxes_mask=None, _active_level_ranges=None):
        """Same arguments, return value ``(trav, event)`` and exceptions as
        ``FMMTraversalBuilder.__call__`` (traversal.py:1969-1990).

        ``_force_generic`` (or env ``BOXTREE_HIP_FORCE_GENERIC=1``) selects the
        walk-from-root kernels even for trees whose numbering allows the faster
        parent-colleague kernels; both produce identical lists."""
        assert isinstance(actx, HIPArrayContext)
        _lib.host_trace("tg:enter")

        from_sep_smaller_min_nsources_cumul = _from_sep_smaller_min_nsources_cumul
        if from_sep_smaller_min_nsources_cumul is None:
            from_sep_smaller_min_nsources_cumul = 0     # traversal.py:1995-1997

        if not tree._is_pruned:
            raise ValueError("tree must be pruned for traversal generation")
        if tree.sources_have_extent:
            raise NotImplementedError(
                "trees with source extent are not supported for "
                "traversal generation")

        # traversal.py:1776-1805
        from_sep_smaller_crit = self.from_sep_smaller_crit
        if from_sep_smaller_crit is None:
            from_sep_smaller_crit = "precise_linf"
        extent_norm = tree.extent_norm
        if extent_norm == "linf":
            pass
        elif extent_norm == "l2":
            if from_sep_smaller_crit == "static_linf":
                raise ValueError(
                    "the static l^inf from-sep-smaller criterion "
                    "cannot be used with the l^2 extent norm")
        elif extent_norm is None:
            assert not (tree.sources_have_extent or tree.targets_have_extent)
        else:
            raise ValueError(f"unexpected value of 'extent_norm': {extent_norm}")
        if from_sep_smaller_crit not in ["static_linf", "precise_linf", "static_l2"]:
            raise ValueError(
                "unexpected value of 'from_sep_smaller_crit': "
                f"{from_sep_smaller_crit}")

        def dev(a):
            if a is None:
                return None
            from boxtree_amd.array_context import as_device_array
            return as_device_array(actx, a).contiguous()

        nlevels = int(tree.nlevels)
        sources_are_targets = getattr(tree, "sources_are_targets", True)
        coord_dtype = np.dtype(tree.coord_dtype)
        dims = int(tree.dimensions)
        nboxes = int(tree.nboxes)

        # a TreeOfBoxes made by boxtree.tree_of_boxes arrives as numpy arrays with
        # int32 levels and a root whose parent is -1 (tree_of_boxes.py:392-465)
        torch = actx.torch
        subtree_sizes = None
        if getattr(tree, "_host_level_starts", None) is not None:
            # made by TreeBuilder on this device: contiguous arrays of the library's types
            box_centers, box_levels, box_child_ids = (tree.box_centers, tree.box_levels,
                                                      tree.box_child_ids)
            box_flags, box_parent_ids = tree.box_flags, tree.box_parent_ids
            if os.environ.get("BOXTREE_HIP_SUBTREE_SIZES", "1") != "0":
                subtree_sizes = getattr(tree, "_subtree_sizes", None)
        else:
            box_centers = dev(tree.box_centers)
            box_levels = dev(tree.box_levels).to(torch.uint8)
            box_child_ids = dev(tree.box_child_ids).to(torch.int32)
            box_flags = dev(tree.box_flags).to(torch.uint8)
            box_parent_ids = dev(tree.box_parent_ids).to(torch.int32)
            if int(box_parent_ids[0]) != 0:
                box_parent_ids = box_parent_ids.clone()
                box_parent_ids[0] = 0
        assert np_dtype_of(box_centers) == coord_dtype
        from boxtree_amd.tree import level_start_box_nrs_of
        lsb = level_start_box_nrs_of(actx, tree)

        _lib.host_trace("tg:arrays")
        tp = _lib.TravParams()
        tp.dims = dims
        tp.coord_kind = _lib.BT_F64 if coord_dtype == np.float64 else _lib.BT_F32
        tp.nlevels = nlevels
        tp.nboxes = nboxes
        tp.aligned_nboxes = int(tree.aligned_nboxes)
        tp.root_extent = float(coord_dtype.type(tree.root_extent))
        tp.stick_out_factor = float(coord_dtype.type(tree.stick_out_factor))
        tp.box_centers = ptr(box_centers)
        tp.box_levels = ptr(box_levels)
        tp.box_child_ids = ptr(box_child_ids)
        tp.box_flags = ptr(box_flags)
        tp.box_parent_ids = ptr(box_parent_ids)
        tp.box_subtree_sizes = ptr(subtree_sizes)
        keep = []
        if tree.targets_have_extent:
            for name in ("box_target_bounding_box_min", "box_target_bounding_box_max",
                         "box_source_counts_cumul"):
                t = dev(getattr(tree, name))
                keep.append(t)
                setattr(tp, name, ptr(t))
        tp.level_start_box_nrs = lsb.ctypes.data_as(ct.POINTER(ct.c_int32))
        tp.sources_are_targets = int(bool(sources_are_targets))
        tp.sources_have_extent = int(bool(tree.sources_have_extent))
        tp.targets_have_extent = int(bool(tree.targets_have_extent))
        tp.well_sep_is_n_away = int(self.well_sep_is_n_away)
        tp.from_sep_smaller_crit = _lib.CRITS[from_sep_smaller_crit]
        tp.from_sep_smaller_min_nsources_cumul = int(from_sep_smaller_min_nsources_cumul)
        sbm = dev(source_boxes_mask)
        spbm = dev(source_parent_boxes_mask)
        tp.source_boxes_mask = ptr(sbm)
        tp.source_parent_boxes_mask = ptr(spbm)
        if _force_generic is None:
            _force_generic = os.environ.get("BOXTREE_HIP_FORCE_GENERIC", "0") == "1"
        # True: walk-from-root kernels; "float": parent-colleague kernels with the float
        # predicates; False: the default choice (integer-lattice form where it applies)
        tp.force_generic = 2 if _force_generic == "float" else int(bool(_force_generic))
        # sharded traversals (boxtree_amd/distributed/__init__.py step 6): lists of a subset
        # of the target boxes of a tree whose box arrays are complete
        tbm = dev(_target_boxes_mask)
        tp.target_boxes_mask = ptr(tbm)
        alr = None
        if _active_level_ranges is not None:
            alr = np.ascontiguousarray(_active_level_ranges, dtype=np.int32).reshape(-1)
            assert alr.shape[0] == 2 * nlevels
            tp.active_level_ranges = alr.ctypes.data_as(ct.POINTER(ct.c_int32))

        lib = actx.lib
        # One int32 block for every output array: the library asks for it once all list
        # sizes are known (one allocation instead of ~70), builds the large lists in
        # place, and reports where each array lives (bt_trav_packed).
        block = []

        def alloc(_user, nbytes):
            block.append(actx.empty(int(nbytes) // 4, np.int32))
            return block[0].data_ptr()

        packed = _lib.TravPacked()
        actx.sync_in()
        code = lib.bt_traversal_build_packed(actx.handle, ct.byref(tp), _lib.ALLOC_FN(alloc), None,
                                             ct.byref(packed))
        if code == _lib.BT_ERR_UNSUPPORTED:
            raise NotImplementedError(lib.bt_last_error_string().decode())
        if code == _lib.BT_ERR_INVALID:
            raise ValueError(lib.bt_last_error_string().decode())
        _lib.check(code)
        _lib.host_trace("tg:built")
        info = _info_from_packed(tree, self.well_sep_is_n_away, packed, block[0], nlevels,
                                 share_target_boxes=sources_are_targets and tbm is None)
        if debug:
            # (the reference waits after every stage and compiles its walks with a stack check,
            # traversal.py:150-156, 2035-2039; here: wait, then every list must be a well-formed CSR)
            from boxtree_amd.debug import check_traversal
            actx.synchronize()
            check_traversal(actx.torch, info, nboxes)
        return actx.freeze(info), (StreamEvent(actx) if actx.stream_ordered else DoneEvent())

# vim: fdm=marker
