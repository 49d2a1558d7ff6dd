"""Output containers of the tree build: same names/layouts as boxtree/tree.py."""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from functools import cached_property
from typing import Any

import numpy as np


class box_flags_enum:  # noqa: N801  (name kept from boxtree/tree.py:109-145)
    """Constants for the box flags bit field."""
    c_name = "box_flags_t"
    dtype = np.dtype(np.uint8)
    c_value_prefix = "BOX_"

    IS_SOURCE_BOX = 1 << 0
    IS_TARGET_BOX = 1 << 1
    IS_SOURCE_OR_TARGET_BOX = IS_SOURCE_BOX | IS_TARGET_BOX
    HAS_SOURCE_CHILD_BOXES = 1 << 2
    HAS_TARGET_CHILD_BOXES = 1 << 3
    HAS_SOURCE_OR_TARGET_CHILD_BOXES = HAS_SOURCE_CHILD_BOXES | HAS_TARGET_CHILD_BOXES
    IS_LEAF_BOX = 1 << 4          # TreeOfBoxes only (tree.py:141-142)
    HAS_CHILDREN = HAS_SOURCE_OR_TARGET_CHILD_BOXES


class _Container:
    def _map_arrays(self, f):
        """Return a copy with *f* applied to every field (used by to_numpy)."""
        kw = {}
        memo = {}
        for fld in dataclasses.fields(self):
            v = getattr(self, fld.name)
            key = id(v)
            if key not in memo:
                memo[key] = f(v)      # preserves aliasing (sources is targets, ...)
            kw[fld.name] = memo[key]
        return type(self)(**kw)


@dataclass(frozen=True)
class TreeOfBoxes(_Container):
    """A quad/octree of pure boxes (boxtree/tree.py:154-289); accepted by the
    traversal builder."""
    root_extent: Any
    box_centers: Any
    box_parent_ids: Any
    box_child_ids: Any
    box_levels: Any
    box_flags: Any
    level_start_box_nrs: Any
    box_id_dtype: Any
    box_level_dtype: Any
    coord_dtype: Any
    sources_have_extent: bool
    targets_have_extent: bool
    extent_norm: Any
    stick_out_factor: Any
    _is_pruned: bool

    @property
    def dimensions(self):
        return self.box_centers.shape[0]

    @property
    def nboxes(self):
        return len(self.box_levels)

    @property
    def aligned_nboxes(self):
        return self.box_child_ids.shape[-1]

    @property
    def nlevels(self):
        if self.level_start_box_nrs is not None:
            return len(self.level_start_box_nrs) - 1
        return int(self.box_levels.max()) + 1

    # helpers of the reference's class (boxtree/tree.py:266-287); numpy-backed trees
    @property
    def leaf_boxes(self):
        boxes = np.arange(self.nboxes, dtype=self.box_id_dtype)
        return boxes[self.box_flags & box_flags_enum.IS_LEAF_BOX != 0]

    # (a cached_property, not a property, as in the reference: Tree below replaces it by a FIELD, and
    # a data descriptor on the base class would refuse the frozen dataclass's object.__setattr__)
    @cached_property
    def bounding_box(self):
        # host values also for device-resident box arrays (local essential trees)
        c0 = self.box_centers[:, 0]
        if not isinstance(c0, np.ndarray):
            c0 = c0.detach().cpu().numpy()
        lows = c0 - 0.5 * self.root_extent
        return lows, lows + self.root_extent

    def get_box_size(self, ibox):
        return self.root_extent * 0.5 ** int(self.box_levels[ibox])

    def get_box_extent(self, ibox):
        box_size = self.get_box_size(ibox)
        extent_low = self.box_centers[:, ibox] - 0.5 * box_size
        return extent_low, extent_low + box_size


@dataclass(frozen=True)
class Tree(TreeOfBoxes):
    """Particles sorted into a hierarchy of boxes; field-for-field the
    reference's :class:`boxtree.Tree` (boxtree/tree.py:298-686), and like it a
    :class:`TreeOfBoxes` (tree.py:298 ``class Tree(TreeOfBoxes)``)."""
    sources_are_targets: bool
    sources_have_extent: bool
    targets_have_extent: bool

    particle_id_dtype: Any
    box_id_dtype: Any
    coord_dtype: Any
    box_level_dtype: Any

    # (bbox_min, bbox_max) numpy vectors; an explicit field() so that it replaces the base
    # class's cached property (tree.py:571-574)
    bounding_box: Any = dataclasses.field(init=True)
    root_extent: Any
    stick_out_factor: Any
    extent_norm: Any

    level_start_box_nrs: Any     # int32 [nlevels+1]

    sources: Any                 # object array of d coordinate arrays
    targets: Any
    source_radii: Any
    target_radii: Any

    box_source_starts: Any
    box_source_counts_nonchild: Any
    box_source_counts_cumul: Any
    box_target_starts: Any
    box_target_counts_nonchild: Any
    box_target_counts_cumul: Any

    box_parent_ids: Any
    box_child_ids: Any           # [2^d, aligned_nboxes]
    box_centers: Any             # [d, aligned_nboxes]
    box_levels: Any
    box_flags: Any

    user_source_ids: Any
    sorted_target_ids: Any

    box_source_bounding_box_min: Any
    box_source_bounding_box_max: Any
    box_target_bounding_box_min: Any
    box_target_bounding_box_max: Any

    _is_pruned: bool

    @property
    def dimensions(self):
        return len(self.sources)

    @property
    def nboxes(self):
        return len(self.box_flags)

    @property
    def aligned_nboxes(self):
        return self.box_child_ids.shape[-1]

    @property
    def nsources(self):
        return len(self.sources[0])

    @property
    def ntargets(self):
        return len(self.targets[0])

    @property
    def nlevels(self):
        return len(self.level_start_box_nrs) - 1

    def get_box_extent(self, ibox):
        lev = int(self.box_levels[ibox])
        box_size = self.root_extent / (1 << lev)
        extent_low = self.box_centers[:, ibox] - 0.5 * box_size
        extent_high = extent_low + box_size
        return extent_low, extent_high

    # debugging aids of the reference (tree.py:639-686); numpy arrays only
    def _reverse_index_lookup(self, ary, new_key_size):
        result = np.empty(new_key_size, dtype=ary.dtype)
        result.fill(-1)
        result[ary] = np.arange(len(ary), dtype=ary.dtype)
        return result

    def indices_to_tree_source_order(self, user_indices):
        return self._reverse_index_lookup(self.user_source_ids, self.nsources)[user_indices]

    def indices_to_tree_target_order(self, user_indices):
        return self.sorted_target_ids[user_indices]

    def find_box_nr_for_target(self, itarget):
        crit = ((self.box_target_starts <= itarget)
                & (itarget < self.box_target_starts + self.box_target_counts_nonchild))
        return int(np.where(crit)[0])

    def find_box_nr_for_source(self, isource):
        crit = ((self.box_source_starts <= isource)
                & (isource < self.box_source_starts + self.box_source_counts_nonchild))
        return int(np.where(crit)[0])


# {{{ tree with linked point sources (boxtree/tree.py:690-949)

@dataclass(frozen=True)
class TreeWithLinkedPointSources(Tree):
    """A :class:`Tree` whose (extent-having) sources are expanded into point
    sources; the additional fields of boxtree/tree.py:762-769."""
    npoint_sources: int
    point_source_starts: Any
    point_source_counts: Any
    point_sources: Any
    user_point_source_ids: Any
    box_point_source_starts: Any
    box_point_source_counts_nonchild: Any
    box_point_source_counts_cumul: Any


def level_start_box_nrs_of(actx, tree):
    """Host int32 [nlevels+1] level starts of *tree*; derived from ``box_levels`` when
    the tree does not carry them (a :class:`TreeOfBoxes` may have
    ``level_start_box_nrs=None``, boxtree/tree.py:236) -- which needs the boxes to be
    numbered level by level."""
    cached = getattr(tree, "_host_level_starts", None)
    if cached is not None:
        return cached
    if tree.level_start_box_nrs is not None:
        return np.ascontiguousarray(actx.to_numpy(tree.level_start_box_nrs), dtype=np.int32)
    levels = np.asarray(actx.to_numpy(tree.box_levels), dtype=np.int64)
    if len(levels) == 0 or levels[0] != 0 or np.any(np.diff(levels) < 0):
        raise NotImplementedError(
            "trees without level_start_box_nrs must number their boxes level by level")
    nlevels = int(levels[-1]) + 1
    counts = np.bincount(levels, minlength=nlevels)
    starts = np.zeros(nlevels + 1, dtype=np.int32)
    starts[1:] = np.cumsum(counts)
    return starts


def _gather(actx, src, ids):
    """src[ids] through the library's gather kernel."""
    import ctypes as ct

    from boxtree_amd import _lib
    src = src.contiguous()
    out = actx.torch.empty(ids.shape[0], dtype=src.dtype, device=src.device)
    if ids.shape[0]:
        actx.sync_in()
        _lib.check(actx.lib.bt_gather(
            actx.handle, src.element_size(), ct.c_void_p(src.data_ptr()),
            ct.c_void_p(ids.data_ptr()), int(ids.shape[0]), ct.c_void_p(out.data_ptr())))
    return out


def _dev(actx, a):
    t = actx.from_numpy(a) if isinstance(a, np.ndarray) else a
    return t.contiguous()


def link_point_sources(actx, tree, point_source_starts, point_sources, *, debug=False):
    r"""Links point sources to the extent-having sources of *tree*
    (boxtree/tree.py:772-949).

    :arg point_source_starts: ``point_source_starts[isrc]`` and ``[isrc+1]`` delimit
        the point sources of source *isrc* (user source order) in *point_sources*.
    :arg point_sources: an object array of (XYZ) point coordinate arrays.
    """
    import ctypes as ct

    from boxtree_amd import _lib
    from boxtree_amd.array_context import make_obj_array, np_dtype_of, ptr

    if not tree.sources_have_extent:
        raise ValueError("only allowed on trees whose sources have extent")
    pss = _dev(actx, point_source_starts)
    if np_dtype_of(pss) != np.int32:
        raise TypeError("point_source_starts must have the tree's particle_id_dtype (int32)")
    nsources = int(tree.nsources)
    nboxes = int(tree.nboxes)
    if pss.shape[0] != nsources + 1:
        raise ValueError("point_source_starts must have nsources+1 entries")
    ends = actx.to_numpy(pss[[0, nsources]])
    npoint_sources = int(ends[1]) - int(ends[0])
    e = actx.empty
    i32 = np.int32
    to_starts, to_counts = e(nsources, i32), e(nsources, i32)
    ids = e(npoint_sources, i32)
    bstarts, bnonchild, bcumul = e(nboxes, i32), e(nboxes, i32), e(nboxes, i32)
    usi = _dev(actx, tree.user_source_ids)
    bss = _dev(actx, tree.box_source_starts)
    bsn = _dev(actx, tree.box_source_counts_nonchild)
    bsc = _dev(actx, tree.box_source_counts_cumul)
    actx.sync_in()
    code = actx.lib.bt_link_point_sources(
        actx.handle, nsources, nboxes, npoint_sources, ptr(pss), ptr(usi), ptr(bss), ptr(bsn),
        ptr(bsc), ptr(to_starts), ptr(to_counts), ptr(ids), ptr(bstarts), ptr(bnonchild),
        ptr(bcumul))
    if code == _lib.BT_ERR_INVALID:
        raise ValueError(actx.lib.bt_last_error_string().decode())
    _lib.check(code)
    tree_order_point_sources = make_obj_array(
        # flat storage order, like cl_array.take upstream: the reference's own test
        # passes [nsources, npoint_sources_per_source] arrays (test/test_tree.py:638-646)
        [_gather(actx, _dev(actx, point_sources[i]).reshape(-1), ids)
         for i in range(tree.dimensions)])
    if debug:
        h = actx.to_numpy(ids)
        assert np.all(h >= 0) and np.all(h < npoint_sources)
    tree_attrs = {f.name: getattr(tree, f.name) for f in dataclasses.fields(tree)}
    result = TreeWithLinkedPointSources(
        npoint_sources=npoint_sources,
        point_source_starts=to_starts,
        point_source_counts=to_counts,
        point_sources=tree_order_point_sources,
        user_point_source_ids=ids,
        box_point_source_starts=bstarts,
        box_point_source_counts_nonchild=bnonchild,
        box_point_source_counts_cumul=bcumul,
        **tree_attrs)
    return actx.freeze(result)

# }}}


# {{{ particle list filter (boxtree/tree.py:957-1243)

@dataclass(frozen=True)
class FilteredTargetListsInUserOrder(_Container):
    """Per box the flagged targets, as user-order target numbers (CSR:
    ``target_starts [nboxes+1]``, ``target_lists``); boxtree/tree.py:957-998."""
    nfiltered_targets: int
    target_starts: Any
    target_lists: Any


@dataclass(frozen=True)
class FilteredTargetListsInTreeOrder(_Container):
    """A renumbering of the flagged targets that keeps the targets of a box
    consecutive; boxtree/tree.py:1001-1055."""
    nfiltered_targets: int
    box_target_starts: Any
    box_target_counts_nonchild: Any
    targets: Any
    unfiltered_from_filtered_target_indices: Any


class ParticleListFilter:
    """boxtree/tree.py:1059-1241"""

    def __init__(self, array_context):
        self._setup_actx = array_context

    def _inputs(self, actx, tree, flags):
        from boxtree_amd.array_context import np_dtype_of
        flags = _dev(actx, flags)
        if np_dtype_of(flags) != np.int8:
            raise TypeError("flags must be an array of int8")
        if flags.shape[0] != tree.ntargets:
            raise ValueError("flags must have one entry per target")
        return (flags, _dev(actx, tree.sorted_target_ids), _dev(actx, tree.box_target_starts),
                _dev(actx, tree.box_target_counts_nonchild))

    def filter_target_lists_in_user_order(self, actx, tree, flags):
        """:arg flags: int8 [ntargets] in user target order; nonzero = keep."""
        import ctypes as ct

        from boxtree_amd import _lib
        from boxtree_amd.array_context import ptr
        flags, sti, bts, btc = self._inputs(actx, tree, flags)
        nboxes, ntargets = int(tree.nboxes), int(tree.ntargets)
        starts = actx.empty(nboxes + 1, np.int32)
        lists = actx.empty(ntargets, np.int32)
        n = ct.c_int64(0)
        actx.sync_in()
        _lib.check(actx.lib.bt_filter_targets_user_order(
            actx.handle, nboxes, ntargets, ptr(flags), ptr(sti), ptr(bts), ptr(btc),
            ptr(starts), ptr(lists), ct.byref(n)))
        result = FilteredTargetListsInUserOrder(
            nfiltered_targets=int(n.value), target_starts=starts,
            target_lists=lists[:n.value])
        return actx.freeze(result)

    def filter_target_lists_in_tree_order(self, actx, tree, flags):
        """:arg flags: int8 [ntargets] in user target order; nonzero = keep."""
        import ctypes as ct

        from boxtree_amd import _lib
        from boxtree_amd.array_context import make_obj_array, ptr
        flags, sti, bts, btc = self._inputs(actx, tree, flags)
        nboxes, ntargets = int(tree.nboxes), int(tree.ntargets)
        starts_f = actx.empty(nboxes, np.int32)
        counts_f = actx.empty(nboxes, np.int32)
        uff = actx.empty(ntargets, np.int32)
        n = ct.c_int64(0)
        actx.sync_in()
        _lib.check(actx.lib.bt_filter_targets_tree_order(
            actx.handle, nboxes, ntargets, ptr(flags), ptr(sti), ptr(bts), ptr(btc),
            ptr(starts_f), ptr(counts_f), ptr(uff), ct.byref(n)))
        uff = uff[:n.value].contiguous()
        targets = make_obj_array([_gather(actx, _dev(actx, t), uff) for t in tree.targets])
        result = FilteredTargetListsInTreeOrder(
            nfiltered_targets=int(n.value), box_target_starts=starts_f,
            box_target_counts_nonchild=counts_f, targets=targets,
            unfiltered_from_filtered_target_indices=uff)
        return actx.freeze(result)

# }}}
