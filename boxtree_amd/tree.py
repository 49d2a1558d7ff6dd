"""Output containers of the tree build: same names/layouts as boxtree/tree.py."""

from __future__ import annotations

import dataclasses
from dataclasses import dataclass
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


@dataclass(frozen=True)
class Tree(_Container):
    """Particles sorted into a hierarchy of boxes; field-for-field the
    reference's :class:`boxtree.Tree` (boxtree/tree.py:298-686)."""
    sources_are_targets: bool
    sources_have_extent: bool
    targets_have_extent: bool

    particle_id_dtype: Any
    box_id_dtype: Any
    coord_dtype: Any
    box_level_dtype: Any

    bounding_box: Any            # (bbox_min, bbox_max) numpy vectors
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
