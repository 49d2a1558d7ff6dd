"""``ConstantOneExpansionWrangler``: the 'analytical routines' of a Green's
function that is 1 everywhere (boxtree/constant_one.py:49-237), on the device.
With unit charges every target must come out with the number of sources -- the
reference's completeness test of the interaction lists (test/test_fmm.py:141-391),
here runnable at full problem size."""

from __future__ import annotations

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import ptr
from boxtree_amd.fmm import ExpansionWranglerInterface, TreeIndependentDataForWrangler

__all__ = ["ConstantOneExpansionWrangler", "ConstantOneTreeIndependentDataForWrangler"]


class ConstantOneTreeIndependentDataForWrangler(TreeIndependentDataForWrangler):
    """Nothing to precompute for a constant kernel (constant_one.py:43-46)."""


class ConstantOneExpansionWrangler(ExpansionWranglerInterface):
    """An 'expansion' is one float64 per box; translations are sums.

    :arg traversal: a device :class:`~boxtree_amd.traversal.FMMTraversalInfo`.
    """

    def __init__(self, tree_indep, traversal):
        super().__init__(tree_indep, traversal)
        self._box_weight_cache = None

    # -- helpers ------------------------------------------------------------------
    def _call(self, actx, code):
        if code == _lib.BT_ERR_INVALID:
            raise ValueError(actx.lib.bt_last_error_string().decode())
        _lib.check(code)

    def _source_box_weights(self, actx, src_weights):
        """[nboxes] sum of the weights of each box's own sources."""
        # keyed on the tensor object and its version counter: an address alone is
        # recycled by the caching allocator for the next call's weights
        c = self._box_weight_cache
        if c is not None and c[0] is src_weights and c[1] == src_weights._version:
            return c[2]
        tree = self.tree
        out = actx.zeros(int(tree.nboxes), np.float64)
        actx.sync_in()
        self._call(actx, actx.lib.bt_fmm_box_particle_sums(
            actx.handle, int(tree.nboxes), None, ptr(tree.box_source_starts),
            ptr(tree.box_source_counts_nonchild), ptr(src_weights), ptr(out), 0))
        self._box_weight_cache = (src_weights, src_weights._version, out)
        return out

    def _csr_rows(self, actx, starts, lists, box_values):
        nrows = int(starts.shape[0]) - 1
        out = actx.zeros(nrows, np.float64)
        actx.sync_in()
        self._call(actx, actx.lib.bt_fmm_csr_sum(
            actx.handle, nrows, ptr(starts), ptr(lists.contiguous()), ptr(box_values), None,
            ptr(out), 0))
        return out

    def _csr_scatter(self, actx, row_boxes, starts, lists, box_values, dst):
        nrows = int(starts.shape[0]) - 1
        actx.sync_in()
        self._call(actx, actx.lib.bt_fmm_csr_sum(
            actx.handle, nrows, ptr(starts), ptr(lists.contiguous()), ptr(box_values),
            ptr(row_boxes), ptr(dst), 1))

    def _to_targets(self, actx, row_boxes, row_values, box_values, pot, accumulate):
        tree = self.tree
        actx.sync_in()
        self._call(actx, actx.lib.bt_fmm_box_to_particles(
            actx.handle, int(row_boxes.shape[0]), ptr(row_boxes), ptr(tree.box_target_starts),
            ptr(tree.box_target_counts_nonchild), ptr(row_values), ptr(box_values), ptr(pot),
            int(accumulate)))

    # -- storage ------------------------------------------------------------------
    def multipole_expansion_zeros(self, actx):
        return actx.zeros(int(self.tree.nboxes), np.float64)

    local_expansion_zeros = multipole_expansion_zeros

    def output_zeros(self, actx):
        return actx.zeros(int(self.tree.ntargets), np.float64)

    def reorder_sources(self, source_array):
        return source_array[self.tree.user_source_ids.long()]          # constant_one.py:74-75

    def reorder_potentials(self, potentials):
        return potentials[self.tree.sorted_target_ids.long()]          # constant_one.py:77-78

    # (one float per box: there is nothing to slice per level; upstream leaves the two views
    # unimplemented as well, constant_one.py:80-86)
    def multipole_expansions_view(self, mpole_exps, level):
        raise NotImplementedError

    def local_expansions_view(self, local_exps, level):
        raise NotImplementedError

    def finalize_potentials(self, actx, potentials):
        return potentials

    # -- the stages ---------------------------------------------------------------------
    def form_multipoles(self, actx, level_start_source_box_nrs, source_boxes, src_weight_vecs):
        src_weights, = src_weight_vecs
        tree = self.tree
        mpoles = self.multipole_expansion_zeros(actx)
        actx.sync_in()
        self._call(actx, actx.lib.bt_fmm_box_particle_sums(
            actx.handle, int(source_boxes.shape[0]), ptr(source_boxes),
            ptr(tree.box_source_starts), ptr(tree.box_source_counts_nonchild),
            ptr(src_weights), ptr(mpoles), 1))
        return mpoles

    def coarsen_multipoles(self, actx, level_start_source_parent_box_nrs, source_parent_boxes,
                           mpoles):
        tree = self.tree
        lev = actx.to_numpy(level_start_source_parent_box_nrs)
        nchildren = int(tree.box_child_ids.shape[0])
        # constant_one.py:109-121: source levels nlevels-1 .. 3
        for source_level in range(int(tree.nlevels) - 1, 2, -1):
            target_level = source_level - 1
            start, stop = int(lev[target_level]), int(lev[target_level + 1])
            if stop > start:
                actx.sync_in()
                self._call(actx, actx.lib.bt_fmm_tree_sweep(
                    actx.handle, stop - start, ptr(source_parent_boxes[start:stop].contiguous()),
                    ptr(tree.box_child_ids), int(tree.aligned_nboxes), nchildren, None,
                    ptr(mpoles)))
        return mpoles

    def eval_direct(self, actx, target_boxes, neighbor_sources_starts, neighbor_sources_lists,
                    src_weight_vecs):
        src_weights, = src_weight_vecs
        pot = self.output_zeros(actx)
        rows = self._csr_rows(actx, neighbor_sources_starts, neighbor_sources_lists,
                              self._source_box_weights(actx, src_weights))
        self._to_targets(actx, target_boxes, rows, None, pot, accumulate=False)   # :144
        return pot

    def multipole_to_local(self, actx, level_start_target_or_target_parent_box_nrs,
                           target_or_target_parent_boxes, starts, lists, mpole_exps):
        local_exps = self.local_expansion_zeros(actx)
        self._csr_scatter(actx, target_or_target_parent_boxes, starts, lists, mpole_exps,
                          local_exps)
        return local_exps

    def eval_multipoles(self, actx, target_boxes_by_source_level,
                        from_sep_smaller_nonsiblings_by_level, mpole_exps):
        pot = self.output_zeros(actx)
        for level, ssn in enumerate(from_sep_smaller_nonsiblings_by_level):
            tboxes = target_boxes_by_source_level[level]
            if int(tboxes.shape[0]) == 0:
                continue
            rows = self._csr_rows(actx, ssn.starts, ssn.lists, mpole_exps)
            self._to_targets(actx, tboxes, rows, None, pot, accumulate=True)
        return pot

    def form_locals(self, actx, level_start_target_or_target_parent_box_nrs,
                    target_or_target_parent_boxes, starts, lists, src_weight_vecs):
        src_weights, = src_weight_vecs
        local_exps = self.local_expansion_zeros(actx)
        self._csr_scatter(actx, target_or_target_parent_boxes, starts, lists,
                          self._source_box_weights(actx, src_weights), local_exps)
        return local_exps

    def refine_locals(self, actx, level_start_target_or_target_parent_box_nrs,
                      target_or_target_parent_boxes, local_exps):
        tree = self.tree
        lev = actx.to_numpy(level_start_target_or_target_parent_box_nrs)
        for target_lev in range(1, int(tree.nlevels)):                 # constant_one.py:217-221
            start, stop = int(lev[target_lev]), int(lev[target_lev + 1])
            if stop > start:
                actx.sync_in()
                self._call(actx, actx.lib.bt_fmm_tree_sweep(
                    actx.handle, stop - start,
                    ptr(target_or_target_parent_boxes[start:stop].contiguous()), None, 0, 0,
                    ptr(tree.box_parent_ids), ptr(local_exps)))
        return local_exps

    def eval_locals(self, actx, level_start_target_box_nrs, target_boxes, local_exps):
        pot = self.output_zeros(actx)
        self._to_targets(actx, target_boxes, None, local_exps, pot, accumulate=True)
        return pot
