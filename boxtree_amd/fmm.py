"""``drive_fmm``: the top-level FMM driver with the reference's stages and
wrangler interface (boxtree/fmm.py:342-532), for a single rank."""

from __future__ import annotations

import logging

logger = logging.getLogger(__name__)

__all__ = ["drive_fmm"]


def drive_fmm(actx, wrangler, src_weight_vecs, *, global_src_idx_all_ranks=None,
              global_tgt_idx_all_ranks=None):
    """Runs the eight FMM stages on *wrangler* (an object with the methods of
    ``ExpansionWranglerInterface``, boxtree/fmm.py:80-339) and returns the
    potentials in user target order.

    :arg src_weight_vecs: a sequence of source weight arrays in user source order,
        passed to the wrangler unmodified apart from the reordering.
    """
    traversal = wrangler.traversal

    src_weight_vecs = [wrangler.reorder_sources(weight) for weight in src_weight_vecs]
    src_weight_vecs = wrangler.distribute_source_weights(
        actx, src_weight_vecs, global_src_idx_all_ranks)

    # Step 2.1: multipoles of the source boxes
    mpole_exps = wrangler.form_multipoles(
        actx, traversal.level_start_source_box_nrs, traversal.source_boxes, src_weight_vecs)
    # Step 2.2: upward pass
    mpole_exps = wrangler.coarsen_multipoles(
        actx, traversal.level_start_source_parent_box_nrs, traversal.source_parent_boxes,
        mpole_exps)
    wrangler.communicate_mpoles(actx, mpole_exps)

    # Stage 3: list 1, directly
    potentials = wrangler.eval_direct(
        actx, traversal.target_boxes, traversal.neighbor_source_boxes_starts,
        traversal.neighbor_source_boxes_lists, src_weight_vecs)

    # Stage 4: list 2, multipole to local
    local_exps = wrangler.multipole_to_local(
        actx, traversal.level_start_target_or_target_parent_box_nrs,
        traversal.target_or_target_parent_boxes, traversal.from_sep_siblings_starts,
        traversal.from_sep_siblings_lists, mpole_exps)

    # Stage 5: list 3, multipoles evaluated at the targets
    potentials = potentials + wrangler.eval_multipoles(
        actx, traversal.target_boxes_sep_smaller_by_source_level,
        traversal.from_sep_smaller_by_level, mpole_exps)
    if traversal.from_sep_close_smaller_starts is not None:
        potentials = potentials + wrangler.eval_direct(
            actx, traversal.target_boxes, traversal.from_sep_close_smaller_starts,
            traversal.from_sep_close_smaller_lists, src_weight_vecs)

    # Stage 6: list 4, sources to locals
    local_exps = local_exps + wrangler.form_locals(
        actx, traversal.level_start_target_or_target_parent_box_nrs,
        traversal.target_or_target_parent_boxes, traversal.from_sep_bigger_starts,
        traversal.from_sep_bigger_lists, src_weight_vecs)
    if traversal.from_sep_close_bigger_starts is not None:
        potentials = potentials + wrangler.eval_direct(
            actx, traversal.target_boxes, traversal.from_sep_close_bigger_starts,
            traversal.from_sep_close_bigger_lists, src_weight_vecs)

    # Stage 7: downward pass
    local_exps = wrangler.refine_locals(
        actx, traversal.level_start_target_or_target_parent_box_nrs,
        traversal.target_or_target_parent_boxes, local_exps)

    # Stage 8: locals evaluated at the targets
    potentials = potentials + wrangler.eval_locals(
        actx, traversal.level_start_target_box_nrs, traversal.target_boxes, local_exps)

    potentials = wrangler.gather_potential_results(actx, potentials, global_tgt_idx_all_ranks)
    result = wrangler.reorder_potentials(potentials)
    return wrangler.finalize_potentials(actx, result)
