"""The call surface of an FMM evaluation on this package's lists: the abstract
wrangler interface downstream codes subclass (sumpy, pytential; boxtree/fmm.py:51-339) and
:func:`drive_fmm` (boxtree/fmm.py:342-532), written here as a table of stages: which wrangler
method consumes which arrays of the traversal, and where its result goes."""

from __future__ import annotations

import logging
from abc import ABC, abstractmethod

logger = logging.getLogger(__name__)

__all__ = ["TreeIndependentDataForWrangler", "ExpansionWranglerInterface", "drive_fmm", "FMM_STAGES"]


class TreeIndependentDataForWrangler:
    """Base class for what a wrangler needs that depends on the kernel only: compiled
    translation operators, expansion orders, precomputed tables.  One instance serves every
    tree and traversal, so it must not keep a :class:`~boxtree_amd.tree.Tree` alive
    (boxtree/fmm.py:51-66)."""


class ExpansionWranglerInterface(ABC):
    """What :func:`drive_fmm` calls (boxtree/fmm.py:69-339).  A wrangler belongs to one
    traversal -- ``tree_indep`` holds the reusable, kernel-specific part -- and supplies:

    * the two reorderings between the caller's particle order and tree order,
    * views of the multipole / local expansion storage of one level,
    * one method per translation; each takes the arrays of the traversal it iterates over
      and returns potentials (tree target order) or expansion storage,
    * ``finalize_potentials``, applied to the result in the caller's order,
    * three hooks that are identities on one rank and communicate in a distributed run.

    Attributes: ``tree_indep`` (:class:`TreeIndependentDataForWrangler`), ``traversal``
    (:class:`~boxtree_amd.traversal.FMMTraversalInfo`), ``tree`` (its tree)."""

    def __init__(self, tree_indep, traversal):
        self.tree_indep = tree_indep
        self.traversal = traversal

    @property
    def tree(self):
        return self.traversal.tree

    # ---- particle order -------------------------------------------------------------------------

    @abstractmethod
    def reorder_sources(self, source_array):
        """*source_array* in the caller's source order -> tree source order
        (``tree.user_source_ids``)."""

    @abstractmethod
    def reorder_potentials(self, potentials):
        """*potentials* (one array, or an object array of them) in tree target order -> the
        caller's target order (``tree.sorted_target_ids``)."""

    # ---- expansion storage ----------------------------------------------------------------------

    @abstractmethod
    def multipole_expansions_view(self, mpole_exps, level):
        """``(first box of the level, the level's part of mpole_exps [boxes of the level, coefficients])``"""

    @abstractmethod
    def local_expansions_view(self, local_exps, level):
        """Like :meth:`multipole_expansions_view`, for local expansions."""

    # ---- translations ---------------------------------------------------------------------------

    @abstractmethod
    def form_multipoles(self, actx, level_start_source_box_nrs, source_boxes, src_weight_vecs):
        """Sources of every box in *source_boxes* -> its multipole expansion.  Returns the
        multipole storage (every box, zero where nothing was formed)."""

    @abstractmethod
    def coarsen_multipoles(self, actx, level_start_source_parent_box_nrs, source_parent_boxes, mpoles):
        """Upward pass: for every box of *source_parent_boxes*, deepest level first, the
        children's multipoles are translated to the box and added.  Returns *mpoles*."""

    @abstractmethod
    def eval_direct(self, actx, target_boxes, neighbor_sources_starts, neighbor_sources_lists,
                    src_weight_vecs):
        """Sources of the listed boxes evaluated at the targets of each box of *target_boxes*
        (the CSR list is indexed like *target_boxes*).  Returns potentials in tree target
        order."""

    @abstractmethod
    def multipole_to_local(self, actx, level_start_target_or_target_parent_box_nrs,
                           target_or_target_parent_boxes, starts, lists, mpole_exps):
        """Multipoles of the listed boxes -> local expansions of each box of
        *target_or_target_parent_boxes* (the CSR list is indexed like it).  Returns new local
        storage."""

    @abstractmethod
    def eval_multipoles(self, actx, target_boxes_by_source_level, from_sep_smaller_by_level, mpole_exps):
        """Per source level: the multipoles listed for ``target_boxes_by_source_level[level]``
        evaluated at those boxes' targets.  Returns potentials in tree target order."""

    @abstractmethod
    def form_locals(self, actx, level_start_target_or_target_parent_box_nrs,
                    target_or_target_parent_boxes, starts, lists, src_weight_vecs):
        """Sources of the listed boxes -> local expansions of each box of
        *target_or_target_parent_boxes*.  Returns new local storage."""

    @abstractmethod
    def refine_locals(self, actx, level_start_target_or_target_parent_box_nrs,
                      target_or_target_parent_boxes, local_exps):
        """Downward pass: every box of levels 1.. receives its parent's local expansion.
        Returns *local_exps*."""

    @abstractmethod
    def eval_locals(self, actx, level_start_target_box_nrs, target_boxes, local_exps):
        """Local expansion of every box of *target_boxes* evaluated at its targets.  Returns
        potentials in tree target order."""

    @abstractmethod
    def finalize_potentials(self, actx, potentials):
        """Last word on the result, in the caller's target order (scaling, type changes)."""

    # ---- hooks of a distributed run (identities on one rank) ------------------------------------

    def distribute_source_weights(self, actx, src_weight_vecs, src_idx_all_ranks):
        """Tree-ordered source weights of the global tree -> the weights of this rank's local
        tree (boxtree/fmm.py:281-301).  One rank: unchanged."""
        return src_weight_vecs

    def gather_potential_results(self, actx, potentials, tgt_idx_all_ranks):
        """The ranks' potentials -> potentials of the global tree on the root rank
        (boxtree/fmm.py:303-320).  One rank: unchanged."""
        return potentials

    def communicate_mpoles(self, actx, mpole_exps, return_stats=False):
        """Multipoles of boxes other ranks are responsible for are summed into *mpole_exps*
        (boxtree/fmm.py:322-338).  One rank: nothing to do."""


# The FMM as data: (stage name, wrangler method, attributes of the traversal handed over in order,
# what else the method reads, where the result goes).  "?" marks a stage that exists only when
# the traversal has that list (close lists: trees with extents, traversal.py:842-868, 1003).
# Inputs: "w" source weights in tree order, "m" multipole storage, "l" local storage.
# Outputs: "m" / "l" replace the storage, "+l" adds to it, "+p" adds to the potentials.
FMM_STAGES = (
    ("form multipoles", "form_multipoles",
     ("level_start_source_box_nrs", "source_boxes"), "w", "m"),
    ("propagate multipoles upward", "coarsen_multipoles",
     ("level_start_source_parent_box_nrs", "source_parent_boxes"), "m", "m"),
    ("communicate multipoles", "communicate_mpoles", (), "m", None),
    ("direct evaluation from neighbor source boxes (list 1)", "eval_direct",
     ("target_boxes", "neighbor_source_boxes_starts", "neighbor_source_boxes_lists"), "w", "+p"),
    ("translate separated siblings (list 2) to local", "multipole_to_local",
     ("level_start_target_or_target_parent_box_nrs", "target_or_target_parent_boxes",
      "from_sep_siblings_starts", "from_sep_siblings_lists"), "m", "+l"),
    ("evaluate separated smaller multipoles (list 3) at the targets", "eval_multipoles",
     ("target_boxes_sep_smaller_by_source_level", "from_sep_smaller_by_level"), "m", "+p"),
    ("direct evaluation of list 3 close", "eval_direct",
     ("target_boxes", "?from_sep_close_smaller_starts", "from_sep_close_smaller_lists"), "w", "+p"),
    ("form locals for separated bigger source boxes (list 4)", "form_locals",
     ("level_start_target_or_target_parent_box_nrs", "target_or_target_parent_boxes",
      "from_sep_bigger_starts", "from_sep_bigger_lists"), "w", "+l"),
    ("direct evaluation of list 4 close", "eval_direct",
     ("target_boxes", "?from_sep_close_bigger_starts", "from_sep_close_bigger_lists"), "w", "+p"),
    ("propagate local expansions downward", "refine_locals",
     ("level_start_target_or_target_parent_box_nrs", "target_or_target_parent_boxes"), "l", "l"),
    ("evaluate locals", "eval_locals",
     ("level_start_target_box_nrs", "target_boxes"), "l", "+p"),
)


def drive_fmm(actx, wrangler, src_weight_vecs, *, global_src_idx_all_ranks=None,
              global_tgt_idx_all_ranks=None):
    """Evaluates the FMM of *wrangler* (an :class:`ExpansionWranglerInterface`, or any object
    with its methods) for *src_weight_vecs* -- a sequence of source weight arrays in the
    caller's source order -- and returns the potentials in the caller's target order.  Same
    stages, in the same order, as boxtree/fmm.py:380-532; *global_src_idx_all_ranks* /
    *global_tgt_idx_all_ranks* go to the wrangler's two distribution hooks untouched."""
    trav = wrangler.traversal
    state = {
        "w": wrangler.distribute_source_weights(
            actx, [wrangler.reorder_sources(w) for w in src_weight_vecs], global_src_idx_all_ranks),
        "m": None, "l": None, "p": None,
    }
    for name, method, fields, reads, writes in FMM_STAGES:
        args, present = [], True
        for f in fields:
            optional = f.startswith("?")
            value = getattr(trav, f.lstrip("?"))
            if optional and value is None:
                present = False
                break
            args.append(value)
        if not present:
            continue
        logger.debug("fmm stage: %s", name)
        result = getattr(wrangler, method)(actx, *args, state[reads])
        if writes is None:
            continue
        key = writes.lstrip("+")
        if writes.startswith("+") and state[key] is not None:
            state[key] = state[key] + result
        else:
            state[key] = result
    potentials = wrangler.gather_potential_results(actx, state["p"], global_tgt_idx_all_ranks)
    return wrangler.finalize_potentials(actx, wrangler.reorder_potentials(potentials))
