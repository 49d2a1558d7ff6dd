"""Interaction lists of a local tree (boxtree/distributed/local_traversal.py:34-62)."""

from __future__ import annotations

__all__ = ["generate_local_travs"]


def generate_local_travs(actx, local_tree, traversal_builder, merge_close_lists=False):
    """Runs *traversal_builder* on *local_tree* with the source boxes restricted to
    the rank's own boxes and the source parent boxes to their ancestors: a source
    sits in the local trees of several ranks (everyone whose list 1 reaches it), but
    only its owner may put it into a multipole expansion."""
    local_trav, _ = traversal_builder(
        actx, local_tree,
        source_boxes_mask=local_tree.responsible_boxes_mask,
        source_parent_boxes_mask=local_tree.ancestor_mask)
    if merge_close_lists and local_tree.targets_have_extent:
        local_trav = local_trav.merge_close_lists(actx)
    return local_trav
