"""A checksum of a (possibly sharded) tree that is LINEAR in the per-box particle counts.

``tree_checksum = sum_b counts_cumul[b] * w(global number of b)`` in wrapping int64
arithmetic, ``w(g) = (g * 2654435761 mod 2^32) | 1``.  A rank of a sharded build holds
the global tree restricted to the boxes that contain its particles: its deep boxes are
its own, the shared top boxes carry the rank's LOCAL counts.  Because the sum is linear,
adding the ranks' checksums (one all-reduce of an int64) gives the checksum of the
single-GPU tree whatever the number of ranks -- which is what
``tests/golden/c5_global_counts.json`` records for BASELINE configs[4] and what
``bench.py --gpus N`` compares itself with.
"""

from __future__ import annotations

_MULT = 2654435761


def box_weights(torch, global_box_ids):
    g = global_box_ids.to(torch.int64)
    return ((g * _MULT) & 0xFFFFFFFF) | 1


def tree_checksum(torch, global_box_ids, counts_cumul):
    """int (two's-complement int64) checksum of the boxes given; device tensors in."""
    w = box_weights(torch, global_box_ids)
    return int((counts_cumul.to(torch.int64) * w).sum().item())


def wrap_int64(v):
    """Python int -> the int64 it wraps to (sums of per-rank checksums)."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def particle_order_checksum(torch, global_user_ids_in_tree_order, first_position=0):
    """Checksum of ``user_source_ids`` (tree.py:426-431) of a tree, or of a rank's slice of it:
    ``sum_p (p + 1) * id(p)`` in wrapping int64 arithmetic, ``p`` the GLOBAL tree position
    (*first_position* = ``numbering["source_offset"]`` of a sharded build) and ``id(p)`` the
    global user id of the source there -- for a sharded build the library's own
    (``ParticleRoute.global_user_source_ids``).  The ranks' values add up (wrapping) to the
    single-GPU tree's: equal sums mean every particle sits at its position of the global
    order."""
    ids = global_user_ids_in_tree_order.to(torch.int64)
    pos = torch.arange(1, len(ids) + 1, device=ids.device, dtype=torch.int64) + int(first_position)
    return int((pos * ids).sum().item())
