"""Distributed FMM evaluation on local trees: what a wrangler needs on top of the
single-rank interface (boxtree/distributed/calculation.py:49-413).

One process per GPU; *comm* is ``torch.distributed`` (backend ``nccl`` = RCCL) or
an object with the same collective calls.  Everything that moves between ranks is
a variable-size message between two GPUs, expressed as ``all_to_all_single`` with
explicit split sizes: RCCL lowers that to grouped point-to-point sends over the
direct xGMI links, empty pairs cost nothing, and messages are cut below
``A2A_MESSAGE_LIMIT_BYTES`` (see boxtree_amd/distributed/__init__.py).
"""

from __future__ import annotations

import ctypes as ct

import numpy as np

from boxtree_amd.array_context import ptr
from boxtree_amd.distributed import all_to_all_chunked
from boxtree_amd.distributed.partition import _call

__all__ = ["DistributedExpansionWranglerMixin", "DistributedConstantOneExpansionWrangler",
           "reduce_scatter_stage",
           "reduce_scatter_num_stages", "gather_to_root", "scatter_from_root"]


# {{{ the stage structure of the multipole exchange

def reduce_scatter_stage(rank, left, right):
    """One stage of the sparse all-reduce of Lashuk et al. (CACM 55(5), 2012),
    Algorithm 3, as the reference runs it (``AllReduceCommPattern``,
    boxtree/tools.py:756-855; driven from calculation.py:264-413).

    The ranks ``[left, right)`` still exchanging with each other are cut at
    ``mid = (left + right) // 2``; rank ``left + k`` pairs with rank ``mid + k``.
    The upper half may have one more rank than the lower: that last rank sends to
    the last lower rank and receives from no one this stage (it is served by its
    own half in the following stages).

    :returns: ``(sinks, sources, (users_lo, users_hi), (next_left, next_right))``:
        the ranks this rank sends to, the ranks it hears from, the range of user
        ranks whose boxes it sends, and its range for the next stage.
    """
    assert left <= rank < right and right - left > 1
    mid = (left + right) // 2
    nlower = mid - left
    if rank < mid:
        k = rank - left
        sources = [mid + k]
        if k == nlower - 1 and mid + k + 2 == right:
            sources.append(mid + k + 1)
        return [mid + k], sources, (mid, right), (left, mid)
    k = rank - mid
    if k >= nlower:
        return [mid - 1], [], (left, mid), (mid, right)
    return [left + k], [left + k], (left, mid), (mid, right)


def reduce_scatter_num_stages(size):
    """Stages until every rank's range has shrunk to itself (the deepest chain)."""
    nstages = 0
    ranges = {(0, size)}
    while any(r - l > 1 for l, r in ranges):
        nxt = set()
        for l, r in ranges:
            if r - l > 1:
                mid = (l + r) // 2
                nxt.update({(l, mid), (mid, r)})
        ranges = nxt
        nstages += 1
    return nstages

# }}}


# {{{ root <-> ranks movement of per-particle arrays

def _counts(actx, comm, n):
    """int64 [size] of every rank's *n*."""
    size = comm.get_world_size()
    mine = actx.torch.tensor([int(n)], dtype=actx.torch.int64, device=actx.device)
    if size == 1:
        return [int(n)]
    out = [actx.torch.empty_like(mine) for _ in range(size)]
    comm.all_gather(out, mine)
    return [int(t.item()) for t in out]


def gather_to_root(actx, comm, t):
    """Rank 0 gets the list of every rank's 1-D tensor *t*, the others ``None``."""
    size, rank = comm.get_world_size(), comm.get_rank()
    t = t.contiguous()
    counts = _counts(actx, comm, t.shape[0])
    if size == 1:
        return [t]
    s_split = [0] * size
    s_split[0] = int(t.shape[0])
    r_split = counts if rank == 0 else [0] * size
    recv = actx.torch.empty(sum(r_split), dtype=t.dtype, device=actx.device)
    all_to_all_chunked(comm, recv, t, r_split, s_split)
    if rank != 0:
        return None
    off = np.concatenate([[0], np.cumsum(counts)])
    return [recv[off[r]:off[r + 1]] for r in range(size)]


def scatter_from_root(actx, comm, chunks, n_mine, dtype):
    """Inverse of :func:`gather_to_root`: rank r receives ``chunks[r]`` (a list that
    only rank 0 needs to supply) as a tensor of *n_mine* elements."""
    size, rank = comm.get_world_size(), comm.get_rank()
    if size == 1:
        return chunks[0]
    if rank == 0:
        send = actx.torch.cat([c.to(dtype) for c in chunks])
        s_split = [int(c.shape[0]) for c in chunks]
    else:
        send = actx.torch.empty(0, dtype=dtype, device=actx.device)
        s_split = [0] * size
    r_split = [0] * size
    r_split[0] = int(n_mine)
    recv = actx.torch.empty(int(n_mine), dtype=dtype, device=actx.device)
    all_to_all_chunked(comm, recv, send, r_split, s_split)
    return recv

# }}}


class DistributedExpansionWranglerMixin:
    """Adds the three distributed steps of :func:`boxtree_amd.fmm.drive_fmm` to a
    wrangler built on a *local* traversal (calculation.py:49-413).

    Expected attributes: ``comm``, ``traversal`` (local), ``global_traversal``,
    ``communicate_mpoles_via_allreduce``.  ``slice_mpoles`` / ``update_mpoles`` may
    be overridden for expansion storage that is not one row per box.
    """
    communicate_mpoles_via_allreduce = False

    @property
    def mpi_rank(self):
        return self.comm.get_rank()

    @property
    def mpi_size(self):
        return self.comm.get_world_size()

    @property
    def is_mpi_root(self):
        return self.mpi_rank == 0

    # -- weights out, potentials back (calculation.py:80-141) ----------------------
    def distribute_source_weights(self, actx, src_weight_vecs, src_idx_all_ranks):
        """Rank r receives, for each weight vector, the entries of its local sources.
        *src_weight_vecs* (global tree order) and *src_idx_all_ranks* matter on the
        root only."""
        n_mine = int(self.traversal.tree.nsources)
        nvecs = _counts(actx, self.comm, len(src_weight_vecs) if self.is_mpi_root else 0)[0]
        dtype = actx.torch.float64
        out = []
        for i in range(nvecs):
            chunks = None
            if self.is_mpi_root:
                w = src_weight_vecs[i]
                dtype = w.dtype
                chunks = [w[idx.long()] for idx in src_idx_all_ranks]
            out.append(scatter_from_root(actx, self.comm, chunks, n_mine, dtype))
        return out

    def gather_potential_results(self, actx, potentials, tgt_idx_all_ranks):
        """Root: potentials of all targets in global tree order; others: ``None``."""
        parts = gather_to_root(actx, self.comm, potentials)
        if not self.is_mpi_root:
            return None
        out = actx.torch.empty(int(self.global_traversal.tree.ntargets),
                               dtype=potentials.dtype, device=actx.device)
        for part, idx in zip(parts, tgt_idx_all_ranks):
            out[idx.long()] = part
        return out

    # -- multipole storage hooks (calculation.py:143-189) ----------------------------
    def slice_mpoles(self, actx, mpoles, box_list):
        return mpoles[box_list.long()].reshape(-1)

    def update_mpoles(self, actx, mpoles, mpole_updates, box_list):
        if int(box_list.shape[0]) == 0:
            return
        mpoles.index_add_(0, box_list.long(),
                          mpole_updates.reshape((int(box_list.shape[0]),) + mpoles.shape[1:]))

    def _boxes_used_by(self, actx, contributing, subrange):
        tree = self.traversal.tree
        nboxes = int(tree.nboxes)
        boxes = actx.empty(nboxes, np.int32)
        n = ct.c_int64(0)
        actx.sync_in()
        _call(actx, actx.lib.bt_boxes_used_by_ranks(
            actx.handle, nboxes, ptr(contributing), int(subrange[0]), int(subrange[1]),
            ptr(tree.box_to_user_rank_starts), ptr(tree.box_to_user_rank_lists), ptr(boxes),
            ct.byref(n)))
        return boxes[:int(n.value)]

    def communicate_mpoles(self, actx, mpole_exps, return_stats=False):
        """Completes, on every rank, the multipole expansions its lists read
        (calculation.py:264-413): a recursive-halving reduce in which a rank passes
        on only the boxes whose users lie in the partner's half.  Updates
        *mpole_exps* in place."""
        comm = self.comm
        size, rank = self.mpi_size, self.mpi_rank
        if size == 1:
            return {"bytes_sent_by_stage": [], "bytes_recvd_by_stage": []} if return_stats else None
        if self.communicate_mpoles_via_allreduce:
            comm.all_reduce(mpole_exps, op=comm.ReduceOp.SUM)
            return None

        torch = actx.torch
        tree = self.traversal.tree
        # boxes this rank holds a (partial) expansion of: its own boxes and their
        # ancestors; boxes it receives parts of join the set (calculation.py:291-312)
        contributing = tree.ancestor_mask.clone()
        contributing[tree.responsible_boxes_list.long()] = 1
        row = 1
        for s in mpole_exps.shape[1:]:
            row *= int(s)

        stats = {"bytes_sent_by_stage": [], "bytes_recvd_by_stage": []}
        left, right = 0, size
        for _stage in range(reduce_scatter_num_stages(size)):
            n_send = [0] * size
            send_boxes = torch.empty(0, dtype=torch.int32, device=actx.device)
            sources = []
            if right - left > 1:
                sinks, sources, users, (left, right) = reduce_scatter_stage(rank, left, right)
                boxes = self._boxes_used_by(actx, contributing, users)
                for sink in sinks:
                    n_send[sink] = int(boxes.shape[0])
                send_boxes = boxes
            # message sizes, then box numbers, then expansions -- all ranks take part
            # in every stage so that the collective calls line up
            t_send = torch.tensor(n_send, dtype=torch.int64, device=actx.device)
            t_recv = torch.empty_like(t_send)
            comm.all_to_all_single(t_recv, t_send)
            n_recv = [int(v) for v in t_recv.tolist()]
            assert all(n_recv[r] == 0 for r in range(size) if r not in sources)

            # a rank has a single sink, so the send buffer is one segment
            recv_boxes = torch.empty(sum(n_recv), dtype=torch.int32, device=actx.device)
            all_to_all_chunked(comm, recv_boxes, send_boxes[:sum(n_send)], n_recv, n_send)
            send_vals = (self.slice_mpoles(actx, mpole_exps, send_boxes) if sum(n_send)
                         else torch.empty(0, dtype=mpole_exps.dtype, device=actx.device))
            recv_vals = torch.empty(sum(n_recv) * row, dtype=mpole_exps.dtype,
                                    device=actx.device)
            all_to_all_chunked(comm, recv_vals, send_vals, [n * row for n in n_recv],
                               [n * row for n in n_send])
            off = 0
            for r in range(size):
                if n_recv[r]:
                    blist = recv_boxes[off:off + n_recv[r]]
                    self.update_mpoles(actx, mpole_exps, recv_vals[off * row:(off + n_recv[r]) * row],
                                       blist)
                    contributing[blist.long()] = 1
                    off += n_recv[r]
            es = mpole_exps.element_size()
            stats["bytes_sent_by_stage"].append(sum(n_send) * (row * es + 4))
            stats["bytes_recvd_by_stage"].append(sum(n_recv) * (row * es + 4))
        return stats if return_stats else None


def _constant_one_base():
    from boxtree_amd.constant_one import ConstantOneExpansionWrangler
    return ConstantOneExpansionWrangler


class DistributedConstantOneExpansionWrangler(DistributedExpansionWranglerMixin,
                                              _constant_one_base()):
    """The constant-one wrangler on a local traversal: with unit weights rank 0 must
    get ``nsources`` at every target, which exercises the partition, the local trees
    and the multipole exchange end to end (the reference builds this class inside
    test/test_distributed.py:182-217; its shipped concrete wrangler wraps FMMLib,
    calculation.py:416-453)."""

    def __init__(self, comm, tree_indep, local_traversal, global_traversal,
                 communicate_mpoles_via_allreduce=False):
        super().__init__(tree_indep, local_traversal)
        self.comm = comm
        self.global_traversal = global_traversal
        self.communicate_mpoles_via_allreduce = communicate_mpoles_via_allreduce
        self.level_orders = np.ones(int(local_traversal.tree.nlevels), dtype=np.int32)

    def reorder_sources(self, source_array):
        if self.is_mpi_root:
            return source_array[self.global_traversal.tree.user_source_ids.long()]
        return None

    def reorder_potentials(self, potentials):
        if self.is_mpi_root:
            return potentials[self.global_traversal.tree.sorted_target_ids.long()]
        return None

    def finalize_potentials(self, actx, potentials):
        return potentials if self.is_mpi_root else None
