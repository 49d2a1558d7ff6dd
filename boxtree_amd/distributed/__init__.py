"""Multi-GPU sharding of the tree build (one process per GPU, RCCL over xGMI).

The reference never builds the tree in parallel (root builds the global tree and
broadcasts it, boxtree/distributed/__init__.py:183-199); the build itself shards
naturally on disjoint point chunks.  This module is the
exchange step that precedes the per-rank single-GPU pipeline:

1. local bounding box -> ``all_reduce(MIN/MAX)`` (2*d doubles): every rank
   derives the identical root box with the reference's host arithmetic
   (tree_build.py:464-476);
2. level-``k`` Morton cell of every particle (same float expression as the key
   kernel, tbk:374-376), local histogram of the ``2^(d*k)`` cells ->
   ``all_reduce(SUM)``;
3. contiguous Morton ranges of cells are assigned to ranks, balanced by particle
   count (identical computation on every rank, no further communication);
4. ``all_to_all_single`` of per-destination counts, then of each coordinate
   array (and radii / global ids) -- the only bandwidth-relevant collective:
   about (world-1)/world of the local data leaves every GPU, spread over all
   xGMI links;
5. every rank runs ``TreeBuilder`` on what it owns with ``bbox=`` the global root
   box, so its boxes are exactly the global tree's boxes over its cell range
   (a level->=k box lies inside one cell; the shared top levels split on every
   rank because each owns whole, heavy cells).

Steps 1-5 use ``torch.distributed`` collectives (backend "nccl" = RCCL on GPUs,
"gloo" in the CPU tests).  On a GPU the per-particle work (cell index + histogram,
stable bucketing by owner, gather into send order) runs in the library's HIP
kernels (``bt_morton_cells``, ``bt_bucket_permutation`` = one onesweep digit
pass, ``bt_gather``); with ``actx=None`` (the gloo CPU tests of the plan/exchange
logic) the same routing is computed with torch tensor ops.  Global box numbering
(:func:`number_sharded_tree`) and the cross-boundary traversal
(:func:`build_local_essential_tree`, or :func:`gather_global_box_tree`) follow
further down.

The second half of this package is the reference's own distributed interface --
FMM *evaluation* on a replicated global tree: :mod:`.partition`,
:mod:`.local_tree`, :mod:`.local_traversal`, :mod:`.calculation` and
:class:`DistributedFMMRunner` below (boxtree/distributed/__init__.py:156-311).
"""

from __future__ import annotations

import os

import numpy as np

ROOT_EXTENT_STRETCH_FACTOR = 1e-4        # tree_build.py:101


def _local_minmax(actx, arrays, radii):
    import torch
    dims = len(arrays)
    if arrays[0].is_cuda and actx is not None:
        from boxtree_amd.bounding_box import AXIS_NAMES, BoundingBoxFinder
        bbox, _ = BoundingBoxFinder(actx)(actx, arrays, radii)
        mn = [float(bbox[f"min_{AXIS_NAMES[i]}"]) for i in range(dims)]
        mx = [float(bbox[f"max_{AXIS_NAMES[i]}"]) for i in range(dims)]
    else:
        big = float(np.finfo(np.float64).max)
        if len(arrays[0]) == 0:
            mn, mx = [big] * dims, [-big] * dims
        else:
            r = radii if radii is not None else 0
            mn = [float(torch.amin(a - r)) for a in arrays]
            mx = [float(torch.amax(a + r)) for a in arrays]
    return mn, mx


def global_root_box(actx, dist, particles, targets=None, source_radii=None,
                    target_radii=None):
    """Steps 1: identical (bbox_min, bbox_max, root_extent) on every rank."""
    import torch
    dims = len(particles)
    dev = particles[0].device
    mn, mx = _local_minmax(actx, particles, source_radii)
    if targets is not None:
        mn2, mx2 = _local_minmax(actx, targets, target_radii)
        mn = [min(a, b) for a, b in zip(mn, mn2)]
        mx = [max(a, b) for a, b in zip(mx, mx2)]
    tmn = torch.tensor(mn, dtype=torch.float64, device=dev)
    tmx = torch.tensor(mx, dtype=torch.float64, device=dev)
    dist.all_reduce(tmn, op=dist.ReduceOp.MIN)
    dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
    coord_dtype = np.dtype(str(particles[0].dtype).replace("torch.", ""))
    gmin = tmn.cpu().numpy().astype(coord_dtype)
    gmax = tmx.cpu().numpy().astype(coord_dtype)
    # tree_build.py:464-476 (numpy, coordinate dtype)
    root_extent = max(gmax[i] - gmin[i] for i in range(dims)) * (
        1 + ROOT_EXTENT_STRETCH_FACTOR)
    bbox_min = gmin.copy()
    bbox_max = bbox_min + root_extent
    return bbox_min, bbox_max, root_extent


def morton_cells(arrays, bbox_min, bbox_max, level):
    """Level-``level`` Morton cell index of every point (int64 tensor)."""
    import torch
    dims = len(arrays)
    cell = torch.zeros(len(arrays[0]), dtype=torch.int64, device=arrays[0].device)
    for ax in range(dims):
        gmin = float(bbox_min[ax])
        gext = float(bbox_max[ax]) - gmin
        v = (((arrays[ax] - gmin) / gext) * float(1 << level)).to(torch.int64)
        v = torch.clamp(v, 0, (1 << level) - 1)
        for b in range(level):
            bit = (v >> b) & 1
            cell |= bit << (dims * b + (dims - 1 - ax))     # x most significant
    return cell


def top_tree_plan(global_hist, dims, top_level, max_particles_in_box):
    """The top of the GLOBAL tree (levels 0..top_level), a pure function of the
    all-reduced level-``top_level`` cell histogram (kind="adaptive", point
    particles, unit weights: a box splits iff it holds more than
    *max_particles_in_box* particles, tree_build_kernels.py:577-591; empty boxes
    are pruned).  Cell and box indices are Morton paths, x most significant in
    every digit, i.e. the order of the reference's box numbering within a level.

    Returns a dict: ``counts[l]``, ``exists[l]``, ``split[l]`` (arrays over the
    C^l paths of level l), ``index[l]`` (number of a box among the existing boxes of
    its level), ``nboxes[l]``, ``unit_start`` (for every level-k cell the first cell
    of the frontier box -- leaf above level k or the cell itself -- it lies in) and
    ``cell_prefix`` (exclusive prefix sum of the histogram, length C^k + 1)."""
    C = 1 << dims
    k = int(top_level)
    hist = np.asarray(global_hist, dtype=np.int64)
    assert hist.shape == (C ** k,)
    counts = [None] * (k + 1)
    counts[k] = hist
    for lev in range(k - 1, -1, -1):
        counts[lev] = counts[lev + 1].reshape(-1, C).sum(axis=1)
    exists = [None] * (k + 1)
    split = [None] * (k + 1)
    exists[0] = np.ones(1, dtype=bool)
    for lev in range(k + 1):
        split[lev] = exists[lev] & (counts[lev] > max_particles_in_box)
        if lev < k:
            exists[lev + 1] = np.repeat(split[lev], C) & (counts[lev + 1] > 0)
    cells = np.arange(C ** k, dtype=np.int64)
    leaf_level = np.full(C ** k, k, dtype=np.int64)
    for lev in range(k - 1, -1, -1):
        anc = cells >> (dims * (k - lev))
        leaf_level = np.where(split[lev][anc], leaf_level, lev)
    sh = dims * (k - leaf_level)
    unit_start = (cells >> sh) << sh
    prefix = np.zeros(C ** k + 1, dtype=np.int64)
    np.cumsum(hist, out=prefix[1:])
    return dict(
        dims=dims, top_level=k, max_particles_in_box=int(max_particles_in_box),
        counts=counts, exists=exists, split=split,
        index=[np.cumsum(e) - 1 for e in exists],
        nboxes=[int(e.sum()) for e in exists],
        unit_start=unit_start, cell_prefix=prefix)


def partition_cells(global_hist, world, unit_start=None):
    """Step 3: owner rank of every cell; contiguous Morton ranges balanced by
    particle count.  Pure function of the (identical) global histogram.  With
    *unit_start* (:func:`top_tree_plan`) all cells of a frontier box of the global
    top tree get the owner of its first cell, so no global leaf straddles ranks."""
    counts = np.asarray(global_hist, dtype=np.int64)
    total = int(counts.sum())
    cum = np.cumsum(counts) - counts           # exclusive prefix
    # cell goes to the rank whose ideal range contains its first particle
    owner = np.minimum((cum * world) // max(total, 1), world - 1).astype(np.int64)
    if unit_start is not None:
        owner = owner[unit_start]
    return owner


def global_box_numbering(plan, level_counts_by_rank, rank):
    """Step 5: the numbers the boxes of rank *rank*'s local tree carry in the global
    (single-GPU) tree.  *level_counts_by_rank* [world][nlevels_max]: boxes per level
    of every rank's local tree (all-gathered).  Boxes above ``top_level`` are shared
    between ranks and numbered by their Morton path (:func:`top_tree_plan`); below,
    ranks own increasing Morton ranges, so a level is the concatenation of the
    ranks' level slices.

    Returns ``(global_level_start_box_nrs, deep_base)``: ``deep_base[l]`` is the
    global number of this rank's first level-l box, for l > top_level."""
    k = plan["top_level"]
    lc = np.asarray(level_counts_by_rank, dtype=np.int64)
    nlevels = int((lc > 0).sum(axis=1).max())     # the deepest local tree
    starts = np.zeros(nlevels + 1, dtype=np.int64)
    deep_base = np.zeros(nlevels, dtype=np.int64)
    for lev in range(nlevels):
        if lev <= k:
            n = plan["nboxes"][lev]
        else:
            n = int(lc[:, lev].sum())
            deep_base[lev] = starts[lev] + int(lc[:rank, lev].sum())
        starts[lev + 1] = starts[lev] + n
    return starts, deep_base


def active_boxes(plan, global_level_starts, deep_base, level_counts, device=None):
    """Step 6: the boxes rank-local lists are built for -- every box of the shared
    top levels plus the rank's own boxes below -- as ``(mask, ranges)``: an int8
    mask over the global boxes and per level the box range [begin, end) holding
    them (``_target_boxes_mask`` / ``_active_level_ranges`` of
    :class:`~boxtree_amd.traversal.FMMTraversalBuilder`)."""
    import torch
    k = plan["top_level"]
    nlevels = len(global_level_starts) - 1
    ranges = np.zeros((nlevels, 2), dtype=np.int32)
    mask = torch.zeros(int(global_level_starts[-1]), dtype=torch.int8, device=device)
    for lev in range(nlevels):
        if lev <= k:
            b0, b1 = int(global_level_starts[lev]), int(global_level_starts[lev + 1])
        else:
            b0 = int(deep_base[lev])
            b1 = b0 + int(level_counts[lev])
        ranges[lev] = (b0, b1)
        mask[b0:b1] = 1
    return mask, ranges


def local_to_global_box_ids(tree, plan, global_level_starts, deep_base, bbox_min, root_extent):
    """int64 tensor [local nboxes] of global box numbers (see
    :func:`global_box_numbering`).  Boxes at levels <= top_level are located by the
    Morton path of their centre (a few ten thousand boxes: done on the host),
    deeper boxes by their position in the rank's level slice."""
    import torch
    k = plan["top_level"]
    dims = plan["dims"]
    lsb = np.asarray(tree.level_start_box_nrs if isinstance(tree.level_start_box_nrs, np.ndarray)
                     else tree.level_start_box_nrs.cpu().numpy(), dtype=np.int64)
    nlev = len(lsb) - 1
    nboxes = int(tree.nboxes)
    centers = tree.box_centers
    dev = centers.device
    # deep levels: global = local + (deep_base[level] - local level start)
    shift = np.zeros(max(nlev, 1), dtype=np.int64)
    for lev in range(k + 1, nlev):
        shift[lev] = int(deep_base[lev]) - int(lsb[lev])
    levels = tree.box_levels[:nboxes].long()
    out = torch.arange(nboxes, dtype=torch.int64, device=dev) + torch.from_numpy(shift).to(dev)[levels]
    # shared top levels
    ntop = int(lsb[min(k + 1, nlev)])
    ctr = centers[:, :ntop].double().cpu().numpy()
    lev_h = np.repeat(np.arange(min(k + 1, nlev)), np.diff(lsb[:min(k + 1, nlev) + 1]))
    path = np.zeros(ntop, dtype=np.int64)
    for ax in range(dims):
        # centres sit at (i + 1/2) / 2^level of the root box: floor is robust
        v = np.floor((ctr[ax] - float(bbox_min[ax])) / float(root_extent) * 2.0 ** lev_h).astype(np.int64)
        v = np.clip(v, 0, (1 << lev_h) - 1)
        for bit in range(k):
            path |= ((v >> bit) & 1) << (dims * bit + (dims - 1 - ax))
    gid = np.empty(ntop, dtype=np.int64)
    for lev in range(min(k + 1, nlev)):
        b0, b1 = int(lsb[lev]), int(lsb[lev + 1])
        gid[b0:b1] = plan["index"][lev][path[b0:b1]] + int(global_level_starts[lev])
    out[:ntop] = torch.from_numpy(gid).to(dev)
    return out


# Largest single peer-to-peer message handed to all_to_all_single.  Measured on
# MI355X / RCCL (one rank sending to itself): a 1.44 GB message arrived with its
# second half corrupted, 0.96 GB arrived intact -- messages stay well below 1 GiB.
A2A_MESSAGE_LIMIT_BYTES = 512 << 20


def all_to_all_chunked(dist, recv, send, r_split, s_split, limit_bytes=None,
                       biggest_bytes=None, self_in_place=False):
    """``dist.all_to_all_single(recv, send, r_split, s_split)`` in as many rounds as
    it takes to keep every peer-to-peer message below *limit_bytes*.  Round k moves
    the k-th slice of every peer's segment; sender and receiver cut a segment of
    length c at ``floor(k*c/R)``, so both sides agree without communicating.  A
    rank's segment for itself never enters the collective: it is one device copy.
    *biggest_bytes*: the largest peer-to-peer message of ANY rank, if the caller
    already knows it (saves the MAX all-reduce that agrees on the round count).
    *self_in_place*: the caller has already written this rank's own segment into *recv*
    (and need not have filled it in *send*).  On CUDA tensors with a backend that has the
    list form of the collective (RCCL), every round exchanges VIEWS of the two buffers --
    no staging copies of the payload on either side."""
    import torch
    if limit_bytes is None:
        limit_bytes = A2A_MESSAGE_LIMIT_BYTES
    es = send.element_size()
    me = dist.get_rank()
    s_split, r_split = list(s_split), list(r_split)
    s_off = np.concatenate([[0], np.cumsum(s_split)]).astype(np.int64)
    r_off = np.concatenate([[0], np.cumsum(r_split)]).astype(np.int64)
    assert s_split[me] == r_split[me]
    s_peer, r_peer = list(s_split), list(r_split)
    s_peer[me] = r_peer[me] = 0
    biggest = max(max(s_peer, default=0), max(r_peer, default=0)) * es
    # every rank must run the same number of rounds
    if biggest_bytes is not None:
        assert biggest_bytes >= biggest
        biggest = int(biggest_bytes)
    else:
        t = torch.tensor([biggest], dtype=torch.int64, device=send.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        biggest = int(t.item())
    rounds = max(1, -(-biggest // limit_bytes))
    # The self segment is always a device copy and every decision below uses the
    # globally agreed `biggest` only: a choice that depended on a rank's own segment
    # sizes could send some ranks into a collective the others skip.
    if s_split[me] and not self_in_place:
        recv[r_off[me]:r_off[me + 1]].copy_(send[s_off[me]:s_off[me + 1]])
    if biggest == 0:
        return 1

    def cut(c, k):
        return (k * int(c)) // rounds

    if send.is_cuda and hasattr(dist, "all_to_all") and getattr(dist, "get_backend", None) \
            and dist.get_backend() == "nccl":
        # grouped point-to-point transfers between views: nothing is staged
        for k in range(rounds):
            ins = [send[s_off[p] + cut(s_peer[p], k):s_off[p] + cut(s_peer[p], k + 1)]
                   for p in range(len(s_peer))]
            outs = [recv[r_off[p] + cut(r_peer[p], k):r_off[p] + cut(r_peer[p], k + 1)]
                    for p in range(len(r_peer))]
            dist.all_to_all(outs, ins)
        return rounds

    for k in range(rounds):
        s_sub = [cut(c, k + 1) - cut(c, k) for c in s_peer]
        r_sub = [cut(c, k + 1) - cut(c, k) for c in r_peer]
        send_k = torch.cat([send[s_off[p] + cut(s_peer[p], k):s_off[p] + cut(s_peer[p], k + 1)]
                            for p in range(len(s_peer))])
        recv_k = torch.empty(int(sum(r_sub)), dtype=recv.dtype, device=recv.device)
        dist.all_to_all_single(recv_k, send_k, r_sub, s_sub)
        off = 0
        for p in range(len(r_peer)):
            recv[r_off[p] + cut(r_peer[p], k):r_off[p] + cut(r_peer[p], k + 1)] = \
                recv_k[off:off + r_sub[p]]
            off += r_sub[p]
    return rounds


class ParticleRoute:
    """Particle identity across :func:`exchange_particles` (the torch implementation; the
    library's own is ``bt_mgpu_route``, :class:`boxtree_amd.distributed.native.ParticleRoute`):
    the stable send order of every exchanged particle set and the split sizes of its
    all-to-all-v, with which any per-particle array travels between the order of this rank's
    chunk and the order of the received particles -- the order ``user_source_ids`` /
    ``sorted_target_ids`` of the rank's tree refer to.  What the reference keeps for the same
    purpose: ``src_idx`` / ``tgt_idx`` (boxtree/distributed/__init__.py:238-248,
    distributed/calculation.py:86-142).  Every method is collective."""

    def __init__(self, dist):
        self.dist = dist
        self._sets = {}
        self._offsets = {}

    def _add(self, which, n, nrecv, s_split, r_split, order_fn):
        self._sets[which] = dict(n=int(n), nrecv=int(nrecv), s_split=list(s_split),
                                 r_split=list(r_split), order_fn=order_fn, order=None)

    def _get(self, which):
        if which not in ("sources", "targets"):
            raise ValueError("which must be 'sources' or 'targets'")
        if which not in self._sets:
            raise ValueError("no separate targets were exchanged")
        st = self._sets[which]
        if st["order"] is None:
            if st["order_fn"] is None:
                raise RuntimeError("ParticleRoute.release() was called: the send order is gone")
            st["order"] = st["order_fn"]().long()
            st["order_fn"] = None          # (the closure holds the cell indices: 4-12 B per particle)
        return st

    def release(self):
        """Drop what the route keeps of the exchange (the send order, or the cell indices it can be
        made from on demand: 4-12 bytes per particle of device memory).  For callers that keep the
        exchange's statistics but never ask who their particles are."""
        for st in self._sets.values():
            st["order"] = None
            st["order_fn"] = None

    def n_owned(self, which="sources"):
        return self._get(which)["nrecv"]

    def to_owners(self, array, which="sources"):
        """``[n]`` in the chunk's order -> ``[n_owned]`` in the order of the received particles."""
        import torch
        st = self._get(which)
        if len(array) != st["n"]:
            raise ValueError(f"ParticleRoute: {len(array)} values for {st['n']} {which}")
        send = array[st["order"]].contiguous()
        recv = torch.empty(st["nrecv"], dtype=array.dtype, device=array.device)
        all_to_all_chunked(self.dist, recv, send, st["r_split"], st["s_split"])
        return recv

    def to_callers(self, array, which="sources"):
        """The inverse: ``[n_owned]`` in received order -> ``[n]`` in the chunk's order."""
        import torch
        st = self._get(which)
        if len(array) != st["nrecv"]:
            raise ValueError(f"ParticleRoute: {len(array)} values for {st['nrecv']} owned {which}")
        back = torch.empty(st["n"], dtype=array.dtype, device=array.device)
        all_to_all_chunked(self.dist, back, array.contiguous(), st["s_split"], st["r_split"])
        out = torch.empty_like(back)
        out[st["order"]] = back
        return out

    def chunk_offset(self, which="sources"):
        """(global id of this chunk's first particle, number of particles of all chunks)"""
        import torch
        st = self._get(which)
        if which not in self._offsets:
            world, rank = self.dist.get_world_size(), self.dist.get_rank()
            mine = torch.zeros(world, dtype=torch.int64, device=st["order"].device)
            mine[rank] = st["n"]
            self.dist.all_reduce(mine)
            counts = mine.cpu().tolist()
            self._offsets[which] = (int(sum(counts[:rank])), int(sum(counts)))
        return self._offsets[which]

    def global_ids(self, which="sources", dtype=None):
        """Global user id of every received particle: the sender's chunk offset + the index in
        its chunk (int32 like the reference's ``particle_id_t`` unless *dtype* says otherwise)."""
        import torch
        st = self._get(which)
        off, total = self.chunk_offset(which)
        dtype = dtype or torch.int32
        if dtype == torch.int32 and total > np.iinfo(np.int32).max:
            raise NotImplementedError(f"{total} particles do not fit int32 ids; pass dtype=torch.int64")
        ids = torch.arange(off, off + st["n"], dtype=dtype, device=st["order"].device)
        return self.to_owners(ids, which)

    def global_user_source_ids(self, tree):
        """This rank's share of the global tree's ``user_source_ids`` (tree.py:426-431)."""
        return self.global_ids("sources")[tree.user_source_ids.long()]

    def global_sorted_target_ids(self, tree, target_offset):
        """The global tree's ``sorted_target_ids`` (tree.py:433-438) for the targets of this rank's
        chunk, in the chunk's order."""
        which = "targets" if "targets" in self._sets else "sources"
        return self.to_callers(tree.sorted_target_ids + int(target_offset), which)


def exchange_particles(actx, dist, particles, targets=None, build_kw=None,
                       top_level=None, return_plan=False, max_particles_in_box=None):
    """Steps 1-4.  Returns ``(particles, targets, build_kw, stats)`` for the local
    ``TreeBuilder`` call; ``build_kw`` gains ``_root_box=`` (the global root box) and
    the exchanged ``target_radii`` if present."""
    import torch
    build_kw = dict(build_kw or {})
    world = dist.get_world_size()
    rank = dist.get_rank()
    dims = len(particles)
    dev = particles[0].device
    target_radii = build_kw.get("target_radii")
    source_radii = build_kw.get("source_radii")

    import os
    import time
    times = {} if os.environ.get("BOXTREE_HIP_EXCHANGE_TIMES") else None
    t_last = [time.perf_counter()]

    def tick(name):
        # diagnostic only (BOXTREE_HIP_EXCHANGE_TIMES=1): synchronising sub-stage timer
        if times is None:
            return
        if dev.type == "cuda":
            torch.cuda.synchronize()
        now = time.perf_counter()
        times[name] = times.get(name, 0.0) + 1e3 * (now - t_last[0])
        t_last[0] = now

    bbox_min, bbox_max, root_extent = global_root_box(
        actx, dist, particles, targets, source_radii, target_radii)
    tick("root box")

    if top_level is None:
        # enough cells for a balanced split, few enough for a tiny all-reduce
        top_level = 5 if dims == 3 else (7 if dims == 2 else 12)
    ncells = 1 << (dims * top_level)

    native = actx is not None and particles[0].is_cuda

    def cells_of(arrs):
        """(cells, hist) -- HIP kernel on the GPU, torch ops in the CPU tests."""
        if not native:
            c = morton_cells(arrs, bbox_min, bbox_max, top_level)
            return c, torch.bincount(c, minlength=ncells)
        import ctypes as ct
        from boxtree_amd import _lib
        n = len(arrs[0])
        cells = torch.empty(n, dtype=torch.int32, device=dev)
        hist = torch.zeros(ncells, dtype=torch.int32, device=dev)
        ptrs = (ct.c_void_p * dims)(*[a.data_ptr() for a in arrs])
        bmin = (ct.c_double * dims)(*[float(v) for v in bbox_min])
        bmax = (ct.c_double * dims)(*[float(v) for v in bbox_max])
        kind = _lib.BT_F64 if arrs[0].dtype == torch.float64 else _lib.BT_F32
        actx.sync_in()
        _lib.check(actx.lib.bt_morton_cells(
            actx.handle, dims, kind, ptrs, n, bmin, bmax, top_level,
            ct.c_void_p(cells.data_ptr()), ct.c_void_p(hist.data_ptr())))
        return cells, hist.long()

    src_cells, src_hist = cells_of(particles)
    hist = src_hist.clone()
    tgt_cells = tgt_hist = None
    if targets is not None:
        tgt_cells, tgt_hist = cells_of(targets)
        hist = hist + tgt_hist
    tick("cells")
    dist.all_reduce(hist)
    ghist = hist.cpu().numpy()
    tick("hist allreduce")
    # With point particles and unit weights the top of the global tree follows from
    # the histogram alone: ownership respects its leaves and the local builds split
    # the shared top boxes where the global tree does (TreeBuilder ``_top_tree``).
    plan = None
    # (separate targets: the flags of the shared top boxes would need per-cell source
    # AND target counts; the plan is for sources == targets)
    if (max_particles_in_box is not None and targets is None
            and source_radii is None and target_radii is None
            and build_kw.get("refine_weights") is None
            and build_kw.get("kind", "adaptive") == "adaptive"):
        plan = top_tree_plan(ghist, dims, top_level, max_particles_in_box)
    owner = partition_cells(ghist, world, None if plan is None else plan["unit_start"])
    owner_t = torch.from_numpy(owner).to(dev)
    tick("plan (host)")

    stats = {"bytes_sent": 0, "top_level": top_level}
    plan_route = ParticleRoute(dist)
    stats["route"] = plan_route

    def send_order(cells, local_hist):
        """(order, send_counts): original indices grouped by owner, stable."""
        if not native:
            dest = owner_t[cells]
            return torch.argsort(dest, stable=True), torch.bincount(dest, minlength=world)
        import ctypes as ct
        from boxtree_amd import _lib
        n = len(cells)
        perm = torch.empty(n, dtype=torch.int32, device=dev)
        owner32 = owner_t.to(torch.int32)
        actx.sync_in()
        _lib.check(actx.lib.bt_bucket_permutation(
            actx.handle, ct.c_void_p(cells.data_ptr()), n, ct.c_void_p(owner32.data_ptr()),
            world, ct.c_void_p(perm.data_ptr())))
        # per-owner counts from the cell histogram of THIS rank's points
        send_counts = torch.zeros(world, dtype=torch.int64, device=dev)
        send_counts.index_add_(0, owner_t, local_hist)
        return perm, send_counts

    def take(a, order):
        if not native:
            return a[order].contiguous()
        import ctypes as ct
        from boxtree_amd import _lib
        out = torch.empty_like(a)
        actx.sync_in()
        _lib.check(actx.lib.bt_gather(
            actx.handle, a.element_size(), ct.c_void_p(a.data_ptr()),
            ct.c_void_p(order.data_ptr()), len(a), ct.c_void_p(out.data_ptr())))
        return out

    def timed_a2a(recv, send, r_split, s_split, self_in_place=False):
        # device time of the payload collectives, for the per-link rate bench.py
        # reports (events on the stream the collective is enqueued on; resolved by
        # the caller after its own synchronisation)
        if dev.type != "cuda":
            all_to_all_chunked(dist, recv, send, r_split, s_split, self_in_place=self_in_place)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        all_to_all_chunked(dist, recv, send, r_split, s_split, self_in_place=self_in_place)
        e1.record()
        stats.setdefault("a2a_events", []).append((e0, e1))

    def route(arrs, cells, local_hist, extra, keep_interleaved=False, which="sources"):
        """all-to-all-v of the coordinate arrays (+ extras) by owner of `cells`."""
        # coordinates alone are partitioned by owner in one sweep that writes the send buffer
        # (bt_partition_pack); arrays that travel with them need the permutation
        # (bt_partition_pack keeps one run per owner in LDS: BT_MGPU_MAX_RANKS = 256)
        use_partition = (native and not extra and world <= 256
                         and os.environ.get("BOXTREE_HIP_PARTITION_PACK", "1") != "0")
        if use_partition:
            order = None
            send_counts = torch.zeros(world, dtype=torch.int64, device=dev)
            send_counts.index_add_(0, owner_t, local_hist)
        else:
            order, send_counts = send_order(cells, local_hist)
        tick("bucket")
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts)
        s_split = send_counts.cpu().tolist()
        r_split = recv_counts.cpu().tolist()
        tick("counts a2a")
        nrecv = int(sum(r_split))
        outs = []
        d_ = len(arrs)
        # particle identity: the send order (made on demand where the one-sweep partition never
        # materialises it) + the splits, for ParticleRoute
        kept = order
        plan_route._add(which, len(arrs[0]), nrecv, s_split, r_split,
                        (lambda: kept) if kept is not None else (lambda: send_order(cells, local_hist)[0]))
        if native:
            # coordinates travel interleaved: one message per peer instead of d
            import ctypes as ct
            from boxtree_amd import _lib
            d = len(arrs)
            n = len(arrs[0])
            es = arrs[0].element_size()
            # the segment a rank keeps is packed straight into the receive buffer; what
            # goes to the peers into the send buffer, at the offsets of the full layout
            send = torch.empty(n * d, dtype=arrs[0].dtype, device=dev)
            recv = torch.empty(nrecv * d, dtype=arrs[0].dtype, device=dev)
            ptrs = (ct.c_void_p * d)(*[a.contiguous().data_ptr() for a in arrs])
            s_off = np.concatenate([[0], np.cumsum(s_split)]).astype(np.int64)
            r_off = np.concatenate([[0], np.cumsum(r_split)]).astype(np.int64)
            actx.sync_in()
            if use_partition:
                owner32 = owner_t.to(torch.int32)
                _lib.check(actx.lib.bt_partition_pack(
                    actx.handle, d, es, ptrs, ct.c_void_p(cells.data_ptr()), n,
                    ct.c_void_p(owner32.data_ptr()), world, rank, int(s_off[rank]), int(r_off[rank]),
                    ct.c_void_p(send.data_ptr()), ct.c_void_p(recv.data_ptr())))
            else:
                pieces = [(0, int(s_off[rank]), send, 0),
                          (int(s_off[rank]), int(s_off[rank + 1]), recv, int(r_off[rank])),
                          (int(s_off[rank + 1]), n, send, int(s_off[rank + 1]))]
                for lo, hi, dst, dst_at in pieces:
                    if hi > lo:
                        _lib.check(actx.lib.bt_gather_pack(
                            actx.handle, d, es, ptrs, ct.c_void_p(order.data_ptr() + 4 * lo), hi - lo,
                            ct.c_void_p(dst.data_ptr() + dst_at * d * es)))
            tick("pack")
            timed_a2a(recv, send, [r * d for r in r_split], [c * d for c in s_split],
                      self_in_place=True)
            tick("payload a2a")
            if keep_interleaved:
                # the tree build reads the receive buffer in place
                # (bt_tree_params.source_stride): no unpacking pass
                outs = [recv.view(nrecv, d)[:, ax] for ax in range(d)]
            else:
                outs = [torch.empty(nrecv, dtype=arrs[0].dtype, device=dev) for _ in range(d)]
                optrs = (ct.c_void_p * d)(*[o.data_ptr() for o in outs])
                actx.sync_in()
                _lib.check(actx.lib.bt_unpack(actx.handle, d, es, ct.c_void_p(recv.data_ptr()),
                                              nrecv, optrs))
            stats["bytes_sent"] += (n - s_split[rank]) * es * d
            rest = list(extra)
        else:
            rest = list(arrs) + list(extra)
        for a in rest:
            send = take(a.contiguous(), order)
            recv = torch.empty(nrecv, dtype=a.dtype, device=dev)
            timed_a2a(recv, send, r_split, s_split)
            outs.append(recv)
            stats["bytes_sent"] += (len(a) - s_split[rank]) * a.element_size()
        return outs[:len(arrs)], outs[len(arrs):]

    interleaved = (native and targets is None and source_radii is None and dims > 1
                   and build_kw.get("refine_weights") is None)
    new_particles, extra = route(
        particles, src_cells, src_hist, [source_radii] if source_radii is not None else [],
        keep_interleaved=interleaved)
    if interleaved and not new_particles[0].is_contiguous():
        build_kw["_point_stride"] = dims
    if source_radii is not None:
        build_kw["source_radii"] = extra[0]
    new_targets = None
    if targets is not None:
        new_targets, extra = route(
            targets, tgt_cells, tgt_hist, [target_radii] if target_radii is not None else [],
            which="targets")
        if target_radii is not None:
            build_kw["target_radii"] = extra[0]

    # hand the agreed root box to TreeBuilder verbatim (the public ``bbox=`` path
    # re-derives root_extent from max-min and asserts squareness to 1e-15, which a
    # rounded ``min + extent`` need not satisfy)
    build_kw["_root_box"] = (bbox_min, bbox_max, root_extent)
    if plan is not None and native:
        build_kw["_top_tree"] = (top_level, torch.from_numpy(plan["cell_prefix"]).to(dev))
    stats["plan"] = plan
    stats["bbox_min"], stats["bbox_max"] = bbox_min, bbox_max
    stats["root_extent"] = root_extent
    stats["owner"] = owner
    if times is not None:
        tick("rest")
        stats["times_ms"] = times
    return new_particles, new_targets, build_kw, stats


def number_sharded_tree(dist, tree, stats):
    """Step 5 for the local *tree* of this rank (built from the output of
    :func:`exchange_particles`, whose *stats* carry the top-tree plan): all-gathers
    the per-level box counts and particle counts and returns the placement of the
    local tree inside the global one::

        box_ids                      int64 [local nboxes]: global box numbers
        global_level_start_box_nrs   int64 [global nlevels + 1]
        source_offset/target_offset  global tree-order index of local particle 0
        nboxes, nsources, ntargets   global totals

    Renumbering the local arrays with these gives, rank by rank, exactly the
    slices of the tree a single GPU builds from the concatenated input
    (tests/test_gpu_parity.py::test_sharded_build_global_numbering)."""
    import torch
    plan = stats.get("plan")
    if plan is None:
        raise NotImplementedError(
            "global numbering needs the top-tree plan: kind='adaptive', point particles, "
            "unit refine weights, max_particles_in_box passed to exchange_particles")
    world = dist.get_world_size()
    rank = dist.get_rank()
    lsb = tree.level_start_box_nrs
    lsb = np.asarray(lsb if isinstance(lsb, np.ndarray) else lsb.cpu().numpy(), dtype=np.int64)
    nmax = 64
    dev = tree.box_centers.device
    mine = torch.zeros(nmax + 2, dtype=torch.int64, device=dev)
    mine[:len(lsb) - 1] = torch.from_numpy(np.diff(lsb)).to(dev)
    mine[nmax] = int(tree.nsources)
    mine[nmax + 1] = int(tree.ntargets)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    allc = torch.stack(gathered).cpu().numpy()
    starts, deep_base = global_box_numbering(plan, allc[:, :nmax], rank)
    box_ids = local_to_global_box_ids(tree, plan, starts, deep_base, stats["bbox_min"],
                                      stats["root_extent"])
    mask, ranges = active_boxes(plan, starts, deep_base, allc[rank, :nmax], dev)
    return dict(
        nboxes_by_rank=allc[:, :nmax].sum(axis=1), target_boxes_mask=mask,
        active_level_ranges=ranges, deep_base=deep_base,
        box_ids=box_ids, global_level_start_box_nrs=starts,
        source_offset=int(allc[:rank, nmax].sum()), target_offset=int(allc[:rank, nmax + 1].sum()),
        nboxes=int(starts[-1]), nsources=int(allc[:, nmax].sum()),
        ntargets=int(allc[:, nmax + 1].sum()))


def gather_global_box_tree(actx, dist, tree, numbering):
    """Step 6: all-gathers the box arrays of the ranks' local trees into the global
    numbering.  Every rank gets the complete :class:`~boxtree_amd.tree.TreeOfBoxes`
    (centres, levels, flags, parents, children, level starts) the traversal kernels
    walk; particles stay where they are.  ``box_child_ids`` of the shared top boxes
    are the union of the ranks' views (their other fields agree between ranks).
    Two collectives: one int32 record per box (number, parent, level, flags,
    children) and the centres."""
    import torch

    from boxtree_amd.tree import TreeOfBoxes
    world = dist.get_world_size()
    dims = int(tree.dimensions)
    C = 1 << dims
    nb = int(tree.nboxes)
    ids = numbering["box_ids"]
    dev = ids.device
    counts = [int(c) for c in numbering["nboxes_by_rank"]]
    nmax = max(counts)
    B = int(numbering["nboxes"])
    aligned = -(-B // 32) * 32
    ch = tree.box_child_ids[:, :nb].long()
    rec = torch.zeros((nmax, C + 4), dtype=torch.int32, device=dev)
    rec[:nb, 0] = ids.to(torch.int32)
    rec[:nb, 1] = ids[tree.box_parent_ids[:nb].long()].to(torch.int32)
    rec[:nb, 2] = tree.box_levels[:nb].to(torch.int32)
    rec[:nb, 3] = tree.box_flags[:nb].to(torch.int32)
    rec[:nb, 4:] = torch.where(ch != 0, ids[ch], torch.zeros_like(ch)).to(torch.int32).t()
    ctr = torch.zeros((nmax, dims), dtype=tree.box_centers.dtype, device=dev)
    ctr[:nb] = tree.box_centers[:, :nb].t()
    def all_gather_rows(t):
        """all_gather of a [nmax, w] tensor, rows in slabs below the message limit."""
        out = [torch.empty_like(t) for _ in range(world)]
        row_bytes = max(1, t.shape[1] * t.element_size())
        step = max(1, A2A_MESSAGE_LIMIT_BYTES // row_bytes)
        if t.shape[0] <= step:
            dist.all_gather(out, t)
            return out
        for r0 in range(0, t.shape[0], step):
            part = t[r0:r0 + step].contiguous()
            got = [torch.empty_like(part) for _ in range(world)]
            dist.all_gather(got, part)
            for r in range(world):
                out[r][r0:r0 + step] = got[r]
        return out

    g_rec = all_gather_rows(rec)
    g_ctr = all_gather_rows(ctr)
    rec_all = torch.cat([g_rec[r][:counts[r]] for r in range(world)])
    ctr_all = torch.cat([g_ctr[r][:counts[r]] for r in range(world)])
    gi = rec_all[:, 0].long()
    centers = torch.zeros((dims, aligned), dtype=tree.box_centers.dtype, device=dev)
    centers[:, gi] = ctr_all.t()
    levels = torch.zeros(B, dtype=torch.uint8, device=dev)
    levels[gi] = rec_all[:, 2].to(torch.uint8)
    flags = torch.zeros(B, dtype=torch.uint8, device=dev)
    flags[gi] = rec_all[:, 3].to(torch.uint8)
    parents = torch.zeros(B, dtype=torch.int32, device=dev)
    parents[gi] = rec_all[:, 1]
    # shared top boxes come from several ranks, each knowing the children that hold
    # its own particles: merge by maximum (absent = 0)
    children_t = torch.zeros((aligned, C), dtype=torch.int32, device=dev)
    children_t.scatter_reduce_(0, gi[:, None].expand(-1, C), rec_all[:, 4:], "amax",
                               include_self=True)
    children = children_t.t().contiguous()
    starts = numbering["global_level_start_box_nrs"].astype(np.int32)
    coord_dtype = np.dtype(str(tree.box_centers.dtype).replace("torch.", ""))
    return TreeOfBoxes(
        root_extent=tree.root_extent, box_centers=centers, box_parent_ids=parents,
        box_child_ids=children, box_levels=levels, box_flags=flags,
        level_start_box_nrs=starts, box_id_dtype=np.dtype(np.int32),
        box_level_dtype=np.dtype(np.uint8), coord_dtype=coord_dtype,
        sources_have_extent=False, targets_have_extent=False, extent_norm=None,
        stick_out_factor=tree.stick_out_factor, _is_pruned=True)


# {{{ local essential tree (step 6 without the global all-gather)

_CELL_COORDS = {}


_CELL_INDEX_GRID = {}


def _cell_grid(values, dims, k):
    """Morton-indexed cell array -> [2^k]^dims grid (axis 0 = x)."""
    n = 1 << k
    if (dims, k) not in _CELL_COORDS:
        cells = np.arange(n ** dims, dtype=np.int64)
        coords = []
        for ax in range(dims):
            v = np.zeros_like(cells)
            for bit in range(k):
                v |= ((cells >> (dims * bit + (dims - 1 - ax))) & 1) << bit
            coords.append(v)
        _CELL_COORDS[dims, k] = coords
    coords = _CELL_COORDS[dims, k]
    grid = np.zeros((n,) * dims, dtype=values.dtype)
    grid[tuple(coords)] = values
    return grid, coords


def _cells_needed_by(owner, dims, k, rank, world, ring, counts=None):
    """need[q] = boolean array over cells: my cells within *ring* cells (Chebyshev) of
    a cell owned by rank q -- the subtrees q's lists can reach.  Per rank q: the set
    of q's cells is dilated by *ring* along every axis of the cell grid (whole-array
    shifts) and intersected with my non-empty cells."""
    owner = np.asarray(owner, dtype=np.int64)
    n = 1 << k
    key = (dims, k)
    if key not in _CELL_INDEX_GRID:
        _CELL_INDEX_GRID[key] = _cell_grid(np.arange(n ** dims, dtype=np.int64), dims, k)
    index_grid, _ = _CELL_INDEX_GRID[key]
    owner_grid = owner[index_grid]
    mine = owner == rank
    if counts is not None:
        mine = mine & (np.asarray(counts) > 0)      # empty cells hold no boxes
    need = np.zeros((world, n ** dims), dtype=bool)
    if not mine.any():
        return need
    mine_grid = mine[index_grid]
    for q in np.unique(owner):
        q = int(q)
        if q == rank:
            continue
        d = owner_grid == q
        for ax in range(dims):
            acc = d.copy()
            for sh in range(1, ring + 1):
                lo = [slice(None)] * dims
                hi = [slice(None)] * dims
                lo[ax], hi[ax] = slice(0, n - sh), slice(sh, n)
                acc[tuple(lo)] |= d[tuple(hi)]
                acc[tuple(hi)] |= d[tuple(lo)]
            d = acc
        need[q, index_grid[d & mine_grid]] = True
    return need


def build_local_essential_tree(actx, dist, tree, stats, numbering, well_sep_is_n_away=1):
    """Step 6 with a halo instead of the all-gather of :func:`gather_global_box_tree`:
    the tree this rank needs for the lists of its own boxes -- the shared top levels
    (known to every rank from the plan), its own subtrees, and the subtrees of the
    cells of other ranks within *well_sep_is_n_away* cells of its own cells, which
    those ranks send as (Morton path, level, flags, global number) records.  Boxes
    are numbered level-major, Morton order within a level (what the parent-colleague
    kernels expect); parents, children and centres are recomputed from the paths.

    Returns ``(let, info)``: *let* a :class:`~boxtree_amd.tree.TreeOfBoxes`, *info*
    with ``target_boxes_mask`` / ``active_level_ranges`` for
    :class:`~boxtree_amd.traversal.FMMTraversalBuilder` and ``global_box_ids``
    (LET number -> global number)."""
    import ctypes as ct

    import torch

    from boxtree_amd import _lib
    from boxtree_amd.tree import TreeOfBoxes
    plan = stats["plan"]
    if plan is None:
        raise NotImplementedError("the local essential tree needs the top-tree plan")
    world, rank = dist.get_world_size(), dist.get_rank()
    dims, k = plan["dims"], plan["top_level"]
    C = 1 << dims
    nb = int(tree.nboxes)
    dev = tree.box_centers.device
    coord_t = tree.box_centers.dtype
    coord_dtype = np.dtype(str(coord_t).replace("torch.", ""))
    gstarts = numbering["global_level_start_box_nrs"]
    nlev = len(gstarts) - 1
    ntop_levels = min(k + 1, nlev)
    bbox_min, bbox_max = stats["bbox_min"], stats["bbox_max"]
    root_extent = stats["root_extent"]
    kind = _lib.BT_F64 if coord_dtype == np.float64 else _lib.BT_F32

    import os
    import time
    times = {} if os.environ.get("BOXTREE_HIP_EXCHANGE_TIMES") else None
    t_last = [time.perf_counter()]

    def tick(name):     # diagnostic only, see exchange_particles
        if times is None:
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        times[name] = times.get(name, 0.0) + 1e3 * (now - t_last[0])
        t_last[0] = now

    # -- Morton paths of my boxes ------------------------------------------------------
    paths = torch.empty(nb, dtype=torch.int64, device=dev)
    bmin = (ct.c_double * 3)(*[float(v) for v in bbox_min], *([0.0] * (3 - dims)))
    actx.sync_in()
    _lib.check(actx.lib.bt_box_morton_paths(
        actx.handle, dims, kind, nb, int(tree.aligned_nboxes),
        ct.c_void_p(tree.box_centers.data_ptr()), ct.c_void_p(tree.box_levels.data_ptr()), bmin,
        float(root_extent), ct.c_void_p(paths.data_ptr())))
    levels = tree.box_levels[:nb].long()
    deep = levels > k
    gids = numbering["box_ids"]
    meta = levels.to(torch.int32) | (tree.box_flags[:nb].to(torch.int32) << 8)
    tick("paths+meta")

    # -- halo: my deep boxes in the cells other ranks' lists can reach -----------------------
    need = _cells_needed_by(stats["owner"], dims, k, rank, world, int(well_sep_is_n_away),
                            counts=plan["counts"][k])
    deep_idx = torch.nonzero(deep).flatten()
    cell_of_deep = paths[deep_idx] >> (dims * (levels[deep_idx] - k))
    send_idx, s_split = [], []
    need_t = torch.from_numpy(need).to(dev)
    for q in range(world):
        if q == rank or not need[q].any():
            s_split.append(0)
            continue
        sel = deep_idx[need_t[q][cell_of_deep]]
        send_idx.append(sel)
        s_split.append(int(sel.shape[0]))
    send_idx = torch.cat(send_idx) if send_idx else deep_idx[:0]
    # one record per box: (path, level | flags << 8 | global id << 32).  The counts
    # exchange also carries every rank's largest message, so all ranks know the
    # round count of the payload exchange without a further collective.
    rec_bytes = 16
    counts = torch.tensor([[c, max(s_split) * rec_bytes] for c in s_split], dtype=torch.int64,
                          device=dev).reshape(-1)
    rcounts = torch.empty_like(counts)
    dist.all_to_all_single(rcounts, counts)
    rc = rcounts.cpu().numpy().reshape(world, 2)
    r_split = [int(c) for c in rc[:, 0]]
    biggest = int(max(rc[:, 1].max(), max(s_split) * rec_bytes))
    nrecv = sum(r_split)
    h_paths = paths[:0]
    h_meta = meta[:0]
    h_gids = gids[:0].to(torch.int32)
    if biggest:
        rec = torch.stack([paths[send_idx],
                           meta[send_idx].long() | (gids[send_idx].long() << 32)], dim=1)
        got = torch.empty(nrecv * 2, dtype=torch.int64, device=dev)
        all_to_all_chunked(dist, got, rec.reshape(-1), [2 * c for c in r_split],
                           [2 * c for c in s_split], biggest_bytes=biggest)
        got = got.view(nrecv, 2)
        h_paths = got[:, 0]
        h_meta = (got[:, 1] & 0xffffffff).to(torch.int32)
        h_gids = (got[:, 1] >> 32).to(torch.int32)
    tick("halo exchange")

    # -- the box set: top levels from the plan, my deep boxes, the halo ------------------------
    owner = np.asarray(stats["owner"], dtype=np.int64)
    t_paths, t_meta, t_gid, t_mine, top_counts = [], [], [], [], []
    for lev in range(ntop_levels):
        p = np.nonzero(plan["exists"][lev])[0].astype(np.int64)
        internal = plan["split"][lev][p]
        flags = np.where(internal, 12, 3).astype(np.int64)      # tree.py:109-145 (sources = targets)
        # lists of the shared internal boxes are built by every rank, those of a top
        # LEAF only by the rank that owns its cells
        first_cell = p << (dims * (k - lev))
        t_paths.append(p)
        t_meta.append(lev | (flags << 8))
        t_gid.append(int(gstarts[lev]) + np.arange(len(p), dtype=np.int64))
        t_mine.append((internal | (owner[first_cell] == rank)).astype(np.int64))
        top_counts.append(len(p))
    top = torch.from_numpy(np.stack([np.concatenate(t_paths), np.concatenate(t_meta),
                                     np.concatenate(t_gid), np.concatenate(t_mine)])).to(dev)
    lvl_paths = [top[0]]
    lvl_meta = [top[1].to(torch.int32)]
    lvl_gid = [top[2].to(torch.int32)]
    lvl_mine = [top[3] != 0]
    d_paths = torch.cat([paths[deep_idx], h_paths])
    d_meta = torch.cat([meta[deep_idx], h_meta])
    d_gid = torch.cat([gids[deep_idx].to(torch.int32), h_gids])
    nd = int(d_paths.shape[0])
    n_mine = int(deep_idx.shape[0])
    d_lev = (d_meta & 0xff).long()
    tick("concat")
    # order the deep boxes by (level, Morton path): a stable sort by the path
    # left-aligned to the deepest level (ancestors tie with their first
    # descendants), then a stable one-digit sort by level
    lmax = nlev - 1
    order = torch.arange(nd, dtype=torch.int32, device=dev)
    if nd:
        key = (d_paths << (dims * (lmax - d_lev))).contiguous()
        key_out = torch.empty_like(key)
        order1 = torch.empty_like(order)
        actx.sync_in()
        _lib.check(actx.lib.bt_radix_sort_u64_u32(
            actx.handle, ct.c_void_p(key.data_ptr()), ct.c_void_p(order.data_ptr()),
            ct.c_void_p(key_out.data_ptr()), ct.c_void_p(order1.data_ptr()), nd, 0,
            max(1, min(64, dims * lmax))))
        lev_key = d_lev[order1.long()].to(torch.int32).contiguous()
        lev_out = torch.empty_like(lev_key)
        order = torch.empty_like(order1)
        actx.sync_in()
        _lib.check(actx.lib.bt_radix_sort_u32_u32(
            actx.handle, ct.c_void_p(lev_key.data_ptr()), ct.c_void_p(order1.data_ptr()),
            ct.c_void_p(lev_out.data_ptr()), ct.c_void_p(order.data_ptr()), nd, 0, 8))
    order = order.long()
    tick("sort")
    s_paths, s_meta, s_gid = d_paths[order], d_meta[order], d_gid[order]
    s_mine = order < n_mine
    s_lev = (s_meta & 0xff).long()
    # (sorted by level: boundaries by binary search, no histogram atomics)
    bounds = torch.searchsorted(s_lev.contiguous(),
                                torch.arange(nlev + 1, device=dev)).cpu().numpy()
    deep_counts = np.diff(bounds)
    # my boxes of a level are one contiguous run (my cells are one Morton range)
    mpos = torch.nonzero(s_mine).flatten()          # ascending; levels non-decreasing
    mlev = s_lev[mpos].contiguous()
    probe = torch.arange(nlev, device=dev)
    lo = torch.searchsorted(mlev, probe).cpu().numpy()
    hi = torch.searchsorted(mlev, probe, right=True).cpu().numpy()
    mpos_h = None
    mine_counts = hi - lo
    first_h = np.zeros(nlev, dtype=np.int64)
    last_h = np.zeros(nlev, dtype=np.int64)
    if int(mpos.shape[0]):
        ends = torch.from_numpy(np.stack([np.minimum(lo, len(mlev) - 1),
                                          np.maximum(hi - 1, 0)])).to(dev)
        mpos_h = mpos[ends].cpu().numpy()
        first_h, last_h = mpos_h[0], mpos_h[1]
    ranges = np.zeros((nlev, 2), dtype=np.int32)
    tick("reorder+bounds")
    level_starts = [0]
    ntop = int(sum(top_counts))
    deep_off = 0
    for lev in range(nlev):
        if lev < ntop_levels:
            n_lev = int(top_counts[lev])
            ranges[lev] = (level_starts[-1], level_starts[-1] + n_lev)
        else:
            n_lev = int(deep_counts[lev])
            if mine_counts[lev]:
                assert last_h[lev] - first_h[lev] + 1 == mine_counts[lev]
                ranges[lev] = (ntop + first_h[lev], ntop + last_h[lev] + 1)
            else:
                ranges[lev] = (level_starts[-1], level_starts[-1])
            deep_off += n_lev
        level_starts.append(level_starts[-1] + n_lev)
    lvl_paths.append(s_paths)
    lvl_meta.append(s_meta)
    lvl_gid.append(s_gid)
    lvl_mine.append(s_mine)
    B = level_starts[-1]
    aligned = -(-B // 32) * 32
    all_paths = torch.cat(lvl_paths).contiguous()
    all_meta = torch.cat(lvl_meta)
    let_gid = torch.cat(lvl_gid).contiguous()
    mask = torch.cat(lvl_mine).to(torch.int8).contiguous()
    tick("assemble")

    parents = torch.zeros(B, dtype=torch.int32, device=dev)
    children = torch.empty((C, aligned), dtype=torch.int32, device=dev)   # cleared by bt_let_build
    centers = torch.zeros((dims, aligned), dtype=coord_t, device=dev)
    lsb = np.asarray(level_starts, dtype=np.int32)
    bmax = (ct.c_double * 3)(*[float(v) for v in bbox_max], *([0.0] * (3 - dims)))
    actx.sync_in()
    _lib.check(actx.lib.bt_let_build(
        actx.handle, dims, kind, nlev, lsb.ctypes.data_as(ct.POINTER(ct.c_int32)),
        ct.c_void_p(all_paths.data_ptr()), aligned, bmin, bmax, float(root_extent),
        ct.c_void_p(parents.data_ptr()), ct.c_void_p(children.data_ptr()),
        ct.c_void_p(centers.data_ptr())))
    let = TreeOfBoxes(
        root_extent=tree.root_extent, box_centers=centers, box_parent_ids=parents,
        box_child_ids=children, box_levels=(all_meta & 0xff).to(torch.uint8),
        box_flags=((all_meta >> 8) & 0xff).to(torch.uint8), level_start_box_nrs=lsb,
        box_id_dtype=np.dtype(np.int32), box_level_dtype=np.dtype(np.uint8),
        coord_dtype=coord_dtype, sources_have_extent=False, targets_have_extent=False,
        extent_norm=None, stick_out_factor=tree.stick_out_factor, _is_pruned=True)
    tick("links")
    info = dict(target_boxes_mask=mask, active_level_ranges=ranges, global_box_ids=let_gid,
                halo_boxes_received=nrecv, halo_boxes_sent=int(send_idx.shape[0]),
                nboxes=B)
    if times is not None:
        info["times_ms"] = times
    return let, info

# }}}


# {{{ distributed FMM evaluation on a replicated global tree
#     (boxtree/distributed/__init__.py:156-311)

def broadcast_tree(actx, tree, comm, src=0):
    """Every rank returns a device copy of rank *src*'s *tree* (the others pass
    ``None``): one object broadcast of the field layout, then one broadcast per
    array, GPU to GPU."""
    import dataclasses
    import torch
    from boxtree_amd.array_context import _torch_dtype, make_obj_array, np_dtype_of
    if comm.get_world_size() == 1:
        return tree
    is_src = comm.get_rank() == src

    def describe(v):
        if isinstance(v, torch.Tensor):
            return ("array", tuple(v.shape), np_dtype_of(v).str)
        if isinstance(v, np.ndarray) and v.dtype.char == "O":
            return ("objarray", [describe(a) for a in v])
        return ("value", v)

    layout = [None]
    if is_src:
        memo = {}
        fields = []
        for f in dataclasses.fields(tree):
            v = getattr(tree, f.name)
            alias = memo.setdefault(id(v), f.name)
            fields.append((f.name, ("alias", alias) if alias != f.name else describe(v)))
        layout = [(type(tree), fields)]
    comm.broadcast_object_list(layout, src=src)
    cls, fields = layout[0]

    def move(desc, v):
        if desc[0] == "array":
            t = (v.contiguous() if is_src else
                 torch.empty(desc[1], dtype=_torch_dtype(torch, np.dtype(desc[2])),
                             device=actx.device))
            if t.numel():
                comm.broadcast(t, src=src)
            return t
        if desc[0] == "objarray":
            return make_obj_array([move(d, v[i] if is_src else None)
                                   for i, d in enumerate(desc[1])])
        return desc[1]

    out = {}
    for name, desc in fields:
        if desc[0] == "alias":
            out[name] = out[desc[1]]
        else:
            out[name] = move(desc, getattr(tree, name) if is_src else None)
    return tree if is_src else cls(**out)


def make_distributed_wrangler(actx, global_tree, traversal_builder, wrangler_factory,
                              calibration_params, comm):
    """Collective.  Replicates rank 0's *global_tree*, builds the global traversal on
    every rank, cuts the boxes into per-rank shares by modelled cost, and returns
    ``(wrangler, src_idx_all_ranks, tgt_idx_all_ranks)`` with the wrangler made by
    ``wrangler_factory(local_traversal, global_traversal)``
    (boxtree/distributed/__init__.py:156-271).  The index lists are only populated on
    rank 0."""
    import warnings
    from boxtree_amd.cost import FMMCostModel
    from boxtree_amd.distributed.calculation import gather_to_root
    from boxtree_amd.distributed.local_traversal import generate_local_travs
    from boxtree_amd.distributed.local_tree import generate_local_tree
    from boxtree_amd.distributed.partition import partition_work

    global_tree = broadcast_tree(actx, global_tree, comm)
    global_trav, _ = traversal_builder(actx, global_tree)

    cost_per_box = None
    if comm.get_rank() == 0:
        if calibration_params is None:
            warnings.warn("Calibration parameters for the cost model are not supplied. "
                          "The default one will be used.", stacklevel=2)
            calibration_params = FMMCostModel.get_unit_calibration_params()
        # a wrangler on the global traversal, for its expansion orders
        global_wrangler = wrangler_factory(global_trav, global_trav)
        cost_per_box = FMMCostModel().cost_per_box(
            actx, global_trav, global_wrangler.level_orders, calibration_params)
    responsible_boxes_list = partition_work(actx, cost_per_box, global_trav, comm)

    local_tree, src_idx, tgt_idx = generate_local_tree(
        actx, global_trav, responsible_boxes_list, comm)
    src_idx_all_ranks = gather_to_root(actx, comm, src_idx)
    tgt_idx_all_ranks = gather_to_root(actx, comm, tgt_idx)
    local_trav = generate_local_travs(actx, local_tree, traversal_builder)
    wrangler = wrangler_factory(local_trav, global_trav)
    return wrangler, src_idx_all_ranks, tgt_idx_all_ranks


class DistributedFMMRunner:
    """Sets up and runs a distributed point FMM (boxtree/distributed/__init__.py:
    274-311): *global_tree* matters on rank 0 only, *comm* is ``torch.distributed``
    (or an object with its collectives), potentials come back on rank 0."""

    def __init__(self, array_context, global_tree, traversal_builder, wrangler_factory,
                 calibration_params=None, comm=None):
        if comm is None:
            import torch.distributed as comm
        self.wrangler, self.src_idx_all_ranks, self.tgt_idx_all_ranks = \
            make_distributed_wrangler(array_context, global_tree, traversal_builder,
                                      wrangler_factory, calibration_params, comm)

    def drive_dfmm(self, actx, source_weights):
        from boxtree_amd.fmm import drive_fmm
        return drive_fmm(actx, self.wrangler, source_weights,
                         global_src_idx_all_ranks=self.src_idx_all_ranks,
                         global_tgt_idx_all_ranks=self.tgt_idx_all_ranks)

# }}}
