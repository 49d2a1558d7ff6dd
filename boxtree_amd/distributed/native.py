"""The sharded build through the library's own multi-GPU entries (``bt_mgpu_*``,
``csrc/bt_mgpu.hip``): particle exchange, global numbering and the local essential tree
run in C++/HIP behind the C ABI, over an RCCL communicator -- or over ranks that are
threads of one process (:class:`LocalGroup`), which is how the multi-rank logic is
exercised on a box with one GPU.  This module is the thin Python caller; the torch
implementation of the same steps in :mod:`boxtree_amd.distributed` remains for the CPU
(gloo) tests of the plan and for process groups whose ranks share a GPU.

SURVEY.md section 8e steps 1-6; the reference has no counterpart (it builds on one rank,
boxtree/distributed/__init__.py:183-199).
"""

from __future__ import annotations

import ctypes as ct
import os

import numpy as np

from boxtree_amd import _lib


class _DevicePointer:
    """A device allocation that is not torch's, for ``torch.as_tensor`` (zero copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class NativeComm:
    """Handle on a ``bt_mgpu_comm``."""

    def __init__(self, lib, handle, rank, nranks, kind, keep=None):
        self.lib, self.handle, self.rank, self.nranks, self.kind = lib, handle, rank, nranks, kind
        self._keep = keep

    def close(self):
        if self.handle:
            self.lib.bt_mgpu_comm_destroy(self.handle)
            self.handle = None
        if self._keep is not None and hasattr(self._keep, "close"):
            self._keep.close()
            self._keep = None


class LocalGroup:
    """*nranks* ranks as threads of this process (one ``bt_context`` each, on any device):
    every thread takes ``group.comm(rank)`` and makes the same sequence of ``bt_mgpu_*``
    calls.  For tests: RCCL refuses two ranks on one GPU."""

    def __init__(self, nranks):
        self.lib = _lib.load()
        self.nranks = nranks
        g = ct.c_void_p()
        _lib.check(self.lib.bt_mgpu_local_group_create(nranks, ct.byref(g)))
        self.handle = g

    def comm(self, rank):
        h = ct.c_void_p()
        _lib.check(self.lib.bt_mgpu_comm_local(self.handle, rank, ct.byref(h)))
        return NativeComm(self.lib, h, rank, self.nranks, "threads")

    def close(self):
        if self.handle:
            self.lib.bt_mgpu_local_group_destroy(self.handle)
            self.handle = None


def shm_comm(name, rank, nranks, slot_bytes=0, timeout_s=0.0):
    """A :class:`NativeComm` whose ranks are PROCESSES that share a GPU (``bt_mgpu_comm_shm``): the
    collectives are staged through the POSIX shared-memory segment *name* (``"/..."``, the same fresh
    name on every rank of the job).  RCCL refuses two ranks on one device; this is how the N-rank
    code runs as N real processes on a one-GPU box -- a correctness vehicle, nothing to time."""
    lib = _lib.load()
    h = ct.c_void_p()
    _lib.check(lib.bt_mgpu_comm_shm(name.encode(), int(rank), int(nranks), int(slot_bytes), float(timeout_s),
                                    ct.byref(h)))
    return NativeComm(lib, h, int(rank), int(nranks), "processes")


class _RcclComm:
    """An RCCL communicator of our own, created through ctypes (torch does not hand out
    the ``ncclComm_t`` of its process groups).  The unique id travels over *dist*."""

    def __init__(self, dist, device_index):
        import torch
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.rccl = ct.CDLL(path)
        # the library must call into THIS image of RCCL (the one that makes the communicator)
        _lib.check(_lib.load().bt_mgpu_use_rccl_library(path.encode()))

        class UniqueId(ct.Structure):
            _fields_ = [("internal", ct.c_char * 128)]

        rank, world = dist.get_rank(), dist.get_world_size()
        # (before the broadcast: an NCCL process group moves the pickled id through the
        # current device)
        torch.cuda.set_device(device_index)
        uid = UniqueId()
        if rank == 0:
            if self.rccl.ncclGetUniqueId(ct.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
        box = [bytes(uid)]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
            ct.memmove(ct.byref(uid), box[0], ct.sizeof(uid))
        self.comm = ct.c_void_p()
        self.rccl.ncclCommInitRank.argtypes = [ct.POINTER(ct.c_void_p), ct.c_int, UniqueId, ct.c_int]
        code = self.rccl.ncclCommInitRank(ct.byref(self.comm), world, uid, rank)
        if code != 0:
            raise RuntimeError(f"ncclCommInitRank failed with code {code}")

    def close(self):
        if self.comm:
            self.rccl.ncclCommDestroy(self.comm)
            self.comm = None


def rccl_comm(actx, dist, self_loopback=None):
    """A :class:`NativeComm` over RCCL with the ranks of *dist* (collective).
    *self_loopback* (default: the environment variable ``BT_MGPU_SELF_LOOPBACK``) sends a
    rank's messages to itself through ``ncclSend`` / ``ncclRecv`` as well -- the switch that
    lets a one-GPU box execute the point-to-point branch."""
    rc = _RcclComm(dist, actx.device_index)
    h = ct.c_void_p()
    _lib.check(actx.lib.bt_mgpu_comm_rccl(rc.comm, dist.get_rank(), dist.get_world_size(),
                                          ct.byref(h)))
    if self_loopback is not None:
        _lib.check(actx.lib.bt_mgpu_comm_set_self_loopback(h, int(bool(self_loopback))))
    return NativeComm(actx.lib, h, dist.get_rank(), dist.get_world_size(), "rccl", keep=rc)


def exchange_time_ms(actx):
    """Device time of the payload all-to-all-v of the last exchange on *actx* (waits for it)."""
    ms = ct.c_float()
    _lib.check(actx.lib.bt_mgpu_exchange_time(actx.handle, ct.byref(ms)))
    return float(ms.value)


class _LazyA2aTime:
    """``stats["a2a_ms"]`` of an exchange on a stream-ordered context, which returns with the
    all-to-all-v still queued: a number that is resolved when first used (``float()``,
    formatting, arithmetic, comparisons).  The library keeps ONE event pair per context, so
    the value must be read before the next exchange on the same context: a later one
    invalidates it (``float()`` then raises instead of reporting another exchange's time)."""

    def __init__(self, actx, serial):
        self.actx, self.serial, self.value = actx, serial, None

    def __float__(self):
        if self.value is None:
            if getattr(self.actx, "_mgpu_exchange_serial", None) != self.serial:
                raise RuntimeError("a2a_ms was not read before the next exchange on this context")
            self.value = exchange_time_ms(self.actx)
        return self.value

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))

    def __bool__(self):
        return float(self) != 0.0

    def __round__(self, ndigits=None):
        return round(float(self), ndigits)

    def __eq__(self, other):
        return float(self) == other

    def __lt__(self, other):
        return float(self) < other

    def __le__(self, other):
        return float(self) <= other

    def __gt__(self, other):
        return float(self) > other

    def __ge__(self, other):
        return float(self) >= other

    def __hash__(self):
        return hash(float(self))

    def __add__(self, other):
        return float(self) + other

    __radd__ = __add__

    def __sub__(self, other):
        return float(self) - other

    def __rsub__(self, other):
        return other - float(self)

    def __mul__(self, other):
        return float(self) * other

    __rmul__ = __mul__

    def __truediv__(self, other):
        return float(self) / other

    def __rtruediv__(self, other):
        return other / float(self)

    def __neg__(self):
        return -float(self)


def _a2a_time(actx, shard):
    """Device time of the payload all-to-all-v: a float where the exchange waited for it, a
    number resolved on first use on a stream-ordered context."""
    serial = getattr(actx, "_mgpu_exchange_serial", 0) + 1
    actx._mgpu_exchange_serial = serial
    if float(shard.a2a_ms) >= 0:
        return float(shard.a2a_ms)
    return _LazyA2aTime(actx, serial)


class ParticleRoute:
    """Particle identity across the exchange (``bt_mgpu_route`` / ``bt_mgpu_global_ids``): moves
    any per-particle array between the order of this rank's chunk -- the *particles* /
    *targets* handed to :func:`exchange_particles` -- and the order of what the rank received,
    which is what ``user_source_ids`` / ``sorted_target_ids`` of the rank's tree index.  The
    reference keeps the same map as index arrays on the root rank
    (boxtree/distributed/__init__.py:238-248 ``src_idx`` / ``tgt_idx``) and uses it to hand out
    source weights and to collect potentials (distributed/calculation.py:86-142).  Valid until
    the next exchange on the context; every method is collective over the ranks."""

    def __init__(self, actx, comm, shard, n_sources, n_targets):
        self.actx, self.comm = actx, comm
        have_targets = n_targets is not None
        self.n = {"sources": int(n_sources), "targets": int(n_targets) if have_targets else None}
        self.n_owned = {"sources": int(shard.n_owned),
                        "targets": int(shard.n_owned_targets) if have_targets else None}
        self.chunk_offset = {"sources": int(shard.source_chunk_offset),
                             "targets": int(shard.target_chunk_offset) if have_targets else None}
        self.n_global = {"sources": int(shard.n_global_sources),
                         "targets": int(shard.n_global_targets) if have_targets else None}
        # particles of the chunk that live on another rank now: what one routed array moves
        self.n_sent = {"sources": int(shard.n_sent_sources),
                       "targets": int(shard.n_sent_targets) if have_targets else None}
        # the library keeps ONE plan per context (that of its latest exchange) and takes no array
        # lengths: a route that outlives its exchange would move the new plan's counts through
        # buffers sized for the old one
        self.serial = getattr(actx, "_mgpu_exchange_serial", None)

    def _check_current(self):
        if getattr(self.actx, "_mgpu_exchange_serial", None) != self.serial:
            raise RuntimeError("ParticleRoute: the context has run another exchange since this route "
                               "was made; the library keeps the plan of the latest exchange only")

    def _set(self, which):
        if which not in ("sources", "targets"):
            raise ValueError("which must be 'sources' or 'targets'")
        if self.n_owned[which] is None:
            raise ValueError("no separate targets were exchanged")
        return 0 if which == "sources" else 1

    def _route(self, array, which, direction, n_in, n_out):
        import torch
        if array.dim() != 1 or array.element_size() not in (4, 8):
            raise TypeError("ParticleRoute: 1-D arrays of 4- or 8-byte elements")
        if len(array) != n_in:
            raise ValueError(f"ParticleRoute: {len(array)} values for {n_in} {which}")
        self._check_current()
        a = array.contiguous()
        out = torch.empty(n_out, dtype=a.dtype, device=a.device)
        self.actx.sync_in()
        _lib.check(self.actx.lib.bt_mgpu_route(
            self.actx.handle, self.comm.handle, self._set(which), direction, a.element_size(),
            ct.c_void_p(a.data_ptr()), ct.c_void_p(out.data_ptr())))
        return out

    def to_owners(self, array, which="sources"):
        """*array* ``[n]`` in the order of this rank's chunk -> ``[n_owned]`` in the order of the
        received particles."""
        self._set(which)
        return self._route(array, which, _lib.BT_ROUTE_TO_OWNERS, self.n[which], self.n_owned[which])

    def to_callers(self, array, which="sources"):
        """The inverse: ``[n_owned]`` in received order -> ``[n]`` in the chunk's order."""
        self._set(which)
        return self._route(array, which, _lib.BT_ROUTE_TO_CALLERS, self.n_owned[which], self.n[which])

    def global_ids(self, which="sources", dtype=None):
        """Global user id of every received particle (int32 like the reference's
        ``particle_id_t``; ``torch.int64`` on request): ``chunk_offset`` of the sending rank +
        index in its chunk, i.e. the index in the concatenation of the ranks' chunks."""
        import torch
        iset = self._set(which)
        self._check_current()
        if dtype is None:       # the reference's particle_id_t while it can name every particle
            dtype = torch.int32 if self.n_global[which] <= 2**31 - 1 else torch.int64
        out = torch.empty(self.n_owned[which], dtype=dtype, device=f"cuda:{self.actx.device_index}")
        self.actx.sync_in()
        _lib.check(self.actx.lib.bt_mgpu_global_ids(
            self.actx.handle, self.comm.handle, iset, out.element_size(),
            ct.c_void_p(out.data_ptr())))
        return out

    # the two index arrays of the single-GPU Tree (tree.py:426-438), this rank's share

    def global_user_source_ids(self, tree, dtype=None):
        """``user_source_ids`` of the global tree for this rank's sources: entry ``j`` is entry
        ``numbering["source_offset"] + j`` of the array one GPU builds from the concatenated
        chunks.  *dtype* as in :meth:`global_ids` (int32 while the global count fits, else int64)."""
        return self.global_ids("sources", dtype)[tree.user_source_ids.long()]

    def global_sorted_target_ids(self, tree, target_offset):
        """``sorted_target_ids`` of the global tree for the targets of this rank's CHUNK, in the
        chunk's order: where each of them sits in the global tree's target order
        (*target_offset*: ``numbering["target_offset"]``)."""
        import torch
        which = "targets" if self.n_owned["targets"] is not None else "sources"
        pos = tree.sorted_target_ids
        if self.n_global[which] > 2**31 - 1:        # positions beyond particle_id_t: 8-byte ids
            pos = pos.to(torch.int64)
        return self.to_callers(pos + int(target_offset), which)


def exchange_particles(actx, comm, particles, max_particles_in_box, top_level=None,
                       own_buffer=False, targets=None, target_radii=None, stick_out_factor=None,
                       extent_norm="linf", refine_weights=None, target_refine_weights=None,
                       max_leaf_refine_weight=None):
    """Steps 1-3 (``bt_mgpu_exchange``).  Returns ``(particles, build_kw, stats)`` for the
    local ``TreeBuilder`` call: views of the interleaved receive buffer, and ``_root_box`` /
    ``_top_tree`` / ``_point_stride``.  The receive buffer belongs to the context and is
    valid until its next exchange (the tree build copies what it keeps); with
    *own_buffer* it is a torch allocation the returned views keep alive.  With separate
    point *targets* the return value is ``(particles, targets, build_kw, stats)``: both sets
    travel to the owners of their cells (cells are counted over sources and targets), and
    come back as contiguous arrays.  With *target_radii* (every rank passes them, an empty
    chunk included) the return value is ``(particles, targets, target_radii, build_kw,
    stats)``: a target that sticks out of the boxes of the shared top levels stays in one of
    them and travels to the owner of that box's first cell; ``build_kw`` then carries
    ``stick_out_factor`` / ``extent_norm`` and the top of the global tree as arrival / stay
    counts per top box.  With *max_leaf_refine_weight* (the same on every rank; then
    *max_particles_in_box* is ignored) boxes split by refine weight: *refine_weights* /
    *target_refine_weights* (int32, None = ones) travel with the particles and come back in
    ``build_kw`` (``refine_weights``: sources then targets, ``max_leaf_refine_weight``)."""
    import torch
    dims = len(particles)
    dev = particles[0].device
    dtype = particles[0].dtype
    es = particles[0].element_size()
    par = _lib.MgpuParams()
    par.dims = dims
    par.coord_kind = _lib.BT_F64 if dtype == torch.float64 else _lib.BT_F32
    par.n = len(particles[0])
    keep = [p.contiguous() for p in particles]
    for ax in range(dims):
        par.coords[ax] = keep[ax].data_ptr()
    par.top_level = int(top_level or 0)
    par.max_particles_in_box = int(max_particles_in_box or 0)
    keep_w = []
    if max_leaf_refine_weight is not None:
        par.max_leaf_refine_weight = int(max_leaf_refine_weight)
        for name, w in (("source_refine_weights", refine_weights), ("target_refine_weights", target_refine_weights)):
            if w is not None:
                assert w.dtype == torch.int32
                keep_w.append(w.contiguous())
                setattr(par, name, keep_w[-1].data_ptr())
    elif refine_weights is not None or target_refine_weights is not None:
        raise ValueError("refine weights need max_leaf_refine_weight")
    if targets is not None:
        keep_t = [t.contiguous() for t in targets]
        par.ntargets = len(keep_t[0])
        for ax in range(dims):
            par.targets[ax] = keep_t[ax].data_ptr()
        if target_radii is not None:
            if stick_out_factor is None:
                raise ValueError("stick_out_factor must be given with target_radii")
            keep_r = target_radii.contiguous()
            par.target_radii = keep_r.data_ptr()
            par.stick_out_factor = float(stick_out_factor)
            par.extent_norm = _lib.NORMS[extent_norm]
    elif target_radii is not None:
        raise ValueError("target_radii without targets")
    got = {}
    bufs = []

    def alloc(_user, nbytes):
        got["buf"] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        bufs.append(got["buf"])
        return got["buf"].data_ptr()

    cb = _lib.ALLOC_FN(alloc)
    if own_buffer:
        par.alloc = ct.cast(cb, ct.c_void_p)
    shard = _lib.MgpuShard()
    actx.sync_in()
    _lib.check(actx.lib.bt_mgpu_exchange(actx.handle, comm.handle, ct.byref(par), ct.byref(shard)))
    n_owned = int(shard.n_owned)
    if targets is not None:
        res = _exchanged_with_targets(actx, shard, bufs if own_buffer else [], dims, dtype, es, dev,
                                      max_leaf_refine_weight)
        res[-1]["route"] = ParticleRoute(actx, comm, shard, par.n, par.ntargets)
        if target_radii is None:
            return res
        p2, t2, kw, stats = res
        kw["stick_out_factor"] = stick_out_factor
        kw["extent_norm"] = extent_norm
        return p2, t2[:dims], t2[dims], kw, stats
    if shard.sep_targets:
        raise ValueError("exchange_particles: other ranks passed separate targets; every rank "
                         "must pass `targets` (an empty chunk is fine)")
    slen = int(shard.source_record_len) or dims
    if own_buffer:
        recv = got["buf"][:n_owned * slen * es].view(dtype).view(n_owned, slen)
    else:
        recv = torch.as_tensor(
            _DevicePointer(shard.points, (max(n_owned, 1) * slen,), "<f8" if es == 8 else "<f4"),
            device=dev)[:n_owned * slen].view(n_owned, slen)
    new_particles = [recv[:, ax] for ax in range(dims)]
    coord = np.dtype(np.float64 if dtype == torch.float64 else np.float32)
    bbox_min = np.array(shard.bbox_min[:dims], dtype=coord)
    bbox_max = np.array(shard.bbox_max[:dims], dtype=coord)
    root_extent = coord.type(shard.root_extent)
    build_kw = {"_root_box": (bbox_min, bbox_max, root_extent)}
    if slen > 1:
        build_kw["_point_stride"] = slen
    else:
        new_particles = [p.contiguous() for p in new_particles]
    k = int(shard.top_level)
    if shard.top_cell_prefix:
        prefix = torch.as_tensor(
            _DevicePointer(shard.top_cell_prefix, ((1 << (dims * k)) + 1,), "<i8"), device=dev)
        build_kw["_top_tree"] = (k, prefix) + _top_tables(shard, dims, k, dev)
    _weights_kw(build_kw, shard, n_owned, 0, max_leaf_refine_weight, dev)
    stats = dict(bytes_sent=int(shard.bytes_sent), rounds=int(shard.rounds), top_level=k,
                 a2a_ms=_a2a_time(actx, shard),
                 bbox_min=bbox_min, bbox_max=bbox_max, root_extent=root_extent,
                 planned=bool(shard.top_cell_prefix), recv_buffer=got.get("buf"),
                 route=ParticleRoute(actx, comm, shard, par.n, None))
    return new_particles, build_kw, stats


def _top_tables(shard, dims, k, dev):
    """(arrive, stay) per box of levels 0..k as device tensors, or () for point particles with
    unit weights (``bt_tree_params.top_box_arrive`` / ``top_box_stay``)."""
    import torch
    if not shard.top_box_arrive:
        return ()
    C = 1 << dims
    ntop = (C ** (k + 1) - 1) // (C - 1)
    return tuple(torch.as_tensor(_DevicePointer(ptr_, (ntop,), "<i8"), device=dev)
                 for ptr_ in (shard.top_box_arrive, shard.top_box_stay))


def _weights_kw(build_kw, shard, nsources, ntargets, max_leaf_refine_weight, dev):
    import torch
    if shard.refine_weights:
        n = nsources + ntargets
        build_kw["refine_weights"] = torch.as_tensor(
            _DevicePointer(shard.refine_weights, (max(n, 1),), "<i4"), device=dev)[:n]
        build_kw["max_leaf_refine_weight"] = int(max_leaf_refine_weight)


def _exchanged_with_targets(actx, shard, bufs, dims, dtype, es, dev, max_leaf_refine_weight=None):
    """Views of both received sets (interleaved records, read in place by the tree build:
    ``_point_stride`` / ``_target_stride``; radii as a dense array) + build kwargs.  The buffers
    are the context's (valid until its next exchange) or, with *own_buffer*, torch allocations
    in the order the library asked for them: sources, targets, radii."""
    import torch
    ts = "<f8" if es == 8 else "<f4"
    ns, nt = int(shard.n_owned), int(shard.n_owned_targets)
    nv = int(shard.target_record_len) or dims

    def view(ptr_, buf, n, width):
        if buf is not None:
            flat = buf[:n * width * es].view(dtype)
        else:
            flat = torch.as_tensor(_DevicePointer(ptr_, (max(n, 1) * width,), ts), device=dev)[:n * width]
        return flat.view(n, width)

    own = len(bufs) > 0
    slen = int(shard.source_record_len) or dims
    src = view(shard.points, bufs[0] if own else None, ns, slen)
    tbuf = bufs[1] if own and len(bufs) > 1 else None
    if shard.target_points:
        tgt = view(shard.target_points, tbuf, nt, nv)
    else:       # no rank of the job had a target: the library exchanged one set
        tgt = torch.empty((0, nv), dtype=dtype, device=dev)
    out = [[src[:, ax] for ax in range(dims)], [tgt[:, ax] for ax in range(dims)]]     # (radius / weight columns: below)
    if shard.target_radii:
        rbuf = bufs[2] if own and len(bufs) > 2 else None
        out[1].append(view(shard.target_radii, rbuf, nt, 1)[:, 0])
    coord = np.dtype(np.float64 if dtype == torch.float64 else np.float32)
    bbox_min = np.array(shard.bbox_min[:dims], dtype=coord)
    bbox_max = np.array(shard.bbox_max[:dims], dtype=coord)
    root_extent = coord.type(shard.root_extent)
    build_kw = {"_root_box": (bbox_min, bbox_max, root_extent)}
    if slen > 1:
        build_kw["_point_stride"] = slen
    else:
        out[0] = [p.contiguous() for p in out[0]]
    if nv > 1:
        build_kw["_target_stride"] = nv
    else:
        out[1] = [t.contiguous() for t in out[1]]
    k = int(shard.top_level)
    if shard.top_cell_prefix:
        prefix = torch.as_tensor(
            _DevicePointer(shard.top_cell_prefix, ((1 << (dims * k)) + 1,), "<i8"), device=dev)
        build_kw["_top_tree"] = (k, prefix) + _top_tables(shard, dims, k, dev)
    _weights_kw(build_kw, shard, ns, nt, max_leaf_refine_weight, dev)
    stats = dict(bytes_sent=int(shard.bytes_sent), rounds=int(shard.rounds), top_level=k,
                 a2a_ms=_a2a_time(actx, shard), bbox_min=bbox_min, bbox_max=bbox_max,
                 root_extent=root_extent, planned=bool(shard.top_cell_prefix))
    return out[0], out[1], build_kw, stats


def _local_tree_view(actx, tree):
    import torch

    from boxtree_amd.tree import level_start_box_nrs_of
    lsb = np.ascontiguousarray(level_start_box_nrs_of(actx, tree), dtype=np.int32)
    v = _lib.MgpuLocalTree()
    v.dims = int(tree.dimensions)
    v.coord_kind = _lib.BT_F64 if tree.box_centers.dtype == torch.float64 else _lib.BT_F32
    v.nboxes = int(tree.nboxes)
    v.aligned_nboxes = int(tree.box_centers.shape[1])
    v.nlevels = int(tree.nlevels)
    v.level_start_box_nrs = lsb.ctypes.data_as(ct.POINTER(ct.c_int32))
    v.box_centers = tree.box_centers.data_ptr()
    v.box_levels = tree.box_levels.data_ptr()
    v.box_flags = tree.box_flags.data_ptr()
    v.nsources = int(tree.nsources)
    v.ntargets = int(tree.ntargets)
    if getattr(tree, "targets_have_extent", False):
        v.box_target_bounding_box_min = tree.box_target_bounding_box_min.data_ptr()
        v.box_target_bounding_box_max = tree.box_target_bounding_box_max.data_ptr()
        v.box_source_counts_cumul = tree.box_source_counts_cumul.data_ptr()
    # (what TreeBuilder kept of its export: the LET then comes with subtree sizes, if every rank's
    # tree has them)
    sizes = getattr(tree, "_subtree_sizes", None)
    if sizes is not None and os.environ.get("BOXTREE_HIP_SUBTREE_SIZES", "1") != "0":
        v.box_subtree_sizes = sizes.data_ptr()
    return v, lsb


def number_sharded_tree(actx, comm, tree):
    """Step 5 (``bt_mgpu_number``): global box numbers of the rank's boxes (int32 device
    tensor), global level starts and particle offsets."""
    import torch
    view, _keep = _local_tree_view(actx, tree)
    box_ids = torch.empty(int(tree.nboxes), dtype=torch.int32, device=tree.box_centers.device)
    num = _lib.MgpuNumbering()
    actx.sync_in()
    _lib.check(actx.lib.bt_mgpu_number(actx.handle, comm.handle, ct.byref(view),
                                       ct.c_void_p(box_ids.data_ptr()), ct.byref(num)))
    nl = int(num.nlevels)
    return dict(box_ids=box_ids, struct=num, nlevels=nl,
                global_level_start_box_nrs=np.array(num.level_start_box_nrs[:nl + 1], dtype=np.int64),
                deep_base=np.array(num.deep_base[:nl], dtype=np.int64),
                nboxes=int(num.nboxes), nsources=int(num.nsources), ntargets=int(num.ntargets),
                source_offset=int(num.source_offset), target_offset=int(num.target_offset))


def build_local_essential_tree(actx, comm, tree, numbering, well_sep_is_n_away=1):
    """Step 6 (``bt_mgpu_let_build`` / ``bt_mgpu_let_export``).  Returns ``(let, info)`` like
    :func:`boxtree_amd.distributed.build_local_essential_tree`."""
    import torch

    from boxtree_amd.tree import TreeOfBoxes
    view, _keep = _local_tree_view(actx, tree)
    sizes = _lib.MgpuLetSizes()
    actx.sync_in()
    _lib.check(actx.lib.bt_mgpu_let_build(
        actx.handle, comm.handle, ct.byref(view), ct.c_void_p(numbering["box_ids"].data_ptr()),
        ct.byref(numbering["struct"]), int(well_sep_is_n_away), ct.byref(sizes)))
    B, aligned, nlev = int(sizes.nboxes), int(sizes.aligned_nboxes), int(sizes.nlevels)
    dims = int(tree.dimensions)
    C = 1 << dims
    coord_dtype = np.dtype(np.float64 if tree.box_centers.dtype == torch.float64 else np.float32)
    centers, parents, children, levels, flags, gids, mask = actx.empty_block([
        ((dims, aligned), coord_dtype), (B, np.int32), ((C, aligned), np.int32), (B, np.uint8),
        (B, np.uint8), (B, np.int32), (B, np.int8)])
    arrs = _lib.MgpuLetArrays()
    ext = bool(getattr(tree, "targets_have_extent", False))
    if ext:
        tbb_min, tbb_max, srccum = actx.empty_block([
            ((dims, aligned), coord_dtype), ((dims, aligned), coord_dtype), (B, np.int32)])
        arrs.box_target_bounding_box_min = tbb_min.data_ptr()
        arrs.box_target_bounding_box_max = tbb_max.data_ptr()
        arrs.box_source_counts_cumul = srccum.data_ptr()
    arrs.box_centers = centers.data_ptr()
    arrs.box_parent_ids = parents.data_ptr()
    arrs.box_child_ids = children.data_ptr()
    arrs.box_levels = levels.data_ptr()
    arrs.box_flags = flags.data_ptr()
    arrs.global_box_ids = gids.data_ptr()
    arrs.target_boxes_mask = mask.data_ptr()
    let_sizes = None
    if int(sizes.has_subtree_sizes):
        let_sizes = torch.empty(B, dtype=torch.int32, device=tree.box_centers.device)
        arrs.box_subtree_sizes = let_sizes.data_ptr()
    _lib.check(actx.lib.bt_mgpu_let_export(actx.handle, ct.byref(arrs)))
    lsb = np.array(sizes.level_start_box_nrs[:nlev + 1], dtype=np.int32)
    let = TreeOfBoxes(
        root_extent=tree.root_extent, box_centers=centers, box_parent_ids=parents,
        box_child_ids=children, box_levels=levels, box_flags=flags, level_start_box_nrs=lsb,
        box_id_dtype=np.dtype(np.int32), box_level_dtype=np.dtype(np.uint8),
        coord_dtype=coord_dtype, sources_have_extent=False, targets_have_extent=ext,
        extent_norm=tree.extent_norm if ext else None, stick_out_factor=tree.stick_out_factor,
        _is_pruned=True)
    if ext:
        # what the traversal reads of a tree whose targets have extents
        object.__setattr__(let, "box_target_bounding_box_min", tbb_min)
        object.__setattr__(let, "box_target_bounding_box_max", tbb_max)
        object.__setattr__(let, "box_source_counts_cumul", srccum)
        object.__setattr__(let, "sources_are_targets", False)
    # made by the library on this device: contiguous arrays of its own types -- the traversal
    # builder takes them as they are (no conversions, no look at the root's parent entry)
    object.__setattr__(let, "_host_level_starts", lsb)
    if let_sizes is not None:
        # (bt_trav_params.box_subtree_sizes: the traversal's depth-first ranks start from them)
        object.__setattr__(let, "_subtree_sizes", let_sizes)
    ranges = np.array([[sizes.active_level_ranges[l][0], sizes.active_level_ranges[l][1]]
                       for l in range(nlev)], dtype=np.int32)
    info = dict(target_boxes_mask=mask, active_level_ranges=ranges, global_box_ids=gids,
                halo_boxes_received=int(sizes.halo_boxes_received),
                halo_boxes_sent=int(sizes.halo_boxes_sent), nboxes=B,
                loopback_records=int(sizes.loopback_records),
                loopback_mismatches=int(sizes.loopback_mismatches))
    if ext:
        info.update(box_target_bounding_box_min=tbb_min, box_target_bounding_box_max=tbb_max,
                    box_source_counts_cumul=srccum)
    return let, info


def sharded_tree_and_lists(actx, comm, particles, max_particles_in_box=None, targets=None,
                           well_sep_is_n_away=1, tree_builder=None, traversal_builder=None,
                           target_radii=None, stick_out_factor=None, extent_norm="linf",
                           refine_weights=None, target_refine_weights=None,
                           max_leaf_refine_weight=None):
    """Steps 1-6 in one call (collective over *comm*, a :class:`NativeComm`): exchange the
    particles, build the subtrees this rank owns, number them globally, assemble the local
    essential tree and build the interaction lists of the rank's own boxes.  The keyword
    arguments are ``TreeBuilder.__call__``'s (tree_build.py:145-260), per rank: separate
    *targets*, *target_radii* with *stick_out_factor* / *extent_norm*, *refine_weights* (sources;
    *target_refine_weights* for separate targets) with *max_leaf_refine_weight* instead of
    *max_particles_in_box*.

    Returns a dict: ``tree`` (the rank's :class:`~boxtree_amd.tree.Tree`, local box
    numbers), ``numbering`` (:func:`number_sharded_tree`: ``box_ids`` maps them to the
    global tree's), ``let`` / ``let_info`` (:func:`build_local_essential_tree`:
    ``let_info["global_box_ids"]`` maps LET boxes to global numbers), ``traversal`` (lists
    on the LET for the boxes with ``let_info["target_boxes_mask"]``), ``exchange`` (bytes
    sent, device time of the all-to-all-v, root box) and ``route`` (:class:`ParticleRoute`:
    per-particle data between the caller's order and the tree's, global user ids)."""
    from boxtree_amd import FMMTraversalBuilder, TreeBuilder
    tb = tree_builder or TreeBuilder(actx)
    tg = traversal_builder or FMMTraversalBuilder(actx, well_sep_is_n_away=well_sep_is_n_away)
    xkw = dict(refine_weights=refine_weights, target_refine_weights=target_refine_weights,
               max_leaf_refine_weight=max_leaf_refine_weight)
    bkw = {} if max_leaf_refine_weight is not None else dict(max_particles_in_box=max_particles_in_box)
    if targets is None:
        p2, kw, xs = exchange_particles(actx, comm, particles, max_particles_in_box, **xkw)
        tree, _ = tb(actx, p2, **bkw, **kw)
    elif target_radii is None:
        p2, t2, kw, xs = exchange_particles(actx, comm, particles, max_particles_in_box, targets=targets, **xkw)
        tree, _ = tb(actx, p2, targets=t2, **bkw, **kw)
    else:
        p2, t2, r2, kw, xs = exchange_particles(
            actx, comm, particles, max_particles_in_box, targets=targets, target_radii=target_radii,
            stick_out_factor=stick_out_factor, extent_norm=extent_norm, **xkw)
        tree, _ = tb(actx, p2, targets=t2, target_radii=r2, **bkw, **kw)
    num = number_sharded_tree(actx, comm, tree)
    let, info = build_local_essential_tree(actx, comm, tree, num, well_sep_is_n_away=well_sep_is_n_away)
    trav, _ = tg(actx, let, _target_boxes_mask=info["target_boxes_mask"],
                 _active_level_ranges=info["active_level_ranges"])
    return dict(tree=tree, numbering=num, let=let, let_info=info, traversal=trav, exchange=xs,
                route=xs["route"])
