"""Minimal stand-in for the reference's array context (boxtree/array_context.py:63-132).

The reference threads a PyOpenCL array context through every call; here the
"context" owns a HIP device, a stream and the library handle.  Device arrays
are ``torch`` tensors on that device -- torch is used for device memory and
stream plumbing only.
"""

from __future__ import annotations

import ctypes as ct
import dataclasses
import os

import numpy as np

from boxtree_amd import _lib


class HIPArrayContext:
    """Owns one ``bt_context`` (one HIP device + stream).

    Counterpart of ``PyOpenCLArrayContext``: ``from_numpy``/``to_numpy`` move
    arrays and whole containers (``Tree``, ``FMMTraversalInfo``), ``freeze`` /
    ``thaw`` are identities (outputs are immutable by convention).
    """

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError(
                "boxtree_amd needs a HIP device (torch.cuda.is_available() is False); "
                "there is no CPU fallback")
        self.torch = torch
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.lib = _lib.load()
        handle = ct.c_void_p()
        cur = torch.cuda.current_stream(self.device)
        _lib.check(self.lib.bt_create(self.device_index, ct.c_void_p(cur.cuda_stream),
                                      ct.byref(handle)))
        self.handle = handle
        # The library runs on torch's current stream (bt_create takes NULL as "make your
        # own", so the legacy default stream, handle 0, is set explicitly): inputs made by
        # torch ops, the library's kernels, torch ops on the outputs and the caching
        # allocator's reuse of freed blocks are all ordered by that one stream.
        self._stream_handle = None
        self.sync_in()
        # Results are stream-ordered, like every torch op and like the reference, whose
        # builders return (object, event): the builders return when the host knows the
        # result's sizes, the last kernels may still be filling the arrays.
        self.stream_ordered = os.environ.get("BOXTREE_HIP_STREAM_ORDERED", "1") != "0"
        _lib.check(self.lib.bt_set_stream_ordered(self.handle, int(self.stream_ordered)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.bt_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- stream ordering -------------------------------------------------------
    @property
    def stream(self):
        """The torch stream the library's kernels are queued on."""
        return self.torch.cuda.current_stream(self.device)

    def sync_in(self):
        """Called before every library call: point the library at torch's current stream
        (a no-op unless the caller switched streams since the last call)."""
        h = self.torch.cuda.current_stream(self.device).cuda_stream
        if h != self._stream_handle:
            _lib.check(self.lib.bt_set_stream(self.handle, ct.c_void_p(h)))
            self._stream_handle = h

    def set_stage_timing(self, on):
        """Per-stage HIP events of the builders (``TreeBuilder.last_stage_times``); on by
        default, a few microseconds of stream bubble per stage."""
        _lib.check(self.lib.bt_set_stage_timing(self.handle, int(bool(on))))

    def synchronize(self):
        """Wait for everything queued by the library; raises if a call that returned early
        failed on the device."""
        _lib.check(self.lib.bt_synchronize(self.handle))

    # -- array movement ----------------------------------------------------------
    def from_numpy(self, ary):
        torch = self.torch
        if isinstance(ary, np.ndarray) and ary.dtype.char == "O":
            out = np.empty(ary.shape, dtype=object)
            for i, a in np.ndenumerate(ary):
                out[i] = self.from_numpy(a)
            return out
        if isinstance(ary, torch.Tensor):
            return ary.to(self.device)
        return torch.from_numpy(np.ascontiguousarray(ary)).to(self.device)

    def to_numpy(self, obj):
        """Host copy of an array or a whole container.  Results are stream-ordered and a
        builder may have returned before its last kernels ran: the first host read waits
        for the library's stream and raises if the device reported a failure after the
        call returned (callers that drop the builders' events still see it here)."""
        self.synchronize()
        return self._to_numpy(obj)

    def _to_numpy(self, obj):
        torch = self.torch
        if obj is None or isinstance(obj, (int, float, str, bool, np.generic, np.dtype)):
            return obj
        if isinstance(obj, torch.Tensor):
            return obj.detach().cpu().numpy()
        if isinstance(obj, np.ndarray):
            if obj.dtype.char == "O":
                out = np.empty(obj.shape, dtype=object)
                for i, a in np.ndenumerate(obj):
                    out[i] = self._to_numpy(a)
                return out
            return obj
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._to_numpy(o) for o in obj)
        if dataclasses.is_dataclass(obj):
            return obj._map_arrays(self._to_numpy)
        return obj

    def freeze(self, obj):
        return obj

    def thaw(self, obj):
        return obj

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=_torch_dtype(self.torch, dtype), device=self.device)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=_torch_dtype(self.torch, dtype), device=self.device)

    def empty_block(self, specs):
        """Uninitialised arrays ``[(shape, dtype), ...]`` carved out of ONE allocation
        (256-byte aligned views): the outputs of a call cost one trip to the caching
        allocator instead of one per array, and the views of one element type are made by
        a single ``split_with_sizes`` (a Python-level slice + view per array costs several
        microseconds; a tree has 15 output arrays, a traversal 50-70)."""
        torch = self.torch
        # (index, shape, element count padded to 256 bytes) per element type
        groups = {}
        for i, (shape, dtype) in enumerate(specs):
            if isinstance(shape, tuple):
                count = 1
                for extent in shape:
                    count *= int(extent)
            else:
                count = int(shape)
            tdtype, itemsize = _DTYPES[np.dtype(dtype)]
            per256 = 256 // itemsize
            groups.setdefault((tdtype, itemsize), []).append(
                (i, shape, count, -(-count // per256) * per256 - count))
        total = sum((c + pad) * key[1] for key, items in groups.items() for _, _, c, pad in items)
        block = torch.empty(max(total, 256), dtype=torch.uint8, device=self.device)
        out = [None] * len(specs)
        off = 0
        for (tdtype, itemsize), items in groups.items():
            nbytes = sum(c + pad for _, _, c, pad in items) * itemsize
            typed = block[off:off + nbytes].view(tdtype)
            off += nbytes
            sizes = []
            for _, _, c, pad in items:
                sizes.append(c)
                sizes.append(pad)
            parts = typed.split_with_sizes(sizes)
            for k, (i, shape, _, _) in enumerate(items):
                v = parts[2 * k]
                out[i] = v.view(shape) if isinstance(shape, tuple) else v
        return out


def _dtype_table():
    import torch
    return {
        np.dtype(np.float32): (torch.float32, 4), np.dtype(np.float64): (torch.float64, 8),
        np.dtype(np.int32): (torch.int32, 4), np.dtype(np.int64): (torch.int64, 8),
        np.dtype(np.uint8): (torch.uint8, 1), np.dtype(np.int8): (torch.int8, 1),
        np.dtype(np.uint32): (torch.int32, 4), np.dtype(np.uint64): (torch.int64, 8),
    }


_DTYPES = _dtype_table()      # numpy dtype -> (torch dtype, item size)


def _torch_dtype(torch, dtype):
    return _DTYPES[np.dtype(dtype)][0]


def as_device_array(actx, a):
    """Device tensor for *a*: numpy arrays are uploaded, torch tensors pass through,
    anything exposing ``__cuda_array_interface__`` (CuPy, Numba, ... device arrays)
    is wrapped without a copy."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return actx.from_numpy(a)
    torch = actx.torch
    if not isinstance(a, torch.Tensor) and hasattr(a, "__cuda_array_interface__"):
        return torch.as_tensor(a, device=actx.device)
    return a


def np_dtype_of(t):
    """numpy dtype of a torch tensor / numpy array."""
    if isinstance(t, np.ndarray):
        return t.dtype
    import torch
    return {torch.float32: np.dtype(np.float32), torch.float64: np.dtype(np.float64),
            torch.int32: np.dtype(np.int32), torch.int64: np.dtype(np.int64),
            torch.uint8: np.dtype(np.uint8), torch.int8: np.dtype(np.int8)}[t.dtype]


def ptr(t):
    """Device address of a tensor as a ctypes void pointer (NULL for None)."""
    if t is None:
        return ct.c_void_p(None)
    assert t.is_contiguous()
    return ct.c_void_p(t.data_ptr())


def make_obj_array(arrays):
    out = np.empty(len(arrays), dtype=object)
    for i, a in enumerate(arrays):
        out[i] = a
    return out
