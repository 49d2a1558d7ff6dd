"""TEST INFRASTRUCTURE -- the Python call surface over tests/emu/_build/libboxtree_emu.so: the
kernels of boxtree_amd/csrc compiled for the host against the HIP emulator (tests/emu/hip_emu.hpp).
"Device" arrays are CPU torch tensors.  For CPU-side checks of the kernels' LOGIC against the
oracle where no GPU is at hand; the product (boxtree_amd.HIPArrayContext) has no CPU path and never
sees this module.

    from emu_actx import EmuArrayContext          # builds the emulated library on first use
    actx = EmuArrayContext()
    tree, _ = TreeBuilder(actx)(actx, [actx.from_numpy(x), ...], max_particles_in_box=30)
"""

import ctypes as ct
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(HERE, "_build", "libboxtree_emu.so")


def build():
    subprocess.check_call(["make", "-C", HERE, "-j8"], stdout=subprocess.DEVNULL)


def install():
    """Make boxtree_amd's ctypes layer talk to the emulated library in THIS process (before the
    first boxtree_amd call; a process uses either the product library or the emulated one)."""
    from boxtree_amd import _lib
    if _lib._lib is not None:
        if getattr(_lib._lib, "_is_emu", False):
            return _lib._lib
        raise RuntimeError("boxtree_amd has already loaded the product library in this process")
    build()
    lib = _lib._bind(ct.CDLL(EMU_LIB))
    lib._is_emu = True
    _lib._lib = lib
    return lib


def _make_class():
    from boxtree_amd.array_context import HIPArrayContext

    class EmuArrayContext(HIPArrayContext):
        """A HIPArrayContext whose device is the emulator: CPU tensors, synchronous "stream"."""

        def __init__(self):
            import torch
            from boxtree_amd import _lib
            self.torch = torch
            self.device_index = 0
            self.device = torch.device("cpu")
            self.lib = install()
            handle = ct.c_void_p()
            _lib.check(self.lib.bt_create(0, None, ct.byref(handle)))
            self.handle = handle
            self._stream_handle = None
            self.stream_ordered = False
            _lib.check(self.lib.bt_set_stream_ordered(self.handle, 0))

        @property
        def stream(self):
            return None

        def sync_in(self):
            return None

    return EmuArrayContext


def EmuArrayContext():          # noqa: N802  (a factory named like the class it returns an instance of)
    return _make_class()()


# ---- running the -m gpu tests under emulation (BOXTREE_EMU=1) ----------------------------------------

def _cpu_device(dev):
    import torch
    if dev is None:
        return None
    if isinstance(dev, str):
        return "cpu" if dev.startswith("cuda") else dev
    if isinstance(dev, torch.device):
        return torch.device("cpu") if dev.type == "cuda" else dev
    if isinstance(dev, int):
        return "cpu"
    return dev


def patch_torch():
    """Process-wide: every request for a "cuda" tensor yields a CPU tensor, and the torch.cuda calls
    the tests and the Python layer make become no-ops.  (Plain monkeypatches rather than a
    TorchFunctionMode: the rank threads of the multi-rank tests must see them too.)"""
    import ctypes as ct_
    import numpy as np
    import torch
    if getattr(torch, "_boxtree_emu_patched", False):
        return
    torch._boxtree_emu_patched = True

    def wrap_factory(fn):
        def inner(*a, **k):
            if "device" in k:
                k["device"] = _cpu_device(k["device"])
            if "generator" in k and k["generator"] is not None:
                k["generator"] = getattr(k["generator"], "_gen", k["generator"])
            return fn(*a, **k)
        inner.__name__ = getattr(fn, "__name__", "factory")
        return inner

    for name in ("empty", "zeros", "ones", "full", "arange", "rand", "randn", "randint", "tensor",
                 "empty_like", "zeros_like", "ones_like", "full_like", "linspace", "randperm", "eye"):
        setattr(torch, name, wrap_factory(getattr(torch, name)))

    # uninitialised "device" arrays come back poisoned (0xA5 bytes), as the emulator's hipMalloc does:
    # an output array a kernel does not write completely must not pass because the host allocator
    # happened to hand out zeros (EMU_POISON=0 switches it off)
    if os.environ.get("EMU_POISON", "1") != "0":
        def poisoned(fn):
            def inner(*a, **k):
                t = fn(*a, **k)
                if t.numel() and t.is_contiguous():
                    t.view(torch.uint8).fill_(0xA5)
                return t
            return inner
        torch.empty = poisoned(torch.empty)
        torch.empty_like = poisoned(torch.empty_like)

    real_as_tensor = torch.as_tensor

    def as_tensor(obj, *a, **k):
        if "device" in k:
            k["device"] = _cpu_device(k["device"])
        cai = getattr(obj, "__cuda_array_interface__", None)
        if cai is not None and not isinstance(obj, torch.Tensor):
            # a raw "device" pointer handed over by the library: host memory under emulation
            shape, typestr, ptr = tuple(cai["shape"]), cai["typestr"], int(cai["data"][0])
            dt = np.dtype(typestr)
            count = int(np.prod(shape)) if len(shape) else 1
            if count == 0:
                return torch.from_numpy(np.zeros(shape, dt))
            buf = (ct_.c_char * (count * dt.itemsize)).from_address(ptr)
            return torch.from_numpy(np.frombuffer(buf, dtype=dt, count=count).reshape(shape))
        return real_as_tensor(obj, *a, **k)
    torch.as_tensor = as_tensor

    real_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(_cpu_device(x) if isinstance(x, (str, torch.device)) else x for x in a)
        if "device" in k:
            k["device"] = _cpu_device(k["device"])
        return real_to(self, *a, **k)
    torch.Tensor.to = to
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)

    class Gen:                      # torch.Generator(device="cuda")
        def __init__(self, device=None):
            self._gen = real_generator()

        def manual_seed(self, seed):
            self._gen.manual_seed(seed)
            return self
    real_generator = torch.Generator
    torch.Generator = Gen

    class _Stream:
        cuda_stream = 0
    cu = torch.cuda
    cu.is_available = lambda: True
    cu.synchronize = lambda *a, **k: None
    cu.set_device = lambda *a, **k: None
    cu.current_device = lambda: 0
    cu.device_count = lambda: 1
    cu.empty_cache = lambda: None
    def mem_get_info(*a, **k):
        # what the tests size themselves by: host memory stands in for device memory (EMU_MEM_GB overrides)
        gb = os.environ.get("EMU_MEM_GB")
        if gb:
            return int(float(gb) * 2**30), int(float(gb) * 2**30)
        try:
            import psutil
            vm = psutil.virtual_memory()
            return int(vm.available), int(vm.total)
        except ImportError:
            return 8 << 30, 16 << 30
    cu.mem_get_info = mem_get_info
    cu.current_stream = lambda *a, **k: _Stream()


RCCL_STUB = os.path.join(HERE, "_build", "librccl.so.1")


def patch_rccl(lib):
    """A one-rank stand-in for RCCL (emu_rccl.cpp, SONAME librccl.so.1): whoever loads torch's
    librccl.so through ctypes gets it, and so does the library when it is told which image to bind
    (bt_mgpu_use_rccl_library) or looks for a loaded one by SONAME."""
    real_cdll = ct.CDLL

    class CDLL(real_cdll):
        def __init__(self, name, *a, **k):
            if isinstance(name, str) and os.path.basename(name).startswith(("librccl.so", "libamdhip64.so")):
                name = RCCL_STUB
            super().__init__(name, *a, **k)
    ct.CDLL = CDLL
    # (torch has the real RCCL in the process already, under the same SONAME: name the stand-in's
    # image now, before the library binds its entry points, and keep later calls from naming another)
    use = lib.bt_mgpu_use_rccl_library
    use(RCCL_STUB.encode())
    lib.bt_mgpu_use_rccl_library = lambda path: use(RCCL_STUB.encode())


def install_for_tests():
    """conftest.py calls this when BOXTREE_EMU=1: the emulated library, CPU tensors for "cuda", and
    boxtree_amd.HIPArrayContext constructing the emulator's context."""
    lib = install()
    patch_torch()
    patch_rccl(lib)
    from boxtree_amd import array_context
    emu_cls = _make_class()
    base = array_context.HIPArrayContext

    def init(self, device=None):
        emu_cls.__init__(self)
    base.__init__ = init
    base.stream = property(lambda self: None)
    base.sync_in = lambda self: None
