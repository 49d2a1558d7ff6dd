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
