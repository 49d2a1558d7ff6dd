"""Small helpers mirroring boxtree/tools.py that the hot path needs."""

from __future__ import annotations

import numpy as np

AXIS_NAMES = ("x", "y", "z", "w")


class DoneEvent:
    """Stand-in for ``pyopencl.Event`` of a call that has completed on the host."""

    def wait(self):
        return None


class StreamEvent:
    """Stand-in for ``pyopencl.Event`` of a builder call on a stream-ordered context
    (boxtree_hip.h bt_set_stream_ordered): the arrays of the result are valid for work
    queued on the context's stream; ``wait()`` blocks until they are complete and raises
    if the device reported a failure after the call had returned."""

    def __init__(self, actx):
        self._actx = actx

    def wait(self):
        self._actx.synchronize()
        return None


def padded_bin(i, nbits):
    """Format *i* as binary number, pad it to length *nbits* (tools.py:50-52)."""
    return bin(i)[2:].rjust(nbits, "0")


def make_normal_particle_array(actx, nparticles, dims, dtype, seed=15):
    """Test fixture with the reference's recipe (tools.py:114-119)."""
    from boxtree_amd.array_context import make_obj_array
    rng = np.random.default_rng(seed)
    return make_obj_array([
        actx.from_numpy(rng.standard_normal(nparticles, dtype=dtype))
        for _ in range(dims)])
