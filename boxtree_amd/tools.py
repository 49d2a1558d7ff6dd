"""Small helpers mirroring boxtree/tools.py that the hot path and its callers need: event
stand-ins, ``padded_bin``, and the reference's particle test fixtures (tools.py:114-283)."""

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


def surface_particle_coords(nparticles, dims, dtype):
    """Host arrays of the reference's "surface" fixture (tools.py:122-186): in 2D ``nparticles``
    points on a closed curve with three lobes, in 3D the ``n x n`` lattice (``n =
    int(sqrt(nparticles))``) of a torus of radii 15 and 5 -- deterministic, no random numbers."""
    dtype = np.dtype(dtype)
    two_pi = dtype.type(2 * np.pi)
    if dims == 2:
        phi = two_pi / dtype.type(nparticles) * np.arange(nparticles, dtype=dtype)
        return [(0.5 * (3 * np.cos(phi) + 2 * np.sin(3 * phi))).astype(dtype),
                (0.5 * (np.sin(phi) + 1.5 * np.sin(2 * phi))).astype(dtype)]
    if dims == 3:
        n = int(nparticles ** 0.5)
        ang = two_pi / dtype.type(n) * np.arange(n, dtype=dtype)
        phi, theta = np.meshgrid(ang, ang, indexing="ij")          # [i, j]
        ring = 3 + np.cos(theta)
        return [(5 * np.cos(phi) * ring).ravel().astype(dtype),
                (5 * np.sin(phi) * ring).ravel().astype(dtype),
                (5 * np.sin(theta)).ravel().astype(dtype)]
    raise NotImplementedError


def uniform_particle_coords(nparticles, dims, dtype):
    """Host arrays of the reference's "uniform" fixture (tools.py:189-276): a regular lattice of
    ``n^dims`` points (``n = int(nparticles^(1/dims))``) on a square / cube of side 4, turned by
    0.3 rad (and, in 3D, by 0.7 rad about the second axis) so that no lattice plane is parallel
    to a box face, shifted by -2."""
    dtype = np.dtype(dtype)
    s1, c1 = np.sin(0.3), np.cos(0.3)
    if dims == 2:
        n = int(nparticles ** 0.5)
        t = (4 * np.arange(n, dtype=np.float64) / (n - 1))
        xx, yy = np.meshgrid(t, t, indexing="ij")
        return [(c1 * xx + s1 * yy - 2).ravel().astype(dtype), (-s1 * xx + c1 * yy - 2).ravel().astype(dtype)]
    if dims == 3:
        n = int(nparticles ** (1 / 3))
        t = np.arange(n, dtype=np.float64) / (n - 1)
        xx, yy, zz = np.meshgrid(t, t, t, indexing="ij")
        x1, y1 = c1 * xx + s1 * yy, -s1 * xx + c1 * yy
        s2, c2 = np.sin(0.7), np.cos(0.7)
        return [(4 * (c2 * x1 + s2 * zz) - 2).ravel().astype(dtype), (4 * y1 - 2).ravel().astype(dtype),
                (4 * (-s2 * x1 + c2 * zz) - 2).ravel().astype(dtype)]
    raise NotImplementedError


def make_surface_particle_array(actx, nparticles, dims, dtype, seed=15):
    """Device arrays of :func:`surface_particle_coords` (*seed* is unused, as upstream)."""
    from boxtree_amd.array_context import make_obj_array
    return make_obj_array([actx.from_numpy(a) for a in surface_particle_coords(nparticles, dims, dtype)])


def make_uniform_particle_array(actx, nparticles, dims, dtype, seed=15):
    """Device arrays of :func:`uniform_particle_coords` (*seed* is unused, as upstream)."""
    from boxtree_amd.array_context import make_obj_array
    return make_obj_array([actx.from_numpy(a) for a in uniform_particle_coords(nparticles, dims, dtype)])


def make_rotated_uniform_particle_array(actx, nparticles, dims, dtype, seed=15):
    raise NotImplementedError       # (as upstream, tools.py:279-280)


def particle_array_to_host(actx, particles):
    """``[nparticles, dims]`` host array of an object array of coordinate arrays (tools.py:285-286)."""
    return np.array([actx.to_numpy(x) for x in particles], order="F").T
