"""Bounding box of a particle set on the GPU (boxtree/bounding_box.py:163-174)."""

from __future__ import annotations

import ctypes as ct

import numpy as np

from boxtree_amd import _lib
from boxtree_amd.array_context import np_dtype_of, ptr

AXIS_NAMES = ("x", "y", "z", "w")


def make_bounding_box_dtype(dimensions, coord_dtype):
    """Structured dtype {min_x, max_x, min_y, ...} (bounding_box.py:35-51)."""
    fields = []
    for i in range(dimensions):
        fields.append((f"min_{AXIS_NAMES[i]}", coord_dtype))
        fields.append((f"max_{AXIS_NAMES[i]}", coord_dtype))
    return np.dtype(fields)


class BoundingBoxFinder:
    def __init__(self, array_context):
        self._setup_actx = array_context

    def __call__(self, actx, particles, radii, wait_for=None):
        """Returns ``(bbox, event)``; *bbox* is a 0-d structured numpy array
        with fields ``min_x, max_x, ...`` of the coordinate dtype."""
        dimensions = len(particles)
        coord_dtype = np_dtype_of(particles[0])
        kind = _lib.BT_F64 if coord_dtype == np.float64 else _lib.BT_F32
        n = len(particles[0])
        arr = (ct.c_void_p * dimensions)(*[ptr(p).value for p in particles])
        mn = (ct.c_double * _lib.BT_MAX_DIMS)()
        mx = (ct.c_double * _lib.BT_MAX_DIMS)()
        actx.sync_in()
        _lib.check(actx.lib.bt_bbox(actx.handle, dimensions, kind, arr, ptr(radii),
                                    n, mn, mx))
        bbox = np.empty((), make_bounding_box_dtype(dimensions, coord_dtype))
        for i in range(dimensions):
            bbox[f"min_{AXIS_NAMES[i]}"] = mn[i]
            bbox[f"max_{AXIS_NAMES[i]}"] = mx[i]
        from boxtree_amd.tools import DoneEvent
        return bbox, DoneEvent()
