"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Python front-end of the CPU oracle (``oracle/boxtree_oracle.c``): a restatement
of inducer/boxtree's ``TreeBuilder.__call__`` (boxtree/tree_build.py:145-1878)
and ``FMMTraversalBuilder.__call__`` (boxtree/traversal.py:1969-2345).

PARITY UNPINNED for the kernels: the reference's OpenCL path cannot be imported
here (pyopencl, arraycontext, pytools, mako are absent, no OpenCL device) and its
tests hold no golden vectors; this restatement is validated against the invariants
the reference's tests assert (tests/test_oracle_invariants.py).  The plain-Python
parts of the reference DO run here: tests/golden/make_reference_vectors.py runs
its drive_fmm + constant-one wrangler on this oracle's trees and lists, and its
depth-first order / work partition / communication pattern / rotation-class code,
and tests/test_reference_vectors.py holds the restatements below to those outputs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The host-side numpy arithmetic (root box, argument
checks) below mirrors tree_build.py:223-510 line by line and is deliberately NOT
shared with the product package.
"""

from __future__ import annotations

import ctypes as ct
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAXDIM = 3
AXIS_NAMES = ("x", "y", "z")
ROOT_EXTENT_STRETCH_FACTOR = 1e-4      # tree_build.py:101


class MaxLevelsExceeded(RuntimeError):   # tree_build.py:79
    pass


_VARIANT = "seq"      # "seq": liboracle.so, "omp": liboracle_omp.so (same source, -fopenmp)


def set_variant(variant, num_threads=None):
    """Choose the sequential oracle or its OpenMP build (bench.py's all-core CPU
    baseline; identical results, tests/test_oracle_openmp.py).  Returns the number of
    threads the loops will use."""
    global _VARIANT, _LIB
    assert variant in ("seq", "omp")
    if variant != _VARIANT:
        _VARIANT = variant
        _LIB = None
    lib = _lib()
    lib.orc_set_num_threads(int(num_threads or 0))
    return int(lib.orc_max_threads())


def build_lib(force=False):
    """Compile liboracle.so / liboracle_omp.so with gcc (a few seconds)."""
    name = "liboracle.so" if _VARIANT == "seq" else "liboracle_omp.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in (
        "boxtree_oracle.c", "boxtree_oracle_impl.h", "boxtree_oracle_trav_impl.h",
        "boxtree_oracle_aq_impl.h")]
    if (force or not os.path.exists(so)
            or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)):
        subprocess.check_call(["make", "-C", _HERE, "-s", name, "-B"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ct.CDLL(build_lib())
        _LIB.orc_free.argtypes = [ct.c_void_p]
        _LIB.orc_free.restype = None
        _LIB.orc_set_num_threads.argtypes = [ct.c_int]
        _LIB.orc_max_threads.restype = ct.c_int
    return _LIB


def _ctype(dtype):
    return ct.c_double if np.dtype(dtype) == np.float64 else ct.c_float


def _sfx(dtype):
    return "_f64" if np.dtype(dtype) == np.float64 else "_f32"


def _make_structs(cf):
    P = ct.POINTER

    class TreeIn(ct.Structure):
        _fields_ = [
            ("dims", ct.c_int32), ("sources_are_targets", ct.c_int32),
            ("nsrcntgts", ct.c_int64), ("nsources", ct.c_int64),
            ("srcntgts", P(cf) * MAXDIM), ("srcntgt_radii", P(cf)),
            ("sources_have_extent", ct.c_int32), ("targets_have_extent", ct.c_int32),
            ("refine_weights", P(ct.c_int32)), ("max_leaf_refine_weight", ct.c_int32),
            ("bbox_min", cf * MAXDIM), ("bbox_max", cf * MAXDIM),
            ("root_extent", cf), ("stick_out_factor", cf),
            ("extent_norm", ct.c_int32), ("kind", ct.c_int32),
            ("skip_prune", ct.c_int32), ("nlevels_max", ct.c_int32),
        ]

    class TreeOut(ct.Structure):
        _fields_ = [
            ("status", ct.c_int32), ("nlevels", ct.c_int32),
            ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
            ("level_start_box_nrs", P(ct.c_int32)),
            ("user_source_ids", P(ct.c_int32)), ("sorted_target_ids", P(ct.c_int32)),
            ("sources", P(cf) * MAXDIM), ("targets", P(cf) * MAXDIM),
            ("source_radii", P(cf)), ("target_radii", P(cf)),
            ("box_source_starts", P(ct.c_int32)),
            ("box_source_counts_nonchild", P(ct.c_int32)),
            ("box_source_counts_cumul", P(ct.c_int32)),
            ("box_target_starts", P(ct.c_int32)),
            ("box_target_counts_nonchild", P(ct.c_int32)),
            ("box_target_counts_cumul", P(ct.c_int32)),
            ("box_parent_ids", P(ct.c_int32)), ("box_child_ids", P(ct.c_int32)),
            ("box_centers", P(cf)), ("box_levels", P(ct.c_uint8)),
            ("box_flags", P(ct.c_uint8)),
            ("box_source_bounding_box_min", P(cf)), ("box_source_bounding_box_max", P(cf)),
            ("box_target_bounding_box_min", P(cf)), ("box_target_bounding_box_max", P(cf)),
        ]

    class BuiltListC(ct.Structure):
        _fields_ = [
            ("n_objects", ct.c_int64), ("count", ct.c_int64),
            ("starts", P(ct.c_int32)), ("lists", P(ct.c_int32)),
            ("num_nonempty_lists", ct.c_int64),
            ("nonempty_indices", P(ct.c_int32)), ("compressed_indices", P(ct.c_int32)),
        ]

    class TravIn(ct.Structure):
        _fields_ = [
            ("dims", ct.c_int32), ("nlevels", ct.c_int32),
            ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
            ("root_extent", cf),
            ("box_centers", P(cf)), ("box_levels", P(ct.c_uint8)),
            ("box_child_ids", P(ct.c_int32)), ("box_flags", P(ct.c_uint8)),
            ("box_parent_ids", P(ct.c_int32)), ("level_start_box_nrs", P(ct.c_int32)),
            ("sources_are_targets", ct.c_int32),
            ("sources_have_extent", ct.c_int32), ("targets_have_extent", ct.c_int32),
            ("stick_out_factor", cf),
            ("box_target_bounding_box_min", P(cf)), ("box_target_bounding_box_max", P(cf)),
            ("box_source_counts_cumul", P(ct.c_int32)),
            ("well_sep_is_n_away", ct.c_int32), ("from_sep_smaller_crit", ct.c_int32),
            ("from_sep_smaller_min_nsources_cumul", ct.c_int32),
            ("source_boxes_mask", P(ct.c_int8)), ("source_parent_boxes_mask", P(ct.c_int8)),
        ]

    class TravOut(ct.Structure):
        _fields_ = [
            ("status", ct.c_int32),
            ("nsource_boxes", ct.c_int64), ("ntarget_boxes", ct.c_int64),
            ("nsource_parent_boxes", ct.c_int64),
            ("ntarget_or_target_parent_boxes", ct.c_int64),
            ("source_boxes", P(ct.c_int32)), ("target_boxes", P(ct.c_int32)),
            ("source_parent_boxes", P(ct.c_int32)),
            ("target_or_target_parent_boxes", P(ct.c_int32)),
            ("level_start_source_box_nrs", P(ct.c_int32)),
            ("level_start_target_box_nrs", P(ct.c_int32)),
            ("level_start_source_parent_box_nrs", P(ct.c_int32)),
            ("level_start_target_or_target_parent_box_nrs", P(ct.c_int32)),
            ("same_level_non_well_sep_boxes", BuiltListC),
            ("neighbor_source_boxes", BuiltListC),
            ("from_sep_siblings", BuiltListC),
            ("from_sep_smaller_by_level", P(BuiltListC)),
            ("from_sep_close_smaller", BuiltListC),
            ("from_sep_bigger", BuiltListC),
            ("from_sep_close_bigger", BuiltListC),
        ]

    class AqTree(ct.Structure):
        _fields_ = [
            ("dims", ct.c_int32), ("nlevels", ct.c_int32),
            ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
            ("root_extent", cf), ("bbox_min", cf * MAXDIM),
            ("box_centers", P(cf)), ("box_levels", P(ct.c_uint8)),
            ("box_child_ids", P(ct.c_int32)), ("box_flags", P(ct.c_uint8)),
        ]

    return SimpleNamespace(TreeIn=TreeIn, TreeOut=TreeOut, BuiltListC=BuiltListC,
                           TravIn=TravIn, TravOut=TravOut, AqTree=AqTree)


_STRUCTS = {}


def _structs(dtype):
    key = np.dtype(dtype).str
    if key not in _STRUCTS:
        _STRUCTS[key] = _make_structs(_ctype(dtype))
    return _STRUCTS[key]


def _take(ptr, n, dtype, free=True):
    """Copy n elements out of a malloc'd C array and free it."""
    n = int(n)
    if not ptr:
        return None
    if n == 0:
        arr = np.zeros(0, dtype)
    else:
        arr = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    if free:
        _lib().orc_free(ct.cast(ptr, ct.c_void_p))
    return arr


def _ptr(arr, ctype):
    return arr.ctypes.data_as(ct.POINTER(ctype))


# {{{ bounding box

def bounding_box(particles, radii=None):
    """boxtree/bounding_box.py:163-174 -> (mins, maxs) arrays of coord dtype."""
    dims = len(particles)
    dtype = particles[0].dtype
    cf = _ctype(dtype)
    particles = [np.ascontiguousarray(p) for p in particles]
    n = len(particles[0])
    arr = (ct.POINTER(cf) * dims)(*[_ptr(p, cf) for p in particles])
    mn = np.empty(dims, dtype)
    mx = np.empty(dims, dtype)
    fn = getattr(_lib(), "orc_bbox" + _sfx(dtype))
    fn.restype = None
    r = None if radii is None else _ptr(np.ascontiguousarray(radii), cf)
    fn(ct.c_int(dims), ct.c_int64(n), arr, r, _ptr(mn, cf), _ptr(mx, cf))
    return mn, mx

# }}}


# {{{ tree build

_KINDS = {"adaptive": 0, "adaptive-level-restricted": 1, "non-adaptive": 2}
_NORMS = {None: 0, "linf": 1, "l2": 2}


def build_tree(particles, kind="adaptive", max_particles_in_box=None,
               targets=None, source_radii=None, target_radii=None,
               stick_out_factor=None, refine_weights=None,
               max_leaf_refine_weight=None, extent_norm=None, bbox=None,
               **kwargs):
    """Restatement of TreeBuilder.__call__ (tree_build.py:145-1878).

    Returns a namespace with the attribute names of :class:`boxtree.Tree`
    (tree.py:298-686), all numpy.
    """
    # {{{ input processing: tree_build.py:223-295
    if kind not in ["adaptive", "adaptive-level-restricted", "non-adaptive"]:
        raise ValueError(f"unknown tree kind: '{kind}'")

    dimensions = len(particles)
    axis_names = AXIS_NAMES[:dimensions]
    sources_are_targets = targets is None
    sources_have_extent = source_radii is not None
    targets_have_extent = target_radii is not None

    if extent_norm is None:
        extent_norm = "linf"
    if extent_norm not in ["linf", "l2"]:
        raise ValueError(f"unexpected value of 'extent_norm': {extent_norm}")
    srcntgts_extent_norm = extent_norm
    srcntgts_have_extent = sources_have_extent or targets_have_extent
    if not srcntgts_have_extent:
        srcntgts_extent_norm = None
    if srcntgts_extent_norm and targets is None:
        raise ValueError("must specify targets when specifying any kind of radii")

    particles = [np.ascontiguousarray(p) for p in particles]
    coord_dtype = particles[0].dtype
    assert all(p.dtype == coord_dtype for p in particles)
    if targets is None:
        nsrcntgts = nsources = len(particles[0])
        ntargets = nsources
    else:
        targets = [np.ascontiguousarray(t) for t in targets]
        nsources = len(particles[0])
        ntargets = len(targets[0])
        nsrcntgts = nsources + ntargets

    if source_radii is not None:
        if source_radii.shape != (nsources,):
            raise ValueError("'source_radii' has an invalid shape")
        if source_radii.dtype != coord_dtype:
            raise TypeError("dtypes of 'particles' and 'source_radii' must agree")
    if target_radii is not None:
        if target_radii.shape != (ntargets,):
            raise ValueError("'target_radii' has an invalid shape")
        if target_radii.dtype != coord_dtype:
            raise TypeError("dtypes of 'particles' and 'target_radii' must agree")

    if sources_have_extent or targets_have_extent:
        if stick_out_factor is None:
            raise ValueError("if sources or targets have extent, "
                             "'stick_out_factor' must be explicitly specified")
    else:
        stick_out_factor = 0
    # }}}

    # {{{ combine sources and targets: tree_build.py:328-388
    if targets is None:
        srcntgts = particles
        srcntgt_radii = None
    else:
        if targets[0].dtype != coord_dtype:
            raise TypeError("sources and targets coordinates must have same dtype")

        def combine(a1, a2):
            result = np.zeros(nsrcntgts, coord_dtype)
            if a1 is not None:
                result[:len(a1)] = a1
            if a2 is not None:
                result[nsources:] = a2
            return result

        srcntgts = [combine(s, t) for s, t in zip(particles, targets)]
        srcntgt_radii = (combine(source_radii, target_radii)
                         if srcntgts_have_extent else None)
    # }}}

    # {{{ refine weights: tree_build.py:405-454
    specified_mpb = max_particles_in_box is not None
    specified_rw = refine_weights is not None and max_leaf_refine_weight is not None
    if specified_mpb and specified_rw:
        raise ValueError("may only specify one of 'max_particles_in_box' and "
                         "'refine_weights'/'max_leaf_refine_weight")
    elif not specified_mpb and not specified_rw:
        raise ValueError("must specify either 'max_particles_in_box' or "
                         "'refine_weights'/'max_leaf_refine_weight'")
    elif specified_mpb:
        refine_weights = np.ones(nsrcntgts, np.int32)
        max_leaf_refine_weight = max_particles_in_box
    else:
        if refine_weights.dtype != np.int32:
            raise TypeError("'refine_weights' must have dtype 'int32'")
    if max_leaf_refine_weight <= 0:
        raise ValueError("'max_leaf_refine_weight' must be positive")
    if nsrcntgts and max_leaf_refine_weight < np.max(refine_weights):
        raise ValueError(
            "entries of 'refine_weights' cannot exceed 'max_leaf_refine_weight'")
    if nsrcntgts and np.min(refine_weights) < 0:
        raise ValueError("all entries of 'refine_weights' must be nonnegative")
    refine_weights = np.ascontiguousarray(refine_weights)
    # }}}

    # {{{ bounding box: tree_build.py:456-510
    auto_min, auto_max = bounding_box(srcntgts, srcntgt_radii)
    if bbox is None:
        root_extent = max(
            auto_max[i] - auto_min[i] for i in range(dimensions)
        ) * (1 + ROOT_EXTENT_STRETCH_FACTOR)
        bbox_min = np.empty(dimensions, coord_dtype)
        for i in range(dimensions):
            bbox_min[i] = auto_min[i]
        bbox_max = bbox_min + root_extent
        # the bbox struct stores coord_dtype values (:475-476)
        bbox_struct_min = bbox_min.copy()
        bbox_struct_max = bbox_max.astype(coord_dtype)
    else:
        bbox = np.asarray(bbox)
        assert len(bbox) == dimensions
        bbox_min = np.empty(dimensions, coord_dtype)
        bbox_max = np.empty(dimensions, coord_dtype)
        for i in range(dimensions):
            bbox_min[i] = bbox[i][0]
            bbox_max[i] = bbox[i][1]
            assert bbox_min[i] < bbox_max[i]
            assert bbox_min[i] <= auto_min[i]
            assert bbox_max[i] >= auto_max[i]
        bbox_exts = bbox_max - bbox_min
        for ext in bbox_exts:
            assert abs(ext - bbox_exts[0]) < 1e-15
        root_extent = bbox_exts[0]
        bbox_struct_min = bbox_min.copy()
        bbox_struct_max = bbox_max.copy()
    # }}}

    S = _structs(coord_dtype)
    cf = _ctype(coord_dtype)
    tin = S.TreeIn()
    tin.dims = dimensions
    tin.sources_are_targets = int(sources_are_targets)
    tin.nsrcntgts = nsrcntgts
    tin.nsources = nsources
    for i in range(dimensions):
        tin.srcntgts[i] = _ptr(srcntgts[i], cf)
    if srcntgt_radii is not None:
        tin.srcntgt_radii = _ptr(srcntgt_radii, cf)
    tin.sources_have_extent = int(sources_have_extent)
    tin.targets_have_extent = int(targets_have_extent)
    tin.refine_weights = _ptr(refine_weights, ct.c_int32)
    tin.max_leaf_refine_weight = int(max_leaf_refine_weight)
    for i in range(dimensions):
        tin.bbox_min[i] = bbox_struct_min[i]
        tin.bbox_max[i] = bbox_struct_max[i]
    tin.root_extent = float(coord_dtype.type(root_extent))
    tin.stick_out_factor = float(coord_dtype.type(stick_out_factor))
    tin.extent_norm = _NORMS[srcntgts_extent_norm]
    tin.kind = _KINDS[kind]
    tin.skip_prune = int(bool(kwargs.get("skip_prune")))
    tin.nlevels_max = 2 * (np.finfo(coord_dtype).nmant + 1)   # :622

    tout = S.TreeOut()
    fn = getattr(_lib(), "orc_tree_build" + _sfx(coord_dtype))
    fn.restype = ct.c_int
    status = fn(ct.byref(tin), ct.byref(tout))
    if status == 1:
        raise MaxLevelsExceeded("Level count exceeded number of significant "
                                "bits in coordinate dtype.")
    if status != 0:
        raise RuntimeError(f"oracle tree build failed with status {status}")

    nboxes = int(tout.nboxes)
    aligned = int(tout.aligned_nboxes)
    nlevels = int(tout.nlevels)
    C = 2 ** dimensions
    i32 = np.int32

    t = SimpleNamespace()
    t.sources_are_targets = sources_are_targets
    t.sources_have_extent = sources_have_extent
    t.targets_have_extent = targets_have_extent
    t.particle_id_dtype = np.dtype(np.int32)
    t.box_id_dtype = np.dtype(np.int32)
    t.coord_dtype = coord_dtype
    t.box_level_dtype = np.dtype(np.uint8)
    t.bounding_box = (bbox_min, bbox_max)
    t.root_extent = root_extent
    t.stick_out_factor = stick_out_factor
    t.extent_norm = srcntgts_extent_norm
    t.level_start_box_nrs = _take(tout.level_start_box_nrs, nlevels + 1, i32)
    t.user_source_ids = _take(tout.user_source_ids, nsources, i32)
    t.sorted_target_ids = _take(tout.sorted_target_ids, ntargets, i32)
    t.sources = [_take(tout.sources[d], nsources, coord_dtype, free=True)
                 for d in range(dimensions)]
    if sources_are_targets:
        t.targets = t.sources
    else:
        t.targets = [_take(tout.targets[d], ntargets, coord_dtype)
                     for d in range(dimensions)]
    t.source_radii = _take(tout.source_radii, nsources, coord_dtype)
    t.target_radii = _take(tout.target_radii, ntargets, coord_dtype)
    if not sources_have_extent:
        t.source_radii = None
    if not targets_have_extent:
        t.target_radii = None
    t.box_source_starts = _take(tout.box_source_starts, nboxes, i32)
    t.box_source_counts_nonchild = _take(tout.box_source_counts_nonchild, nboxes, i32)
    t.box_source_counts_cumul = _take(tout.box_source_counts_cumul, nboxes, i32)
    if sources_are_targets:
        t.box_target_starts = t.box_source_starts
        t.box_target_counts_nonchild = t.box_source_counts_nonchild
        t.box_target_counts_cumul = t.box_source_counts_cumul
    else:
        t.box_target_starts = _take(tout.box_target_starts, nboxes, i32)
        t.box_target_counts_nonchild = _take(tout.box_target_counts_nonchild, nboxes, i32)
        t.box_target_counts_cumul = _take(tout.box_target_counts_cumul, nboxes, i32)
    t.box_parent_ids = _take(tout.box_parent_ids, nboxes, i32)
    t.box_child_ids = _take(tout.box_child_ids, C * aligned, i32).reshape(C, aligned)
    t.box_centers = _take(tout.box_centers, dimensions * aligned,
                          coord_dtype).reshape(dimensions, aligned)
    t.box_levels = _take(tout.box_levels, nboxes, np.uint8)
    t.box_flags = _take(tout.box_flags, nboxes, np.uint8)
    t.box_source_bounding_box_min = _take(
        tout.box_source_bounding_box_min, dimensions * aligned,
        coord_dtype).reshape(dimensions, aligned)
    t.box_source_bounding_box_max = _take(
        tout.box_source_bounding_box_max, dimensions * aligned,
        coord_dtype).reshape(dimensions, aligned)
    if sources_are_targets:
        t.box_target_bounding_box_min = t.box_source_bounding_box_min
        t.box_target_bounding_box_max = t.box_source_bounding_box_max
    else:
        t.box_target_bounding_box_min = _take(
            tout.box_target_bounding_box_min, dimensions * aligned,
            coord_dtype).reshape(dimensions, aligned)
        t.box_target_bounding_box_max = _take(
            tout.box_target_bounding_box_max, dimensions * aligned,
            coord_dtype).reshape(dimensions, aligned)
    t._is_pruned = not kwargs.get("skip_prune")

    t.dimensions = dimensions
    t.nboxes = nboxes
    t.aligned_nboxes = aligned
    t.nlevels = nlevels
    t.nsources = nsources
    t.ntargets = ntargets
    return t

# }}}


# {{{ traversal

_CRITS = {"static_linf": 0, "precise_linf": 1, "static_l2": 2}


def _built_list(bl, free=True):
    n_objects = int(bl.n_objects)
    count = int(bl.count)
    nne = int(bl.num_nonempty_lists)
    r = SimpleNamespace()
    r.count = count
    r.lists = _take(bl.lists, count, np.int32, free)
    if nne < 0:
        r.starts = _take(bl.starts, n_objects + 1, np.int32, free)
        r.num_nonempty_lists = None
        r.nonempty_indices = None
        r.compressed_indices = None
    else:
        r.starts = _take(bl.starts, nne + 1, np.int32, free)
        r.num_nonempty_lists = nne
        r.nonempty_indices = _take(bl.nonempty_indices, nne, np.int32, free)
        r.compressed_indices = _take(bl.compressed_indices, n_objects + 1,
                                     np.int32, free)
    return r


def build_traversal(tree, well_sep_is_n_away=1, from_sep_smaller_crit=None,
                    _from_sep_smaller_min_nsources_cumul=None,
                    source_boxes_mask=None, source_parent_boxes_mask=None):
    """Restatement of FMMTraversalBuilder.__call__ (traversal.py:1969-2345)."""
    if _from_sep_smaller_min_nsources_cumul is None:
        _from_sep_smaller_min_nsources_cumul = 0          # :1995-1997
    if not tree._is_pruned:
        raise ValueError("tree must be pruned for traversal generation")
    if tree.sources_have_extent:
        raise NotImplementedError("trees with source extent are not supported "
                                  "for traversal generation")
    # :1776-1805
    if from_sep_smaller_crit is None:
        from_sep_smaller_crit = "precise_linf"
    extent_norm = tree.extent_norm
    if extent_norm == "l2" and from_sep_smaller_crit == "static_linf":
        raise ValueError("the static l^inf from-sep-smaller criterion "
                         "cannot be used with the l^2 extent norm")
    if extent_norm not in ("linf", "l2", None):
        raise ValueError(f"unexpected value of 'extent_norm': {extent_norm}")
    if from_sep_smaller_crit not in _CRITS:
        raise ValueError("unexpected value of 'from_sep_smaller_crit': "
                         f"{from_sep_smaller_crit}")

    coord_dtype = np.dtype(tree.coord_dtype)
    S = _structs(coord_dtype)
    cf = _ctype(coord_dtype)
    dims = tree.dimensions
    nlevels = int(tree.nlevels)

    keep = []

    def c(arr, dtype):
        a = np.ascontiguousarray(arr, dtype=dtype)
        keep.append(a)
        return a

    tin = S.TravIn()
    tin.dims = dims
    tin.nlevels = nlevels
    tin.nboxes = tree.nboxes
    tin.aligned_nboxes = tree.aligned_nboxes
    tin.root_extent = float(coord_dtype.type(tree.root_extent))
    tin.box_centers = _ptr(c(tree.box_centers, coord_dtype), cf)
    tin.box_levels = _ptr(c(tree.box_levels, np.uint8), ct.c_uint8)
    tin.box_child_ids = _ptr(c(tree.box_child_ids, np.int32), ct.c_int32)
    tin.box_flags = _ptr(c(tree.box_flags, np.uint8), ct.c_uint8)
    tin.box_parent_ids = _ptr(c(tree.box_parent_ids, np.int32), ct.c_int32)
    tin.level_start_box_nrs = _ptr(c(tree.level_start_box_nrs, np.int32), ct.c_int32)
    tin.sources_are_targets = int(getattr(tree, "sources_are_targets", True))
    tin.sources_have_extent = int(tree.sources_have_extent)
    tin.targets_have_extent = int(tree.targets_have_extent)
    tin.stick_out_factor = float(coord_dtype.type(tree.stick_out_factor))
    if tree.targets_have_extent:
        tin.box_target_bounding_box_min = _ptr(
            c(tree.box_target_bounding_box_min, coord_dtype), cf)
        tin.box_target_bounding_box_max = _ptr(
            c(tree.box_target_bounding_box_max, coord_dtype), cf)
        tin.box_source_counts_cumul = _ptr(
            c(tree.box_source_counts_cumul, np.int32), ct.c_int32)
    tin.well_sep_is_n_away = int(well_sep_is_n_away)
    tin.from_sep_smaller_crit = _CRITS[from_sep_smaller_crit]
    tin.from_sep_smaller_min_nsources_cumul = int(_from_sep_smaller_min_nsources_cumul)
    if source_boxes_mask is not None:
        tin.source_boxes_mask = _ptr(c(source_boxes_mask, np.int8), ct.c_int8)
    if source_parent_boxes_mask is not None:
        tin.source_parent_boxes_mask = _ptr(
            c(source_parent_boxes_mask, np.int8), ct.c_int8)

    tout = S.TravOut()
    fn = getattr(_lib(), "orc_trav_build" + _sfx(coord_dtype))
    fn.restype = ct.c_int
    status = fn(ct.byref(tin), ct.byref(tout))
    if status != 0:
        raise RuntimeError(f"oracle traversal failed with status {status}")

    i32 = np.int32
    r = SimpleNamespace()
    r.tree = tree
    r.well_sep_is_n_away = well_sep_is_n_away
    sat = bool(tin.sources_are_targets)
    r.source_boxes = _take(tout.source_boxes, tout.nsource_boxes, i32)
    if sat:
        r.target_boxes = r.source_boxes
    else:
        r.target_boxes = _take(tout.target_boxes, tout.ntarget_boxes, i32)
    r.source_parent_boxes = _take(tout.source_parent_boxes,
                                  tout.nsource_parent_boxes, i32)
    r.target_or_target_parent_boxes = _take(
        tout.target_or_target_parent_boxes,
        tout.ntarget_or_target_parent_boxes, i32)
    for name in ("level_start_source_box_nrs", "level_start_target_box_nrs",
                 "level_start_source_parent_box_nrs",
                 "level_start_target_or_target_parent_box_nrs"):
        setattr(r, name, _take(getattr(tout, name), nlevels + 1, i32))

    slnws = _built_list(tout.same_level_non_well_sep_boxes)
    r.same_level_non_well_sep_boxes_starts = slnws.starts
    r.same_level_non_well_sep_boxes_lists = slnws.lists
    l1 = _built_list(tout.neighbor_source_boxes)
    r.neighbor_source_boxes_starts = l1.starts
    r.neighbor_source_boxes_lists = l1.lists
    l2 = _built_list(tout.from_sep_siblings)
    r.from_sep_siblings_starts = l2.starts
    r.from_sep_siblings_lists = l2.lists

    r.from_sep_smaller_by_level = []
    r.target_boxes_sep_smaller_by_source_level = []
    for ilev in range(nlevels):
        bl = _built_list(tout.from_sep_smaller_by_level[ilev])
        r.from_sep_smaller_by_level.append(bl)
        r.target_boxes_sep_smaller_by_source_level.append(
            r.target_boxes[bl.nonempty_indices])         # :2211-2212
    _lib().orc_free(ct.cast(tout.from_sep_smaller_by_level, ct.c_void_p))

    with_extent = tree.sources_have_extent or tree.targets_have_extent
    if with_extent:
        cs = _built_list(tout.from_sep_close_smaller)
        r.from_sep_close_smaller_starts = cs.starts
        r.from_sep_close_smaller_lists = cs.lists
    else:
        r.from_sep_close_smaller_starts = None
        r.from_sep_close_smaller_lists = None

    l4 = _built_list(tout.from_sep_bigger)
    r.from_sep_bigger_starts = l4.starts
    r.from_sep_bigger_lists = l4.lists
    if with_extent:
        cb = _built_list(tout.from_sep_close_bigger)
        r.from_sep_close_bigger_starts = cb.starts
        r.from_sep_close_bigger_lists = cb.lists
    else:
        r.from_sep_close_bigger_starts = None
        r.from_sep_close_bigger_lists = None

    r.nboxes = tree.nboxes
    r.nlevels = nlevels
    r.ntarget_boxes = len(r.target_boxes)
    r.ntarget_or_target_parent_boxes = len(r.target_or_target_parent_boxes)
    return r

# }}}


# {{{ area queries (boxtree/area_query.py)

def _aq_tree(tree, keep):
    coord_dtype = np.dtype(tree.coord_dtype)
    S = _structs(coord_dtype)
    cf = _ctype(coord_dtype)

    def c(arr, dtype):
        a = np.ascontiguousarray(arr, dtype=dtype)
        keep.append(a)
        return a

    t = S.AqTree()
    t.dims = tree.dimensions
    t.nlevels = int(tree.nlevels)
    t.nboxes = tree.nboxes
    t.aligned_nboxes = tree.aligned_nboxes
    t.root_extent = float(coord_dtype.type(tree.root_extent))
    for d in range(tree.dimensions):
        t.bbox_min[d] = float(coord_dtype.type(tree.bounding_box[0][d]))
    t.box_centers = _ptr(c(tree.box_centers, coord_dtype), cf)
    t.box_levels = _ptr(c(tree.box_levels, np.uint8), ct.c_uint8)
    t.box_child_ids = _ptr(c(tree.box_child_ids, np.int32), ct.c_int32)
    t.box_flags = _ptr(c(tree.box_flags, np.uint8), ct.c_uint8)
    return t


def peer_lists(tree):
    """PeerListFinder.__call__ (area_query.py:1152-1192) -> PeerListLookup fields."""
    keep = []
    t = _aq_tree(tree, keep)
    S = _structs(tree.coord_dtype)
    bl = S.BuiltListC()
    fn = getattr(_lib(), "orc_peer_lists" + _sfx(tree.coord_dtype))
    fn.restype = ct.c_int
    if fn(ct.byref(t), ct.byref(bl)) != 0:
        raise RuntimeError("oracle peer list build failed")
    r = _built_list(bl)
    return SimpleNamespace(tree=tree, peer_list_starts=r.starts, peer_lists=r.lists)


def _check_balls(tree, ball_centers, ball_radii):
    # area_query.py:764-768
    dts = {np.dtype(bc.dtype) for bc in ball_centers}
    if len(dts) != 1 or dts.pop() != np.dtype(tree.coord_dtype):
        raise TypeError("ball_centers dtype must match tree.coord_dtype")
    if np.dtype(ball_radii.dtype) != np.dtype(tree.coord_dtype):
        raise TypeError("ball_radii dtype must match tree.coord_dtype")


def _aq_call(name, tree, ball_centers, ball_radii, pl, out):
    _check_balls(tree, ball_centers, ball_radii)
    if pl is None:
        pl = peer_lists(tree)
    if len(pl.peer_list_starts) != tree.nboxes + 1:                 # :781-782
        raise ValueError("size of peer lists must match with number of boxes")
    keep = []
    t = _aq_tree(tree, keep)
    cf = _ctype(tree.coord_dtype)
    bcs = [np.ascontiguousarray(bc) for bc in ball_centers]
    radii = np.ascontiguousarray(ball_radii)
    arr = (ct.POINTER(cf) * len(bcs))(*[_ptr(b, cf) for b in bcs])
    starts = np.ascontiguousarray(pl.peer_list_starts, np.int32)
    lists = np.ascontiguousarray(pl.peer_lists, np.int32)
    fn = getattr(_lib(), name + _sfx(tree.coord_dtype))
    fn.restype = ct.c_int
    st = fn(ct.byref(t), _ptr(starts, ct.c_int32), _ptr(lists, ct.c_int32),
            ct.c_int64(len(radii)), arr, _ptr(radii, cf), out)
    if st != 0:
        raise RuntimeError(f"oracle {name} failed")


def area_query(tree, ball_centers, ball_radii, peer_lists=None):
    """AreaQueryBuilder.__call__ (area_query.py:744-812)."""
    S = _structs(tree.coord_dtype)
    bl = S.BuiltListC()
    _aq_call("orc_area_query", tree, ball_centers, ball_radii, peer_lists, ct.byref(bl))
    r = _built_list(bl)
    return SimpleNamespace(tree=tree, leaves_near_ball_starts=r.starts,
                           leaves_near_ball_lists=r.lists)


def leaves_to_balls(tree, ball_centers, ball_radii, peer_lists=None):
    """LeavesToBallsLookupBuilder.__call__ (area_query.py:847-924): expand the
    starts into ball numbers, stable key-value sort by box number."""
    aq = area_query(tree, ball_centers, ball_radii, peer_lists)
    nballs = len(ball_radii)
    expanded = np.repeat(np.arange(nballs, dtype=np.int32),
                         np.diff(aq.leaves_near_ball_starts))
    order = np.argsort(aq.leaves_near_ball_lists, kind="stable")
    lists = expanded[order].astype(np.int32)
    counts = np.bincount(aq.leaves_near_ball_lists, minlength=tree.nboxes)
    starts = np.zeros(tree.nboxes + 1, np.int32)
    starts[1:] = np.cumsum(counts)
    return SimpleNamespace(tree=tree, balls_near_box_starts=starts,
                           balls_near_box_lists=lists)


def space_invader_query(tree, ball_centers, ball_radii, peer_lists=None):
    """SpaceInvaderQueryBuilder.__call__ (area_query.py:970-1056): float32
    maxima, cast to the coordinate dtype at the end."""
    out = np.zeros(tree.nboxes, np.float32)
    _aq_call("orc_space_invader", tree, ball_centers, ball_radii, peer_lists,
             _ptr(out, ct.c_float))
    return out.astype(tree.coord_dtype)

# }}}

# {{{ target filtering and point-source linking (boxtree/tree.py:772-949, :1059-1243)

def filter_target_lists_in_user_order(tree, flags):
    """ParticleListFilter.filter_target_lists_in_user_order (tree.py:1115-1150;
    generator :1085-1113)."""
    flags = np.asarray(flags)
    ntargets = tree.ntargets
    user_target_ids = np.zeros(ntargets, tree.sorted_target_ids.dtype)          # :1126-1129
    user_target_ids[tree.sorted_target_ids] = np.arange(ntargets, dtype=user_target_ids.dtype)
    starts = np.zeros(tree.nboxes + 1, np.int32)
    lists = []
    for i in range(tree.nboxes):                                                # :1093-1106
        b_t_start = int(tree.box_target_starts[i])
        b_t_count = int(tree.box_target_counts_nonchild[i])
        ids = user_target_ids[b_t_start:b_t_start + b_t_count]
        ids = ids[flags[ids] != 0]
        lists.append(ids)
        starts[i + 1] = starts[i] + len(ids)
    lists = (np.concatenate(lists) if lists else np.zeros(0)).astype(np.int32)
    return SimpleNamespace(nfiltered_targets=len(lists), target_starts=starts,
                           target_lists=lists)


def filter_target_lists_in_tree_order(tree, flags):
    """ParticleListFilter.filter_target_lists_in_tree_order (tree.py:1175-1241);
    TREE_ORDER_TARGET_FILTER_SCAN_TPL / _INDEX_TPL (tree_build_kernels.py:1951-2021)."""
    flags = np.asarray(flags)
    ntargets = tree.ntargets
    tree_order_flags = np.zeros(ntargets, np.int8)                               # :1184-1185
    tree_order_flags[tree.sorted_target_ids] = flags
    ones = (tree_order_flags != 0).astype(np.int64)
    item = np.cumsum(ones)
    prev_item = item - ones
    filtered_from_unfiltered = prev_item.astype(np.int32)                       # tbk:1966
    nfiltered = int(item[-1]) if ntargets else 0
    unfiltered_from_filtered = np.nonzero(ones)[0].astype(np.int32)             # tbk:1967-1968
    targets = [np.asarray(t)[unfiltered_from_filtered] for t in tree.targets]   # :1203-1206
    nboxes = tree.nboxes
    bstarts = np.zeros(nboxes, np.int32)
    bcounts = np.zeros(nboxes, np.int32)
    for i in range(nboxes):                                                      # tbk:1990-2018
        ustart = int(tree.box_target_starts[i])
        ucount = int(tree.box_target_counts_nonchild[i])
        # the reference reads filtered_from_unfiltered[ustart] unguarded; an empty
        # box that starts at ntargets would read past the end -- take nfiltered there
        fstart = int(filtered_from_unfiltered[ustart]) if ustart < ntargets else nfiltered
        bstarts[i] = fstart
        if ucount > 0:
            upost = ustart + ucount
            fpost = int(filtered_from_unfiltered[upost]) if upost < ntargets else nfiltered
            bcounts[i] = fpost - fstart
    return SimpleNamespace(nfiltered_targets=nfiltered, box_target_starts=bstarts,
                           box_target_counts_nonchild=bcounts, targets=targets,
                           unfiltered_from_filtered_target_indices=unfiltered_from_filtered)


def link_point_sources(tree, point_source_starts, point_sources):
    """link_point_sources (tree.py:772-949); kernels tree_build_kernels.py:1871-1947.
    Sources without point sources write nothing (the reference's multi_put of their
    start index collides with the next source's / runs past the end)."""
    if not tree.sources_have_extent:
        raise ValueError("only allowed on trees whose sources have extent")
    pss = np.asarray(point_source_starts).astype(np.int64)
    usi = np.asarray(tree.user_source_ids).astype(np.int64)
    lengths = pss[usi + 1] - pss[usi]                                            # tbk:1884-1887
    item = np.cumsum(lengths)
    prev_item = item - lengths
    to_starts = prev_item.astype(np.int32)                                       # tbk:1891
    to_counts = lengths.astype(np.int32)
    npoint_sources = int(item[-1]) if len(item) else 0
    ids = np.ones(npoint_sources, np.int64)                                      # :846-853
    boundaries = np.zeros(npoint_sources, np.int8)
    nz = lengths > 0
    ids[prev_item[nz]] = pss[usi][nz]
    boundaries[prev_item[nz]] = 1                                                # :860-869
    # segmented inclusive scan, tbk:1903-1912
    seg = np.cumsum(boundaries) - 1
    seg_first = np.nonzero(boundaries)[0]
    csum = np.cumsum(ids)
    base = csum[seg_first] - ids[seg_first]
    user_point_source_ids = (csum - base[seg]).astype(np.int32) if npoint_sources else \
        np.zeros(0, np.int32)
    # cl_array.take indexes the flat storage (test/test_tree.py:638-646 passes
    # [nsources, npoint_sources_per_source] arrays)
    tree_order_point_sources = [np.asarray(ps).reshape(-1)[user_point_source_ids]
                                for ps in point_sources]
    nboxes = tree.nboxes
    bstarts = np.zeros(nboxes, np.int32)
    out = {"nonchild": np.zeros(nboxes, np.int32), "cumul": np.zeros(nboxes, np.int32)}
    src_counts = {"nonchild": tree.box_source_counts_nonchild,
                  "cumul": tree.box_source_counts_cumul}
    for ibox in range(nboxes):                                                   # tbk:1914-1947
        s_start = int(tree.box_source_starts[ibox])
        ps_start = int(to_starts[s_start]) if s_start < len(to_starts) else npoint_sources
        bstarts[ibox] = ps_start
        for kind in ("nonchild", "cumul"):
            s_count = int(src_counts[kind][ibox])
            if s_count:
                last = s_start + s_count - 1
                out[kind][ibox] = int(to_starts[last]) + int(to_counts[last]) - ps_start
    return SimpleNamespace(
        npoint_sources=npoint_sources, point_source_starts=to_starts,
        point_source_counts=to_counts, point_sources=tree_order_point_sources,
        user_point_source_ids=user_point_source_ids, box_point_source_starts=bstarts,
        box_point_source_counts_nonchild=out["nonchild"],
        box_point_source_counts_cumul=out["cumul"])

# }}}

# {{{ translation / rotation classes (boxtree/translation_classes.py, rotation_classes.py)

def translation_classes(tree, trav, is_translation_per_level=True):
    """TranslationClassesBuilder.__call__ (translation_classes.py:380-442; kernel
    :62-189).  -> (classes per list-2 entry, distance vectors [d, nclasses_used],
    level starts)."""
    nway = trav.well_sep_is_n_away
    dims = tree.dimensions
    coord = np.dtype(tree.coord_dtype).type
    per_level = (4 * nway + 3) ** dims
    nclasses = per_level * (tree.nlevels if is_translation_per_level else 1)
    lists, starts = trav.from_sep_siblings_lists, trav.from_sep_siblings_starts
    ttp = trav.target_or_target_parent_boxes
    raw = np.zeros(len(lists), np.int32)
    used = np.zeros(nclasses, np.int32)
    for itgt, tgt in enumerate(ttp):
        for i in range(starts[itgt], starts[itgt + 1]):
            src = lists[i]
            level = int(tree.box_levels[src])
            if level != int(tree.box_levels[tgt]):
                raise ValueError("could not compute translation classes")
            diam = coord(2) * (coord(tree.root_extent) * coord(1) / coord(1 << (level + 1)))
            result, mult = 0, 1
            for d in range(dims):
                v = int(np.rint((tree.box_centers[d, tgt] - tree.box_centers[d, src]) / diam))
                if not -(2 * nway + 1) <= v <= 2 * nway + 1:
                    raise ValueError("could not compute translation classes")
                result += (2 * nway + 1 + v) * mult
                mult *= 4 * nway + 3
            if is_translation_per_level:
                result += level * per_level
            raw[i] = result
            used[result] = 1
    dense = np.full(nclasses, -1, np.int32)
    dist = []
    level_starts = np.zeros(tree.nlevels + 1, np.int32)
    visited = np.zeros(tree.nlevels + 1, bool)
    count = 0
    for cls in range(nclasses):
        level, c = divmod(cls, per_level)
        if not visited[level]:
            level_starts[level] = count
            visited[level] = True
        if not used[cls]:
            continue
        dense[cls] = count
        vec = np.zeros(dims, np.int32)
        for d in range(dims):
            vec[d] = c % (4 * nway + 3) - (2 * nway + 1)
            c //= 4 * nway + 3
        dist.append(vec * tree.root_extent / (1 << level))
        count += 1
    level_starts[tree.nlevels] = count
    level_starts[:tree.nlevels][~visited[:tree.nlevels]] = count
    distances = (np.array(dist, dtype=tree.coord_dtype).T if dist
                 else np.zeros((dims, 0), tree.coord_dtype))
    return SimpleNamespace(
        raw_classes=raw, class_is_used=used,
        from_sep_siblings_translation_classes=dense[raw],
        from_sep_siblings_translation_class_to_distance_vector=distances,
        from_sep_siblings_translation_classes_level_starts=level_starts)


def rotation_classes(tree, trav):
    """RotationClassesBuilder.__call__ (rotation_classes.py:163-198)."""
    import math
    tc = translation_classes(tree, trav, is_translation_per_level=False)
    nway, dims = trav.well_sep_is_n_away, tree.dimensions
    base, shift = 4 * nway + 3, 2 * nway + 1
    rot_of = np.full(base ** dims, -1, np.int32)
    by_angle, angles = {}, []
    for cls in np.flatnonzero(tc.class_is_used):
        c = int(cls)
        vec = np.zeros(dims, np.int32)
        for d in range(dims):
            vec[d] = c % base - shift
            c //= base
        g = 0
        for e in vec:
            g = math.gcd(g, abs(int(e)))
        vec = vec // g
        angle = np.arccos(vec[-1] / np.linalg.norm(vec))
        if angle not in by_angle:
            by_angle[angle] = len(angles)
            angles.append(angle)
        rot_of[cls] = by_angle[angle]
    return SimpleNamespace(from_sep_siblings_rotation_classes=rot_of[tc.raw_classes],
                           from_sep_siblings_rotation_class_to_angle=np.array(angles))

# }}}

# {{{ cost model (boxtree/cost.py)

def _xlat_costs(dims, nlevels, level_to_order, params, taylor=False):
    """fmm_cost_factors_for_kernels_from_model (cost.py:387-435) with the PDE-aware
    (:152-166) or Taylor (:169-180) translation cost model evaluated numerically."""
    ncoeffs = [(level_to_order[i] + 1) ** (dims if taylor else dims - 1) for i in range(nlevels)]
    pas = (dims == 3) and not taylor

    def e2e(ns, nt):                                                     # cost.py:135-150
        if pas:
            return ns ** (3 / 2) + ns ** (1 / 2) * nt + nt ** (3 / 2)
        return ns * nt

    f = np.float64
    return {
        "p2m_cost": np.array([params["c_p2m"] * ncoeffs[i] for i in range(nlevels)], f),
        "m2m_cost": np.array([params["c_m2m"] * e2e(ncoeffs[i + 1], ncoeffs[i])
                              for i in range(nlevels - 1)], f),
        "c_p2p": params["c_p2p"],
        "m2l_cost": np.array([params["c_m2l"] * e2e(ncoeffs[i], ncoeffs[i])
                              for i in range(nlevels)], f),
        "m2p_cost": np.array([params["c_m2p"] * ncoeffs[i] for i in range(nlevels)], f),
        "p2l_cost": np.array([params["c_p2l"] * ncoeffs[i] for i in range(nlevels)], f),
        "l2l_cost": np.array([params["c_l2l"] * e2e(ncoeffs[i], ncoeffs[i + 1])
                              for i in range(nlevels - 1)], f),
        "l2p_cost": np.array([params["c_l2p"] * ncoeffs[i] for i in range(nlevels)], f),
    }


def cost_model(tree, trav, level_to_order, calibration_params, taylor=False):
    """_PythonFMMCostModel (cost.py:1264-1440) driven by cost_per_box / cost_per_stage
    (:445-624).  -> (cost_per_box [nboxes], cost_per_stage dict)."""
    tc = _xlat_costs(tree.dimensions, tree.nlevels, level_to_order, calibration_params, taylor)
    per_box, per_stage, _ = cost_model_from_factors(tree, trav, tc)
    return per_box, per_stage


def cost_model_from_factors(tree, trav, tc):
    """The per-stage loops of _PythonFMMCostModel (cost.py:1264-1424) for given
    per-level translation costs *tc*.  -> (per_box, per_stage, pieces), *pieces* keyed
    by the reference's method names."""
    nlevels = tree.nlevels
    nsrc = tree.box_source_counts_nonchild
    ntgt = tree.box_target_counts_nonchild
    levels = tree.box_levels

    np2m = np.zeros(len(trav.source_boxes))                              # :1265-1277
    for i, b in enumerate(trav.source_boxes):
        np2m[i] = nsrc[b] * tc["p2m_cost"][levels[b]]

    ndirect = np.zeros(len(trav.target_boxes))                           # :1279-1312
    for i in range(len(trav.target_boxes)):
        for st, li in ((trav.neighbor_source_boxes_starts, trav.neighbor_source_boxes_lists),
                       (trav.from_sep_close_smaller_starts, trav.from_sep_close_smaller_lists),
                       (trav.from_sep_close_bigger_starts, trav.from_sep_close_bigger_lists)):
            if st is not None:
                ndirect[i] += nsrc[li[st[i]:st[i + 1]]].sum()
    direct = ntgt[trav.target_boxes] * ndirect * tc["c_p2p"]            # :1314-1322

    ttp = trav.target_or_target_parent_boxes
    nm2l = tc["m2l_cost"][levels[ttp]] * np.diff(trav.from_sep_siblings_starts)   # :1324-1335

    nm2p = np.zeros(tree.nboxes)                                         # :1337-1353
    for ilevel, ssn in enumerate(trav.from_sep_smaller_by_level):
        for i, b in enumerate(trav.target_boxes_sep_smaller_by_source_level[ilevel]):
            nm2p[b] += ntgt[b] * (ssn.starts[i + 1] - ssn.starts[i]) * tc["m2p_cost"][ilevel]

    np2l = np.zeros(len(ttp))                                            # :1355-1367
    st, li = trav.from_sep_bigger_starts, trav.from_sep_bigger_lists
    for i in range(len(ttp)):
        src = li[st[i]:st[i + 1]]
        np2l[i] = (nsrc[src] * tc["p2l_cost"][levels[src]]).sum()

    nl2p = ntgt[trav.target_boxes] * tc["l2p_cost"][levels[trav.target_boxes]]  # :1369-1385

    coarsen = 0.0                                                        # :1387-1412
    lsp = trav.level_start_source_parent_box_nrs
    for source_level in range(nlevels - 1, 2, -1):
        target_level = source_level - 1
        boxes = trav.source_parent_boxes[lsp[target_level]:lsp[target_level + 1]]
        coarsen += tc["m2m_cost"][target_level] * int(
            (tree.box_child_ids[:, boxes] != 0).sum())
    refine = 0.0                                                         # :1414-1424
    ltt = trav.level_start_target_or_target_parent_box_nrs
    for target_lev in range(1, nlevels):
        refine += (ltt[target_lev + 1] - ltt[target_lev]) * tc["l2l_cost"][target_lev - 1]

    per_box = np.zeros(tree.nboxes)                                      # :487-523
    per_box[trav.source_boxes] += np2m
    per_box[trav.target_boxes] += direct
    per_box[ttp] += nm2l
    per_box += nm2p
    per_box[ttp] += np2l
    per_box[trav.target_boxes] += nl2p
    per_stage = {
        "form_multipoles": np2m.sum(), "coarsen_multipoles": coarsen,
        "eval_direct": direct.sum(), "multipole_to_local": nm2l.sum(),
        "eval_multipoles": nm2p.sum(), "form_locals": np2l.sum(),
        "refine_locals": refine, "eval_locals": nl2p.sum(),
    }
    pieces = {
        "process_form_multipoles": np2m, "get_ndirect_sources_per_target_box": ndirect,
        "process_direct": direct, "process_list2": nm2l, "process_list3": nm2p,
        "process_list4": np2l, "process_eval_locals": nl2p,
        "process_coarsen_multipoles": coarsen, "process_refine_locals": refine,
    }
    return per_box, per_stage, pieces

# }}}

# vim: foldmethod=marker


# ---- distributed FMM evaluation: work partition and local trees ----------------------
# (test infrastructure like everything in this file; plain Python/numpy loops)

def dfs_order(tree):
    """get_box_ids_dfs_order (boxtree/distributed/partition.py:39-57): explicit
    stack, children pushed in ascending child number, hence popped descending."""
    order = np.empty(tree.nboxes, dtype=np.int32)
    idx = 0
    stack = [0]
    nchildren = 2 ** tree.dimensions
    while stack:
        b = stack.pop()
        order[idx] = b
        idx += 1
        for c in range(nchildren):
            ch = int(tree.box_child_ids[c][b])
            if ch > 0:
                stack.append(ch)
    assert idx == tree.nboxes
    return order


def partition_work_segments(cost_per_box, order, nranks):
    """The root-rank loop of partition_work (distributed/partition.py:92-116).
    -> int32 [nranks, 2] of [start, end) depth-first positions; ranks the loop never
    reaches (their rows stay uninitialised upstream) get [nboxes, nboxes)."""
    nboxes = len(order)
    if nranks > nboxes:
        raise RuntimeError("fewer boxes than ranks")                       # :77-79
    total = np.sum(cost_per_box)
    seg = np.full((nranks, 2), nboxes, dtype=np.int32)
    s = 0
    start = 0
    count = 0
    for i in range(nboxes):
        if s + 1 == nranks:
            seg[s] = [start, nboxes]
            break
        count += cost_per_box[order[i]]
        if count > (s + 1) * total / nranks or i == nboxes - 1:
            seg[s] = [start, i + 1]
            start = i + 1
            s += 1
    return seg


def box_masks(tree, trav, responsible_boxes_list):
    """get_box_masks (distributed/partition.py:301-357) -> dict of four int8 masks."""
    nboxes = tree.nboxes
    resp = np.zeros(nboxes, np.int8)
    resp[responsible_boxes_list] = 1

    anc = np.zeros(nboxes, np.int8)                                       # :167-188
    last = resp.copy()
    while last.any():
        new = np.zeros(nboxes, np.int8)
        for b in np.nonzero(last)[0]:
            if b != 0:
                new[tree.box_parent_ids[b]] = 1
        new = new & ~anc.astype(bool)
        anc = anc | new
        last = new

    def add(box_list, mask, starts, lists, out):                          # :134-164
        for i, b in enumerate(box_list):
            if mask[b]:
                out[lists[starts[i]:starts[i + 1]]] = 1

    both = resp | anc
    psrc = resp.copy()                                                    # :191-245
    add(trav.target_boxes, resp, trav.neighbor_source_boxes_starts,
        trav.neighbor_source_boxes_lists, psrc)
    add(trav.target_or_target_parent_boxes, both, trav.from_sep_bigger_starts,
        trav.from_sep_bigger_lists, psrc)
    if tree.targets_have_extent:
        if trav.from_sep_close_smaller_starts is not None:
            add(trav.target_boxes, resp, trav.from_sep_close_smaller_starts,
                trav.from_sep_close_smaller_lists, psrc)
        if trav.from_sep_close_bigger_starts is not None:
            add(trav.target_boxes, both, trav.from_sep_close_bigger_starts,
                trav.from_sep_close_bigger_lists, psrc)

    msrc = np.zeros(nboxes, np.int8)                                      # :248-298
    add(trav.target_or_target_parent_boxes, both, trav.from_sep_siblings_starts,
        trav.from_sep_siblings_lists, msrc)
    for ilevel in range(tree.nlevels):
        ssn = trav.from_sep_smaller_by_level[ilevel]
        add(trav.target_boxes_sep_smaller_by_source_level[ilevel], resp, ssn.starts,
            ssn.lists, msrc)
    return dict(responsible_boxes=resp, ancestor_boxes=anc, point_src_boxes=psrc,
                multipole_src_boxes=msrc)


def local_particles_and_lists(box_mask, starts, counts_nonchild, counts_cumul, nparticles):
    """construct_local_particles_and_lists (distributed/local_tree.py:198-283), index
    part.  -> (local_starts, local_counts_nonchild, local_counts_cumul, particle_idx)."""
    pmask = np.zeros(nparticles, np.int32)
    for b in np.nonzero(box_mask)[0]:
        pmask[starts[b]:starts[b] + counts_nonchild[b]] = 1
    scan = np.zeros(nparticles + 1, np.int64)
    scan[1:] = np.cumsum(pmask)
    lstarts = scan[starts]
    lnonchild = np.where(box_mask, counts_nonchild, 0)
    lcumul = scan[starts + counts_cumul] - scan[starts]
    return (lstarts.astype(np.int32), lnonchild.astype(np.int32), lcumul.astype(np.int32),
            np.nonzero(pmask)[0].astype(np.int32))


def box_to_user_ranks(masks):
    """Compressed [nranks, nboxes] multipole-user masks (local_tree.py:368-399)."""
    nranks, nboxes = masks.shape
    starts = np.zeros(nboxes + 1, np.int32)
    lists = []
    for b in range(nboxes):
        users = [r for r in range(nranks) if masks[r, b]]
        lists.extend(users)
        starts[b + 1] = len(lists)
    return starts, np.array(lists, dtype=np.int32)


def modify_target_flags(box_flags, counts_nonchild, counts_cumul):
    """modify_target_flags_kernel (local_tree.py:155-185)."""
    f = box_flags & ~np.uint8(2 | 8)
    f = f | np.where(counts_nonchild != 0, 2, 0).astype(np.uint8)
    f = f | np.where(counts_nonchild < counts_cumul, 8, 0).astype(np.uint8)
    return f.astype(np.uint8)
