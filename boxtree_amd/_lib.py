"""ctypes binding of ``libboxtree_hip.so`` (C ABI in ``include/boxtree_hip.h``).

The product path has no CPU fallback: if the HIP library is missing or does not
load, importing a builder raises immediately.
"""

from __future__ import annotations

import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libboxtree_hip.so")

BT_MAX_DIMS = 3
BT_MAX_LEVELS = 64
BT_NUM_STAGES = 24

BT_OK, BT_ERR_INVALID, BT_ERR_MAX_LEVELS, BT_ERR_ALLOC, BT_ERR_HIP, \
    BT_ERR_INTERNAL, BT_ERR_UNSUPPORTED = range(7)

BT_F32, BT_F64 = 0, 1
NORMS = {None: 0, "linf": 1, "l2": 2}
KINDS = {"adaptive": 0, "adaptive-level-restricted": 1, "non-adaptive": 2}
CRITS = {"static_linf": 0, "precise_linf": 1, "static_l2": 2}

vp = ct.c_void_p
i32p = ct.c_void_p     # device pointers are passed as plain addresses
P3 = vp * BT_MAX_DIMS
PL = vp * BT_MAX_LEVELS


ABI_VERSION = 11     # BT_ABI_VERSION of include/boxtree_hip.h


class SortStats(ct.Structure):
    _fields_ = [("n", ct.c_int64), ("passes", ct.c_int32), ("pass_ms_avg", ct.c_float),
                ("hist_ms", ct.c_float), ("total_ms", ct.c_float),
                ("first_pass_ms", ct.c_float), ("first_pass_identity", ct.c_int32),
                ("full_pass_ms_avg", ct.c_float), ("full_passes", ct.c_int32),
                ("bytes_per_element_per_pass", ct.c_int32), ("digit_bits", ct.c_int32)]


class TreeParams(ct.Structure):
    _fields_ = [
        ("dims", ct.c_int32), ("coord_kind", ct.c_int32),
        ("nsources", ct.c_int64), ("ntargets", ct.c_int64),
        ("sources", P3), ("targets", P3),
        ("source_radii", vp), ("target_radii", vp),
        ("refine_weights", vp),
        ("max_leaf_refine_weight", ct.c_int32),
        ("kind", ct.c_int32), ("extent_norm", ct.c_int32), ("skip_prune", ct.c_int32),
        ("stick_out_factor", ct.c_double),
        ("bbox_min", ct.c_double * BT_MAX_DIMS), ("bbox_max", ct.c_double * BT_MAX_DIMS),
        ("root_extent", ct.c_double),
        ("top_level", ct.c_int32), ("top_cell_prefix", vp),
        ("source_stride", ct.c_int64), ("target_stride", ct.c_int64),
        ("compute_root_box", ct.c_int32), ("root_extent_stretch", ct.c_double),
        ("top_box_arrive", vp), ("top_box_stay", vp),
    ]


class TreeSizes(ct.Structure):
    _fields_ = [
        ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
        ("nlevels", ct.c_int32), ("key_levels", ct.c_int32),
        ("level_start_box_nrs", ct.c_int32 * (BT_MAX_LEVELS + 1)),
        ("bbox_min", ct.c_double * BT_MAX_DIMS), ("bbox_max", ct.c_double * BT_MAX_DIMS),
        ("root_extent", ct.c_double),
    ]


class TreeArrays(ct.Structure):
    _fields_ = [
        ("user_source_ids", vp), ("sorted_target_ids", vp),
        ("sources", P3), ("targets", P3),
        ("source_radii", vp), ("target_radii", vp),
        ("box_source_starts", vp), ("box_source_counts_nonchild", vp),
        ("box_source_counts_cumul", vp),
        ("box_target_starts", vp), ("box_target_counts_nonchild", vp),
        ("box_target_counts_cumul", vp),
        ("box_parent_ids", vp), ("box_child_ids", vp), ("box_centers", vp),
        ("box_levels", vp), ("box_flags", vp),
        ("box_source_bounding_box_min", vp), ("box_source_bounding_box_max", vp),
        ("box_target_bounding_box_min", vp), ("box_target_bounding_box_max", vp),
        ("level_start_box_nrs", vp),
        ("box_subtree_sizes", vp),
    ]


class StageTimes(ct.Structure):
    _fields_ = [("ms", ct.c_float * BT_NUM_STAGES),
                ("name", ct.c_char_p * BT_NUM_STAGES),
                ("n", ct.c_int32)]


class TravParams(ct.Structure):
    _fields_ = [
        ("dims", ct.c_int32), ("coord_kind", ct.c_int32), ("nlevels", ct.c_int32),
        ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
        ("root_extent", ct.c_double), ("stick_out_factor", ct.c_double),
        ("box_centers", vp), ("box_levels", vp), ("box_child_ids", vp),
        ("box_flags", vp), ("box_parent_ids", vp),
        ("box_target_bounding_box_min", vp), ("box_target_bounding_box_max", vp),
        ("box_source_counts_cumul", vp),
        ("level_start_box_nrs", ct.POINTER(ct.c_int32)),      # host
        ("sources_are_targets", ct.c_int32),
        ("sources_have_extent", ct.c_int32), ("targets_have_extent", ct.c_int32),
        ("well_sep_is_n_away", ct.c_int32),
        ("from_sep_smaller_crit", ct.c_int32),
        ("from_sep_smaller_min_nsources_cumul", ct.c_int32),
        ("source_boxes_mask", vp), ("source_parent_boxes_mask", vp),
        ("force_generic", ct.c_int32),
        ("target_boxes_mask", vp), ("active_level_ranges", ct.POINTER(ct.c_int32)),
        ("box_subtree_sizes", vp),
    ]


class TravSizes(ct.Structure):
    _fields_ = [
        ("nsource_boxes", ct.c_int64), ("ntarget_boxes", ct.c_int64),
        ("nsource_parent_boxes", ct.c_int64),
        ("ntarget_or_target_parent_boxes", ct.c_int64),
        ("n_same_level_non_well_sep", ct.c_int64),
        ("n_neighbor_source", ct.c_int64),
        ("n_from_sep_siblings", ct.c_int64),
        ("n_from_sep_bigger", ct.c_int64),
        ("n_from_sep_close_smaller", ct.c_int64),
        ("n_from_sep_close_bigger", ct.c_int64),
        ("n_from_sep_smaller", ct.c_int64 * BT_MAX_LEVELS),
        ("n_from_sep_smaller_nonempty", ct.c_int64 * BT_MAX_LEVELS),
    ]


class TravArrays(ct.Structure):
    _fields_ = [
        ("source_boxes", vp), ("target_boxes", vp), ("source_parent_boxes", vp),
        ("target_or_target_parent_boxes", vp),
        ("level_start_source_box_nrs", vp), ("level_start_target_box_nrs", vp),
        ("level_start_source_parent_box_nrs", vp),
        ("level_start_target_or_target_parent_box_nrs", vp),
        ("same_level_non_well_sep_boxes_starts", vp),
        ("same_level_non_well_sep_boxes_lists", vp),
        ("neighbor_source_boxes_starts", vp), ("neighbor_source_boxes_lists", vp),
        ("from_sep_siblings_starts", vp), ("from_sep_siblings_lists", vp),
        ("from_sep_bigger_starts", vp), ("from_sep_bigger_lists", vp),
        ("from_sep_close_smaller_starts", vp), ("from_sep_close_smaller_lists", vp),
        ("from_sep_close_bigger_starts", vp), ("from_sep_close_bigger_lists", vp),
        ("from_sep_smaller_starts", PL), ("from_sep_smaller_lists", PL),
        ("from_sep_smaller_nonempty_indices", PL),
        ("from_sep_smaller_compressed_indices", PL),
        ("target_boxes_sep_smaller", PL),
    ]

BT_ROUTE_TO_OWNERS, BT_ROUTE_TO_CALLERS = 0, 1


# every symbol include/boxtree_hip.h declares
class AqTree(ct.Structure):
    """bt_aq_tree"""
    _fields_ = [
        ("dims", ct.c_int32), ("coord_kind", ct.c_int32), ("nlevels", ct.c_int32),
        ("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64),
        ("root_extent", ct.c_double), ("bbox_min", ct.c_double * 3),
        ("box_centers", vp), ("box_levels", vp), ("box_child_ids", vp), ("box_flags", vp),
        ("box_parent_ids", vp), ("level_start_box_nrs", ct.POINTER(ct.c_int32)),
    ]


class MgpuParams(ct.Structure):
    _fields_ = [("dims", ct.c_int32), ("coord_kind", ct.c_int32), ("n", ct.c_int64),
                ("coords", P3), ("top_level", ct.c_int32),
                ("max_particles_in_box", ct.c_int64), ("alloc", vp), ("alloc_user", vp),
                ("ntargets", ct.c_int64), ("targets", P3),
                ("target_radii", vp), ("stick_out_factor", ct.c_double), ("extent_norm", ct.c_int32),
                ("max_leaf_refine_weight", ct.c_int32), ("source_refine_weights", vp),
                ("target_refine_weights", vp)]


class MgpuShard(ct.Structure):
    _fields_ = [("n_owned", ct.c_int64), ("points", vp),
                ("bbox_min", ct.c_double * 3), ("bbox_max", ct.c_double * 3),
                ("root_extent", ct.c_double), ("top_level", ct.c_int32),
                ("top_cell_prefix", vp), ("bytes_sent", ct.c_int64), ("rounds", ct.c_int32),
                ("a2a_ms", ct.c_float), ("n_owned_targets", ct.c_int64), ("target_points", vp),
                ("sep_targets", ct.c_int32), ("target_record_len", ct.c_int32),
                ("target_radii", vp), ("source_record_len", ct.c_int32), ("refine_weights", vp),
                ("top_box_arrive", vp), ("top_box_stay", vp),
                ("source_chunk_offset", ct.c_int64), ("target_chunk_offset", ct.c_int64),
                ("n_global_sources", ct.c_int64), ("n_global_targets", ct.c_int64),
                ("n_sent_sources", ct.c_int64), ("n_sent_targets", ct.c_int64)]


class MgpuLocalTree(ct.Structure):
    _fields_ = [("dims", ct.c_int32), ("coord_kind", ct.c_int32), ("nboxes", ct.c_int64),
                ("aligned_nboxes", ct.c_int64), ("nlevels", ct.c_int32),
                ("level_start_box_nrs", ct.POINTER(ct.c_int32)), ("box_centers", vp),
                ("box_levels", vp), ("box_flags", vp), ("nsources", ct.c_int64),
                ("ntargets", ct.c_int64), ("box_target_bounding_box_min", vp),
                ("box_target_bounding_box_max", vp), ("box_source_counts_cumul", vp),
                ("box_subtree_sizes", vp)]


class MgpuNumbering(ct.Structure):
    _fields_ = [("nlevels", ct.c_int32), ("level_start_box_nrs", ct.c_int32 * (BT_MAX_LEVELS + 2)),
                ("deep_base", ct.c_int32 * (BT_MAX_LEVELS + 1)), ("nboxes", ct.c_int64),
                ("nsources", ct.c_int64), ("ntargets", ct.c_int64),
                ("source_offset", ct.c_int64), ("target_offset", ct.c_int64)]


class MgpuLetSizes(ct.Structure):
    _fields_ = [("nboxes", ct.c_int64), ("aligned_nboxes", ct.c_int64), ("nlevels", ct.c_int32),
                ("level_start_box_nrs", ct.c_int32 * (BT_MAX_LEVELS + 2)),
                ("active_level_ranges", (ct.c_int32 * 2) * (BT_MAX_LEVELS + 1)),
                ("halo_boxes_sent", ct.c_int64), ("halo_boxes_received", ct.c_int64),
                ("loopback_records", ct.c_int64), ("loopback_mismatches", ct.c_int64),
                ("has_subtree_sizes", ct.c_int32)]


class MgpuLetArrays(ct.Structure):
    _fields_ = [("box_centers", vp), ("box_parent_ids", vp), ("box_child_ids", vp),
                ("box_levels", vp), ("box_flags", vp), ("global_box_ids", vp),
                ("target_boxes_mask", vp), ("box_target_bounding_box_min", vp),
                ("box_target_bounding_box_max", vp), ("box_source_counts_cumul", vp),
                ("box_subtree_sizes", vp)]


class Span(ct.Structure):
    _fields_ = [("offset", ct.c_int64), ("count", ct.c_int64)]


SpanL = Span * BT_MAX_LEVELS
LevelStarts = ct.c_int32 * (BT_MAX_LEVELS + 1)
ALLOC_FN = ct.CFUNCTYPE(ct.c_void_p, ct.c_void_p, ct.c_int64)


class TravPacked(ct.Structure):
    _fields_ = [
        ("base", vp), ("total", ct.c_int64), ("nlevels", ct.c_int32),
        ("lattice_path", ct.c_int32), ("sizes", TravSizes),
        ("level_start_source_box_nrs", LevelStarts),
        ("level_start_target_box_nrs", LevelStarts),
        ("level_start_source_parent_box_nrs", LevelStarts),
        ("level_start_target_or_target_parent_box_nrs", LevelStarts),
        ("source_boxes", Span), ("target_boxes", Span), ("source_parent_boxes", Span),
        ("target_or_target_parent_boxes", Span),
        ("same_level_non_well_sep_boxes_starts", Span), ("same_level_non_well_sep_boxes_lists", Span),
        ("neighbor_source_boxes_starts", Span), ("neighbor_source_boxes_lists", Span),
        ("from_sep_siblings_starts", Span), ("from_sep_siblings_lists", Span),
        ("from_sep_bigger_starts", Span), ("from_sep_bigger_lists", Span),
        ("from_sep_close_smaller_starts", Span), ("from_sep_close_smaller_lists", Span),
        ("from_sep_close_bigger_starts", Span), ("from_sep_close_bigger_lists", Span),
        ("from_sep_smaller_starts", SpanL), ("from_sep_smaller_lists", SpanL),
        ("from_sep_smaller_nonempty_indices", SpanL),
        ("from_sep_smaller_compressed_indices", SpanL),
        ("target_boxes_sep_smaller", SpanL),
    ]


EXPORTED_SYMBOLS = [
    "bt_abi_version", "bt_create", "bt_destroy", "bt_trim", "bt_release_cached", "bt_last_error_string",
    "bt_set_stream", "bt_set_stream_ordered", "bt_synchronize", "bt_set_stage_timing",
    "bt_bbox", "bt_radix_sort_u64_u32", "bt_radix_sort_u32_u32", "bt_radix_sort_u64_keys",
    "bt_get_sort_stats",
    "bt_tree_build", "bt_tree_export", "bt_get_stage_times",
    "bt_traversal_build", "bt_traversal_export", "bt_traversal_build_packed", "bt_merge_csr_lists",
    "bt_peer_lists_build", "bt_area_query_build", "bt_csr_export", "bt_leaves_to_balls",
    "bt_space_invader_query",
    "bt_fmm_box_particle_sums", "bt_fmm_csr_sum", "bt_fmm_box_to_particles", "bt_fmm_tree_sweep",
    "bt_translation_classes",
    "bt_filter_targets_user_order", "bt_filter_targets_tree_order", "bt_link_point_sources",
    "bt_box_morton_paths", "bt_let_build", "bt_mgpu_exchange", "bt_mgpu_exchange_time", "bt_mgpu_plan", "bt_mgpu_plan_ext",
    "bt_mgpu_comm_rccl", "bt_mgpu_local_group_create", "bt_mgpu_local_group_destroy",
    "bt_mgpu_comm_local", "bt_mgpu_comm_shm", "bt_mgpu_comm_destroy", "bt_mgpu_use_rccl_library",
    "bt_mgpu_comm_set_self_loopback", "bt_mgpu_number", "bt_mgpu_let_build",
    "bt_mgpu_let_export", "bt_mgpu_route", "bt_mgpu_global_ids",
    "bt_dfs_order", "bt_partition_work", "bt_ancestor_mask", "bt_mark_list_boxes",
    "bt_local_particles", "bt_modify_target_flags", "bt_box_to_user_ranks",
    "bt_boxes_used_by_ranks",
    "bt_morton_cells", "bt_bucket_permutation", "bt_partition_pack", "bt_gather", "bt_gather_pack",
    "bt_unpack",
]

_lib = None


class BoxtreeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libboxtree_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


def load():
    """Load the HIP library; fail loudly if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C boxtree_amd/csrc`). boxtree_amd has no CPU fallback.")
    # PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so).  It
    # must be in the process BEFORE this library is loaded: then the dynamic linker
    # binds libboxtree_hip.so to that copy and both share one runtime (streams,
    # device pointers).  The other order maps /opt/rocm's runtime as well, and the
    # second runtime finds no device.
    import torch  # noqa: F401
    _lib = _bind(ct.CDLL(LIB_PATH))
    return _lib


def _bind(lib):
    """Declare the argument types of every entry (include/boxtree_hip.h) on a loaded image."""
    lib.bt_abi_version.restype = ct.c_int
    lib.bt_last_error_string.restype = ct.c_char_p
    lib.bt_create.argtypes = [ct.c_int, vp, ct.POINTER(vp)]
    lib.bt_destroy.argtypes = [vp]
    lib.bt_destroy.restype = None
    lib.bt_trim.argtypes = [vp]
    lib.bt_release_cached.argtypes = [vp]
    lib.bt_set_stream.argtypes = [vp, vp]
    lib.bt_set_stream_ordered.argtypes = [vp, ct.c_int]
    lib.bt_synchronize.argtypes = [vp]
    lib.bt_set_stage_timing.argtypes = [vp, ct.c_int]
    lib.bt_bbox.argtypes = [vp, ct.c_int, ct.c_int, ct.POINTER(vp), vp, ct.c_int64,
                            ct.POINTER(ct.c_double), ct.POINTER(ct.c_double)]
    for name in ("bt_radix_sort_u64_u32", "bt_radix_sort_u32_u32"):
        getattr(lib, name).argtypes = [vp, vp, vp, vp, vp, ct.c_int64, ct.c_int, ct.c_int]
    lib.bt_radix_sort_u64_keys.argtypes = [vp, vp, vp, ct.c_int64, ct.c_int, ct.c_int]
    lib.bt_get_sort_stats.argtypes = [vp, ct.POINTER(SortStats)]
    lib.bt_tree_build.argtypes = [vp, ct.POINTER(TreeParams), ct.POINTER(TreeSizes)]
    lib.bt_tree_export.argtypes = [vp, ct.POINTER(TreeArrays)]
    lib.bt_get_stage_times.argtypes = [vp, ct.POINTER(StageTimes)]
    lib.bt_traversal_build.argtypes = [vp, ct.POINTER(TravParams), ct.POINTER(TravSizes)]
    lib.bt_traversal_export.argtypes = [vp, ct.POINTER(TravArrays)]
    lib.bt_mgpu_exchange.argtypes = [vp, vp, ct.POINTER(MgpuParams), ct.POINTER(MgpuShard)]
    lib.bt_mgpu_exchange_time.argtypes = [vp, ct.POINTER(ct.c_float)]
    lib.bt_mgpu_comm_rccl.argtypes = [vp, ct.c_int, ct.c_int, ct.POINTER(vp)]
    lib.bt_mgpu_local_group_create.argtypes = [ct.c_int, ct.POINTER(vp)]
    lib.bt_mgpu_local_group_destroy.argtypes = [vp]
    lib.bt_mgpu_local_group_destroy.restype = None
    lib.bt_mgpu_comm_local.argtypes = [vp, ct.c_int, ct.POINTER(vp)]
    lib.bt_mgpu_comm_shm.argtypes = [ct.c_char_p, ct.c_int, ct.c_int, ct.c_int64, ct.c_double, ct.POINTER(vp)]
    lib.bt_mgpu_comm_destroy.argtypes = [vp]
    lib.bt_mgpu_comm_destroy.restype = None
    lib.bt_mgpu_use_rccl_library.argtypes = [ct.c_char_p]
    lib.bt_mgpu_comm_set_self_loopback.argtypes = [vp, ct.c_int]
    lib.bt_mgpu_number.argtypes = [vp, vp, ct.POINTER(MgpuLocalTree), vp, ct.POINTER(MgpuNumbering)]
    lib.bt_mgpu_let_build.argtypes = [vp, vp, ct.POINTER(MgpuLocalTree), vp,
                                      ct.POINTER(MgpuNumbering), ct.c_int, ct.POINTER(MgpuLetSizes)]
    lib.bt_mgpu_let_export.argtypes = [vp, ct.POINTER(MgpuLetArrays)]
    lib.bt_mgpu_route.argtypes = [vp, vp, ct.c_int, ct.c_int, ct.c_int, vp, vp]
    lib.bt_mgpu_global_ids.argtypes = [vp, vp, ct.c_int, ct.c_int, vp]
    lib.bt_mgpu_plan.argtypes = [ct.c_int, ct.c_int, ct.c_int64, ct.c_int, vp, vp, vp]
    lib.bt_mgpu_plan_ext.argtypes = [ct.c_int, ct.c_int, ct.c_int64, ct.c_int, vp, vp, vp, vp, vp, vp]
    lib.bt_traversal_build_packed.argtypes = [vp, ct.POINTER(TravParams), ALLOC_FN, vp,
                                              ct.POINTER(TravPacked)]
    lib.bt_merge_csr_lists.argtypes = [vp, ct.c_int, ct.POINTER(vp), ct.POINTER(vp),
                                       ct.c_int64, vp, vp]
    lib.bt_peer_lists_build.argtypes = [vp, ct.POINTER(AqTree), ct.POINTER(ct.c_int64)]
    lib.bt_area_query_build.argtypes = [vp, ct.POINTER(AqTree), vp, vp, ct.c_int64,
                                        ct.POINTER(vp), vp, ct.POINTER(ct.c_int64)]
    lib.bt_csr_export.argtypes = [vp, vp, vp]
    lib.bt_leaves_to_balls.argtypes = [vp, ct.c_int64, ct.c_int64, vp, vp, ct.c_int64, vp, vp]
    lib.bt_space_invader_query.argtypes = [vp, ct.POINTER(AqTree), vp, vp, ct.c_int64,
                                           ct.POINTER(vp), vp, vp]
    lib.bt_filter_targets_user_order.argtypes = [vp, ct.c_int64, ct.c_int64, vp, vp, vp, vp,
                                                 vp, vp, ct.POINTER(ct.c_int64)]
    lib.bt_filter_targets_tree_order.argtypes = [vp, ct.c_int64, ct.c_int64, vp, vp, vp, vp,
                                                 vp, vp, vp, ct.POINTER(ct.c_int64)]
    lib.bt_link_point_sources.argtypes = [vp, ct.c_int64, ct.c_int64, ct.c_int64] + [vp] * 11
    lib.bt_fmm_box_particle_sums.argtypes = [vp, ct.c_int64, vp, vp, vp, vp, vp, ct.c_int]
    lib.bt_fmm_csr_sum.argtypes = [vp, ct.c_int64, vp, vp, vp, vp, vp, ct.c_int]
    lib.bt_fmm_box_to_particles.argtypes = [vp, ct.c_int64, vp, vp, vp, vp, vp, vp, ct.c_int]
    lib.bt_fmm_tree_sweep.argtypes = [vp, ct.c_int64, vp, vp, ct.c_int64, ct.c_int, vp, vp]
    lib.bt_translation_classes.argtypes = [
        vp, ct.c_int, ct.c_int, ct.c_int64, vp, vp, vp, ct.c_int64, vp, ct.c_int64, ct.c_double,
        vp, ct.c_int, ct.c_int, ct.c_int, vp, vp, ct.POINTER(ct.c_int32)]
    lib.bt_box_morton_paths.argtypes = [vp, ct.c_int, ct.c_int, ct.c_int64, ct.c_int64, vp, vp,
                                        ct.POINTER(ct.c_double), ct.c_double, vp]
    lib.bt_let_build.argtypes = [vp, ct.c_int, ct.c_int, ct.c_int, ct.POINTER(ct.c_int32), vp,
                                 ct.c_int64, ct.POINTER(ct.c_double), ct.POINTER(ct.c_double),
                                 ct.c_double, vp, vp, vp]
    i32p, i64p = ct.POINTER(ct.c_int32), ct.POINTER(ct.c_int64)
    lib.bt_dfs_order.argtypes = [vp, ct.c_int, ct.c_int, i32p, ct.c_int64, ct.c_int64, vp, vp]
    lib.bt_partition_work.argtypes = [vp, ct.c_int64, vp, vp, ct.c_int, i32p]
    lib.bt_ancestor_mask.argtypes = [vp, ct.c_int64, vp, vp, vp]
    lib.bt_mark_list_boxes.argtypes = [vp, ct.c_int64, vp, vp, vp, vp, vp, vp]
    lib.bt_local_particles.argtypes = [vp, ct.c_int64, ct.c_int64, vp, vp, vp, vp, vp, vp, vp,
                                       vp, i64p]
    lib.bt_modify_target_flags.argtypes = [vp, ct.c_int64, vp, vp, vp]
    lib.bt_box_to_user_ranks.argtypes = [vp, ct.c_int, ct.c_int64, vp, vp, vp, i64p]
    lib.bt_boxes_used_by_ranks.argtypes = [vp, ct.c_int64, vp, ct.c_int, ct.c_int, vp, vp, vp,
                                           i64p]
    lib.bt_morton_cells.argtypes = [vp, ct.c_int, ct.c_int, ct.POINTER(vp), ct.c_int64,
                                    ct.POINTER(ct.c_double), ct.POINTER(ct.c_double),
                                    ct.c_int, vp, vp]
    lib.bt_bucket_permutation.argtypes = [vp, vp, ct.c_int64, vp, ct.c_int, vp]
    lib.bt_gather.argtypes = [vp, ct.c_int, vp, vp, ct.c_int64, vp]
    lib.bt_gather_pack.argtypes = [vp, ct.c_int, ct.c_int, ct.POINTER(vp), vp, ct.c_int64, vp]
    lib.bt_partition_pack.argtypes = [vp, ct.c_int, ct.c_int, ct.POINTER(vp), vp, ct.c_int64, vp,
                                      ct.c_int, ct.c_int, ct.c_int64, ct.c_int64, vp, vp]
    lib.bt_unpack.argtypes = [vp, ct.c_int, ct.c_int, vp, ct.c_int64, ct.POINTER(vp)]
    if lib.bt_abi_version() != ABI_VERSION:
        raise RuntimeError("libboxtree_hip.so ABI version mismatch")
    return lib


def check(code):
    if code != BT_OK:
        msg = load().bt_last_error_string().decode("utf-8", "replace")
        raise BoxtreeHipError(code, msg)


# BT_HOST_TRACE=1: stamps of the Python layer on the clock the library's own stage marks
# use (CLOCK_MONOTONIC, microseconds); tools/host_trace.py puts them on one timeline.
_HOST_TRACE = os.environ.get("BT_HOST_TRACE", "0") not in ("", "0")


def host_trace(name):
    if _HOST_TRACE:
        import sys
        import time
        sys.stderr.write(f"[py]      {name:<14s} {time.monotonic_ns() * 1e-3:.1f}\n")
