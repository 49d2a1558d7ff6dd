/*
 * boxtree_hip.h -- C ABI of libboxtree_hip.so (MI355X / gfx950 native).
 *
 * Drop-in boundary for the one data-parallel hot path of inducer/boxtree:
 * particle tree build (TreeBuilder -> Tree) and FMM interaction-list
 * generation (FMMTraversalBuilder -> FMMTraversalInfo).  The reference has no
 * FFI layer of its own: what these entry points replace is the body of
 *
 *   boxtree/bounding_box.py:163   BoundingBoxFinder.__call__        -> bt_bbox
 *   boxtree/tree_build.py:145     TreeBuilder.__call__              -> bt_tree_build
 *                                                                     + bt_tree_export
 *   boxtree/traversal.py:1969     FMMTraversalBuilder.__call__      -> bt_traversal_build
 *                                                                     + bt_traversal_export
 *
 * i.e. everything those methods enqueue on a pyopencl CommandQueue.  The
 * Python layer in boxtree_amd/ (same class names, kwargs and exceptions as the
 * reference) binds them with ctypes; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types.
 *   - every "device pointer" is a HIP device pointer valid on the context's
 *     device; arrays are dense, little-endian, int32 ids, uint8 levels/flags.
 *   - all calls return BT_OK (0) or a BT_ERR_* code; bt_last_error_string()
 *     gives a human-readable message for the calling thread.
 *   - calls are host-synchronous with respect to the context's stream at
 *     return.  A context must not be used from two threads at once.
 *   - two-phase protocol: *_build computes the structure into library-owned
 *     buffers and reports sizes; the caller allocates outputs (e.g. torch
 *     tensors) and *_export writes them.  Inputs passed to *_build must stay
 *     valid until *_export returns.
 */
#ifndef BOXTREE_HIP_H
#define BOXTREE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BT_ABI_VERSION 11
#define BT_MAX_DIMS 3
#define BT_MAX_LEVELS 64       /* capacity of per-level arrays in the structs */

/* device memory from the caller (e.g. a torch tensor it keeps): returns a pointer to
 * nbytes of device memory, or NULL */
typedef void *(*bt_alloc_fn)(void *user, int64_t nbytes);

enum {
    BT_OK = 0,
    BT_ERR_INVALID = 1,        /* bad argument (ValueError/TypeError upstream)   */
    BT_ERR_MAX_LEVELS = 2,     /* boxtree.tree_build.MaxLevelsExceeded            */
    BT_ERR_ALLOC = 3,
    BT_ERR_HIP = 4,
    BT_ERR_INTERNAL = 5,
    BT_ERR_UNSUPPORTED = 6     /* NotImplementedError upstream                    */
};

enum { BT_F32 = 0, BT_F64 = 1 };
enum { BT_NORM_NONE = 0, BT_NORM_LINF = 1, BT_NORM_L2 = 2 };      /* tree_build.py:88 */
enum { BT_KIND_ADAPTIVE = 0, BT_KIND_ADAPTIVE_LEVEL_RESTRICTED = 1,
       BT_KIND_NON_ADAPTIVE = 2 };                                /* tree_build.py:83-86 */
enum { BT_CRIT_STATIC_LINF = 0, BT_CRIT_PRECISE_LINF = 1,
       BT_CRIT_STATIC_L2 = 2 };                                   /* traversal.py:92 */

/* box flags, boxtree/tree.py:109-145 */
enum {
    BT_BOX_IS_SOURCE_BOX = 1, BT_BOX_IS_TARGET_BOX = 2,
    BT_BOX_HAS_SOURCE_CHILD_BOXES = 4, BT_BOX_HAS_TARGET_CHILD_BOXES = 8
};

typedef struct bt_context bt_context;

/* ---- context ----------------------------------------------------------- */

int bt_abi_version(void);

/* hip_stream: a hipStream_t to run on, or NULL for a library-owned stream. */
int bt_create(int device, void *hip_stream, bt_context **out);
void bt_destroy(bt_context *ctx);
/* release cached device workspace (kept between calls otherwise) -- and what the context keeps
 * BETWEEN the calls of one job: a tree not yet exported, the plan of a sharded build
 * (bt_mgpu_number, bt_mgpu_let_build and bt_mgpu_route need the exchange's) */
int bt_trim(bt_context *ctx);
/* give back only the workspace that is not in use: safe between the calls of one job (several
 * contexts taking turns on one device) */
int bt_release_cached(bt_context *ctx);
/* Run later calls on another stream (NULL: the legacy default stream); waits for the work
 * queued on the old one first, because the workspace is ordered by the stream. */
int bt_set_stream(bt_context *ctx, void *hip_stream);
/* Stream-ordered results (off by default).  When on, bt_tree_export and
 * bt_traversal_build_packed / bt_traversal_export return as soon as everything the host
 * needs is known (sizes, level starts, spans): the kernels that fill the caller's arrays
 * may still be running on the context's stream, so the arrays are valid for work queued on
 * that stream afterwards -- or after bt_synchronize -- exactly like the (object, event)
 * pairs the reference returns (tree_build.py:1868, traversal.py:2406).  The device-side
 * status word of such a call is examined at the next wait on this context: an internal
 * failure surfaces there (next call, or bt_synchronize). */
int bt_set_stream_ordered(bt_context *ctx, int on);
/* wait for the context's stream and report any deferred device-side failure */
int bt_synchronize(bt_context *ctx);
/* Per-stage HIP events of the builders (bt_get_stage_times); on by default.  An event
 * record costs a few microseconds of pipeline bubble on the stream, ~30 of them per build +
 * traversal: switch them off where only the result matters. */
int bt_set_stage_timing(bt_context *ctx, int on);
const char *bt_last_error_string(void);

/* ---- bounding box (bounding_box.py:54-122, 163-174) --------------------- */

/* min/max over i of coords[d][i] -/+ radii[i]; results are exact, returned as
 * doubles.  radii may be NULL.  n == 0 gives (+MAX, -MAX) like bbox_neutral(). */
int bt_bbox(bt_context *ctx, int dims, int coord_kind,
            const void *const *coords, const void *radii, int64_t n,
            double *out_min, double *out_max);

/* ---- 64-bit-key radix sort (exposed for the roofline benchmark) ---------- */

/* Stable LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit).
 * keys_in/vals_in are clobbered (ping-pong); the result is in keys_out/vals_out.
 * One digit pass moves 24*n algorithmic bytes (12 read + 12 written per pair). */
int bt_radix_sort_u64_u32(bt_context *ctx, uint64_t *keys_in, uint32_t *vals_in,
                          uint64_t *keys_out, uint32_t *vals_out, int64_t n,
                          int begin_bit, int end_bit);
int bt_radix_sort_u32_u32(bt_context *ctx, uint32_t *keys_in, uint32_t *vals_in,
                          uint32_t *keys_out, uint32_t *vals_out, int64_t n,
                          int begin_bit, int end_bit);

/* Keys only: stable LSD sort of 64-bit words on bits [begin_bit, end_bit); the bits
 * below begin_bit ride along (the tree build of point particles packs the user id there:
 * one word per particle, 16*n algorithmic bytes per digit pass).  8- or 9-bit digits,
 * whichever needs fewer passes.  keys_in is clobbered; the result is in keys_out. */
int bt_radix_sort_u64_keys(bt_context *ctx, uint64_t *keys_in, uint64_t *keys_out, int64_t n,
                           int begin_bit, int end_bit);

typedef struct {
    int64_t n;                 /* pairs sorted                                    */
    int32_t passes;            /* digit passes of the last 64-bit-key sort (the   */
                               /* tree build's main sort or bt_radix_sort_u64_u32) */
    float pass_ms_avg;         /* HIP-event time of the onesweep kernels / passes */
    float hist_ms;             /* up-front histogram kernel                      */
    float total_ms;
    /* the first pass apart: with synthesised values (the tree build's sort starts   */
    /* from 0..n-1) it reads no value array, 20 instead of 24 bytes per pair          */
    float first_pass_ms;
    int32_t first_pass_identity;
    float full_pass_ms_avg;    /* passes that read and write keys and values      */
    int32_t full_passes;
    /* what a full pass moves per element: 24 = (u64 key, u32 value) pairs read and     */
    /* written; 16 = keys only -- a tree build of point particles packs the user id     */
    /* under the Morton path bits of ONE 64-bit word (bt_tree.hip, "packed keys") --    */
    int32_t bytes_per_element_per_pass;
    int32_t digit_bits;        /* 8, or 9 for the keys-only sort when that saves a pass */
} bt_sort_stats;
int bt_get_sort_stats(bt_context *ctx, bt_sort_stats *out);

/* ---- tree build (tree_build.py:145-1878) -------------------------------- */

typedef struct {
    int32_t dims;              /* 2 or 3 (1 accepted) */
    int32_t coord_kind;        /* BT_F32 / BT_F64 */
    int64_t nsources;
    int64_t ntargets;          /* -1: sources are also the targets (targets=None) */
    const void *sources[BT_MAX_DIMS];     /* device, [nsources] each */
    const void *targets[BT_MAX_DIMS];     /* device, [ntargets] each, or NULL */
    const void *source_radii;  /* device [nsources] or NULL */
    const void *target_radii;  /* device [ntargets] or NULL */
    const int32_t *refine_weights;  /* device [nsources(+ntargets)] or NULL = all 1 */
    int32_t max_leaf_refine_weight; /* = max_particles_in_box when weights NULL   */
    int32_t kind;              /* BT_KIND_* */
    int32_t extent_norm;       /* BT_NORM_* (NONE when no radii)                  */
    int32_t skip_prune;        /* keep empty boxes (tree_build.py:1328, undocumented there) */
    double stick_out_factor;
    /* root box exactly as computed on the host by tree_build.py:456-510
     * (values representable in the coordinate type):                       */
    double bbox_min[BT_MAX_DIMS];
    double bbox_max[BT_MAX_DIMS];
    double root_extent;
    /* Sharded builds only (no counterpart in the reference): this call sees the
     * particles of a contiguous Morton range of level-`top_level` cells of a larger
     * point set.  top_cell_prefix[2^(dims*top_level) + 1] (device) is the exclusive
     * prefix sum of the GLOBAL particle counts of those cells; boxes above
     * top_level split where the global tree does.  NULL: a self-contained build. */
    int32_t top_level;
    const int64_t *top_cell_prefix;
    /* Elements between consecutive points in sources[]/targets[] (0 or 1: dense
     * arrays).  Lets the exchange of a sharded build hand over its interleaved
     * receive buffer (x0 y0 z0 x1 ...: stride = dims) without unpacking it. */
    int64_t source_stride, target_stride;
    /* compute_root_box != 0: bbox_min / bbox_max / root_extent above are ignored; the
     * library finds the bounding box of sources and targets on the device and derives
     * the root box there with tree_build.py:456-476's arithmetic in the coordinate type
     * (root_extent = max_axis(max - min) * (1 + root_extent_stretch), upper corner =
     * lower corner + root_extent), without a host round trip; the values come back in
     * bt_tree_sizes.  Point particles (no radii), kind ADAPTIVE or NON_ADAPTIVE, dense
     * arrays, no top_cell_prefix; BT_ERR_UNSUPPORTED otherwise. */
    int32_t compute_root_box;
    double root_extent_stretch;            /* tree_build.py:101: 1e-4 */
    /* Sharded builds of particles with extents (radii): where a particle stops is not a
     * function of the cell counts, so the top of the global tree comes as two tables over the
     * boxes of levels 0..top_level, box (level, Morton path) at index (C^level - 1) / (C - 1) +
     * path with C = 2^dims (device, int64): the particles that ARRIVE in the box (its cumulative
     * count in the global tree) and those that STAY in it (stick out of its children,
     * tree_build_kernels.py:388-428).  A box splits iff arrive - stay exceeds
     * max_leaf_refine_weight.  With refine weights both tables hold weights instead of counts.
     * Both NULL: point particles with unit weights, top_cell_prefix alone. */
    const int64_t *top_box_arrive, *top_box_stay;
} bt_tree_params;

typedef struct {
    int64_t nboxes;
    int64_t aligned_nboxes;    /* ceil(nboxes/32)*32, tree_build.py:1641 */
    int32_t nlevels;
    int32_t key_levels;        /* deepest level the 64-bit key can address */
    int32_t level_start_box_nrs[BT_MAX_LEVELS + 1];   /* [nlevels+1] valid */
    /* the root box the tree was built in (inputs echoed, or computed: compute_root_box) */
    double bbox_min[BT_MAX_DIMS], bbox_max[BT_MAX_DIMS], root_extent;
} bt_tree_sizes;

int bt_tree_build(bt_context *ctx, const bt_tree_params *params, bt_tree_sizes *out);

/* Output arrays, caller-allocated device memory; names/layouts are those of
 * boxtree.Tree (tree.py:298-590).  When sources are targets the target_*
 * pointers may be NULL (the reference shares the arrays). */
typedef struct {
    int32_t *user_source_ids;              /* [nsources] */
    int32_t *sorted_target_ids;            /* [ntargets] */
    void *sources[BT_MAX_DIMS];            /* [nsources] tree order */
    void *targets[BT_MAX_DIMS];            /* [ntargets] or NULL */
    void *source_radii;                    /* or NULL */
    void *target_radii;                    /* or NULL */
    int32_t *box_source_starts, *box_source_counts_nonchild, *box_source_counts_cumul;
    int32_t *box_target_starts, *box_target_counts_nonchild, *box_target_counts_cumul;
    int32_t *box_parent_ids;               /* [nboxes] */
    int32_t *box_child_ids;                /* [2^d, aligned_nboxes] */
    void *box_centers;                     /* [d, aligned_nboxes] */
    uint8_t *box_levels;                   /* [nboxes] */
    uint8_t *box_flags;                    /* [nboxes] */
    void *box_source_bounding_box_min, *box_source_bounding_box_max;  /* [d, aligned] */
    void *box_target_bounding_box_min, *box_target_bounding_box_max;  /* or NULL */
    int32_t *level_start_box_nrs;          /* [nlevels+1] device copy of bt_tree_sizes', or NULL */
    /* Not a boxtree.Tree array: the number of boxes in the subtree of every box (itself
     * included), a by-product of the bottom-up sweep that computes the bounding boxes.
     * bt_trav_params.box_subtree_sizes takes it back and saves the traversal a sweep of
     * its own (one launch per level).  [nboxes], or NULL. */
    int32_t *box_subtree_sizes;
} bt_tree_arrays;

int bt_tree_export(bt_context *ctx, const bt_tree_arrays *out);

/* per-stage milliseconds of the last tree build / traversal (HIP events) */
#define BT_NUM_STAGES 24
typedef struct {
    float ms[BT_NUM_STAGES];
    const char *name[BT_NUM_STAGES];
    int32_t n;
} bt_stage_times;
int bt_get_stage_times(bt_context *ctx, bt_stage_times *out);

/* ---- traversal (traversal.py:1969-2345) --------------------------------- */

typedef struct {
    int32_t dims, coord_kind;
    int32_t nlevels;
    int64_t nboxes, aligned_nboxes;
    double root_extent;
    double stick_out_factor;
    /* device arrays in boxtree.Tree layout */
    const void *box_centers;               /* [d, aligned] */
    const uint8_t *box_levels;
    const int32_t *box_child_ids;          /* [2^d, aligned] */
    const uint8_t *box_flags;
    const int32_t *box_parent_ids;
    const void *box_target_bounding_box_min, *box_target_bounding_box_max;
    const int32_t *box_source_counts_cumul;
    const int32_t *level_start_box_nrs;    /* HOST pointer, [nlevels+1] */
    int32_t sources_are_targets;
    int32_t sources_have_extent, targets_have_extent;
    int32_t well_sep_is_n_away;
    int32_t from_sep_smaller_crit;         /* BT_CRIT_* */
    int32_t from_sep_smaller_min_nsources_cumul;
    const int8_t *source_boxes_mask;           /* device or NULL */
    const int8_t *source_parent_boxes_mask;    /* device or NULL */
    int32_t force_generic;     /* 1: always use the walk-from-root kernels (any tree);
                                  0: use the parent-colleague kernels when the box
                                  numbering allows (integer-lattice form when the
                                  centres are exact lattice centres);
                                  2: as 0, but never the integer-lattice form */
    /* Sharded traversals only (no counterpart in the reference): build the lists of
     * a subset of the target boxes of a tree whose box arrays are complete.
     * target_boxes_mask[nboxes] (device, or NULL = all) filters target_boxes and
     * target_or_target_parent_boxes the way source_boxes_mask filters source_boxes;
     * active_level_ranges[2*nlevels] (HOST, or NULL) gives per level the box range
     * [begin, end) that contains every masked box and their ancestors -- colleague
     * lists are only built (and only valid) inside these ranges. */
    const int8_t *target_boxes_mask;
    const int32_t *active_level_ranges;
    /* Optional: bt_tree_arrays.box_subtree_sizes of the SAME tree (device, [nboxes]), or
     * NULL -- the traversal then counts the subtrees itself. */
    const int32_t *box_subtree_sizes;
} bt_trav_params;

typedef struct {
    int64_t nsource_boxes, ntarget_boxes, nsource_parent_boxes,
            ntarget_or_target_parent_boxes;
    int64_t n_same_level_non_well_sep;     /* total list entries */
    int64_t n_neighbor_source;
    int64_t n_from_sep_siblings;
    int64_t n_from_sep_bigger;
    int64_t n_from_sep_close_smaller;      /* -1 when the list does not exist */
    int64_t n_from_sep_close_bigger;       /* -1 when the list does not exist */
    int64_t n_from_sep_smaller[BT_MAX_LEVELS];          /* entries per source level */
    int64_t n_from_sep_smaller_nonempty[BT_MAX_LEVELS]; /* num_nonempty_lists      */
} bt_trav_sizes;

int bt_traversal_build(bt_context *ctx, const bt_trav_params *params, bt_trav_sizes *out);

typedef struct {
    int32_t *source_boxes, *target_boxes, *source_parent_boxes,
            *target_or_target_parent_boxes;
    int32_t *level_start_source_box_nrs, *level_start_target_box_nrs,
            *level_start_source_parent_box_nrs,
            *level_start_target_or_target_parent_box_nrs;      /* [nlevels+1] */
    int32_t *same_level_non_well_sep_boxes_starts, *same_level_non_well_sep_boxes_lists;
    int32_t *neighbor_source_boxes_starts, *neighbor_source_boxes_lists;
    int32_t *from_sep_siblings_starts, *from_sep_siblings_lists;
    int32_t *from_sep_bigger_starts, *from_sep_bigger_lists;
    int32_t *from_sep_close_smaller_starts, *from_sep_close_smaller_lists;
    int32_t *from_sep_close_bigger_starts, *from_sep_close_bigger_lists;
    /* per source level (BuiltList with eliminate_empty_output_lists) */
    int32_t *from_sep_smaller_starts[BT_MAX_LEVELS];            /* [nonempty+1] */
    int32_t *from_sep_smaller_lists[BT_MAX_LEVELS];
    int32_t *from_sep_smaller_nonempty_indices[BT_MAX_LEVELS];  /* [nonempty]   */
    int32_t *from_sep_smaller_compressed_indices[BT_MAX_LEVELS];/* [ntarget_boxes+1] */
    int32_t *target_boxes_sep_smaller[BT_MAX_LEVELS];           /* [nonempty]   */
} bt_trav_arrays;

int bt_traversal_export(bt_context *ctx, const bt_trav_arrays *out);

/* One-call variant: every output array is a span of ONE int32 block that the caller
 * allocates through a callback once all sizes are known (device memory, 256-byte
 * aligned, `nbytes` long; return NULL to fail).  The library writes the large lists
 * straight into the block -- no export copy -- and returns the spans (offsets and
 * counts in int32 elements from `base`) plus host copies of the four level-start
 * arrays, which the reference hands out as host arrays (traversal.py:2091).
 * target_boxes is the span of source_boxes when the tree's sources are its targets. */
typedef struct { int64_t offset, count; } bt_span;
typedef struct {
    void *base;
    int64_t total;                         /* int32 elements in the block */
    int32_t nlevels;
    int32_t lattice_path;                  /* information: the integer-lattice kernels ran */
    bt_trav_sizes sizes;
    int32_t level_start_source_box_nrs[BT_MAX_LEVELS + 1];
    int32_t level_start_target_box_nrs[BT_MAX_LEVELS + 1];
    int32_t level_start_source_parent_box_nrs[BT_MAX_LEVELS + 1];
    int32_t level_start_target_or_target_parent_box_nrs[BT_MAX_LEVELS + 1];
    bt_span source_boxes, target_boxes, source_parent_boxes, target_or_target_parent_boxes;
    bt_span same_level_non_well_sep_boxes_starts, same_level_non_well_sep_boxes_lists;
    bt_span neighbor_source_boxes_starts, neighbor_source_boxes_lists;
    bt_span from_sep_siblings_starts, from_sep_siblings_lists;
    bt_span from_sep_bigger_starts, from_sep_bigger_lists;
    bt_span from_sep_close_smaller_starts, from_sep_close_smaller_lists;   /* count -1: absent */
    bt_span from_sep_close_bigger_starts, from_sep_close_bigger_lists;
    bt_span from_sep_smaller_starts[BT_MAX_LEVELS];
    bt_span from_sep_smaller_lists[BT_MAX_LEVELS];
    bt_span from_sep_smaller_nonempty_indices[BT_MAX_LEVELS];
    bt_span from_sep_smaller_compressed_indices[BT_MAX_LEVELS];
    bt_span target_boxes_sep_smaller[BT_MAX_LEVELS];
} bt_trav_packed;
int bt_traversal_build_packed(bt_context *ctx, const bt_trav_params *params, bt_alloc_fn alloc,
                              void *user, bt_trav_packed *out);

/* Row-wise concatenation of up to 4 CSR lists with equal row count
 * (FMMTraversalInfo.merge_close_lists, traversal.py:1650-1693 / _ListMerger
 * :1222-1344): out row i = lists[0] row i ++ lists[1] row i ++ ...  out_starts has
 * nrows+1 entries; out_lists must hold the sum of the input list lengths. */
int bt_merge_csr_lists(bt_context *ctx, int nlists, const int32_t *const *starts,
                       const int32_t *const *lists, int64_t nrows, int32_t *out_starts,
                       int32_t *out_lists);

/* ---- area queries (boxtree/area_query.py) -------------------------------- */

/* The tree arrays an area query reads, in boxtree.Tree layout (device pointers). */
typedef struct {
    int32_t dims, coord_kind;
    int32_t nlevels;
    int64_t nboxes, aligned_nboxes;
    double root_extent;
    double bbox_min[3];                    /* tree.bounding_box[0] */
    const void *box_centers;               /* [d, aligned] */
    const uint8_t *box_levels;
    const int32_t *box_child_ids;          /* [2^d, aligned] */
    const uint8_t *box_flags;
    const int32_t *box_parent_ids;         /* peer lists only */
    const int32_t *level_start_box_nrs;    /* HOST pointer, [nlevels+1]; peer lists only */
} bt_aq_tree;

/* PeerListFinder.__call__ (area_query.py:1152-1192; kernel :393-475).  Builds the
 * CSR peer lists of all boxes into the context; *n_entries = len(peer_lists).
 * Fetch with bt_csr_export(starts[nboxes+1], lists[n_entries]). */
int bt_peer_lists_build(bt_context *ctx, const bt_aq_tree *tree, int64_t *n_entries);

/* AreaQueryBuilder.__call__ (area_query.py:744-812; kernels :172-366): for every
 * l^inf ball the leaves that overlap it.  ball_centers[dims] and ball_radii are
 * device arrays of the tree's coordinate type.  Result held in the context:
 * bt_csr_export(leaves_near_ball_starts[nballs+1], leaves_near_ball_lists[n]). */
int bt_area_query_build(bt_context *ctx, const bt_aq_tree *tree,
                        const int32_t *peer_list_starts, const int32_t *peer_lists,
                        int64_t nballs, const void *const *ball_centers,
                        const void *ball_radii, int64_t *n_entries);

/* Copies (device to device) the CSR built by the last bt_peer_lists_build /
 * bt_area_query_build and releases it. */
int bt_csr_export(bt_context *ctx, int32_t *starts, int32_t *lists);

/* LeavesToBallsLookupBuilder.__call__ (area_query.py:847-924): transposes an
 * area query result -- (ball, leaf) pairs stably sorted by leaf.
 * balls_near_box_starts[nboxes+1], balls_near_box_lists[n_entries]. */
int bt_leaves_to_balls(bt_context *ctx, int64_t nballs, int64_t nboxes,
                       const int32_t *leaves_near_ball_starts,
                       const int32_t *leaves_near_ball_lists, int64_t n_entries,
                       int32_t *balls_near_box_starts, int32_t *balls_near_box_lists);

/* SpaceInvaderQueryBuilder.__call__ (area_query.py:970-1056; kernel :613-651):
 * out[nboxes] (float32, like the reference's kernel; zeroed here) receives per
 * leaf the largest l^inf distance from its centre to the centre of a ball that
 * overlaps it. */
int bt_space_invader_query(bt_context *ctx, const bt_aq_tree *tree,
                           const int32_t *peer_list_starts, const int32_t *peer_lists,
                           int64_t nballs, const void *const *ball_centers,
                           const void *ball_radii, float *out);

/* ---- target filtering and point-source linking (boxtree/tree.py) ----------- */

/* ParticleListFilter.filter_target_lists_in_user_order (tree.py:1115-1150): per
 * box the user-order numbers of its (non-child) targets whose flag is nonzero.
 * target_starts[nboxes+1]; target_lists needs room for ntargets entries, the
 * first *nfiltered are valid. */
int bt_filter_targets_user_order(bt_context *ctx, int64_t nboxes, int64_t ntargets,
                                 const int8_t *user_order_flags, const int32_t *sorted_target_ids,
                                 const int32_t *box_target_starts,
                                 const int32_t *box_target_counts_nonchild,
                                 int32_t *target_starts, int32_t *target_lists,
                                 int64_t *nfiltered);

/* ParticleListFilter.filter_target_lists_in_tree_order (tree.py:1175-1241;
 * tree_build_kernels.py:1951-2021): renumbering of the flagged targets in tree
 * order.  Outputs [nboxes], [nboxes], [ntargets capacity; *nfiltered valid]. */
int bt_filter_targets_tree_order(bt_context *ctx, int64_t nboxes, int64_t ntargets,
                                 const int8_t *user_order_flags, const int32_t *sorted_target_ids,
                                 const int32_t *box_target_starts,
                                 const int32_t *box_target_counts_nonchild,
                                 int32_t *box_target_starts_filtered,
                                 int32_t *box_target_counts_nonchild_filtered,
                                 int32_t *unfiltered_from_filtered, int64_t *nfiltered);

/* link_point_sources (tree.py:772-949; tree_build_kernels.py:1871-1947).
 * point_source_starts[nsources+1] is in user source order; npoint_sources must be
 * the number of point sources it describes (the size of user_point_source_ids).
 * Outputs: [nsources] x2, [npoint_sources], [nboxes] x3. */
int bt_link_point_sources(bt_context *ctx, int64_t nsources, int64_t nboxes,
                          int64_t npoint_sources, const int32_t *point_source_starts,
                          const int32_t *user_source_ids, const int32_t *box_source_starts,
                          const int32_t *box_source_counts_nonchild,
                          const int32_t *box_source_counts_cumul,
                          int32_t *tree_order_point_source_starts,
                          int32_t *tree_order_point_source_counts,
                          int32_t *user_point_source_ids, int32_t *box_point_source_starts,
                          int32_t *box_point_source_counts_nonchild,
                          int32_t *box_point_source_counts_cumul);

/* ---- constant-one FMM evaluation (boxtree/constant_one.py:49-237, driven by
 *      boxtree/fmm.py:342-532): one float64 per box stands for an expansion ---- */

/* out[b] (+)= sum(values[box_starts[b] : box_starts[b]+box_counts[b]]) for the n
 * listed boxes (boxes == NULL: boxes 0..n-1).  form_multipoles (:86-97) and the
 * per-source-box weights that eval_direct / form_locals add up (:125-146, :190-212). */
int bt_fmm_box_particle_sums(bt_context *ctx, int64_t n, const int32_t *boxes,
                             const int32_t *box_starts, const int32_t *box_counts,
                             const double *values, double *out, int accumulate);

/* Row sums of a CSR interaction list: r[i] = sum(box_values[lists[starts[i]:starts[i+1]]]).
 * scatter_add == 0: out[i] = r[i] (out has nrows entries);
 * scatter_add == 1: out[row_boxes[i]] += r[i] (out is indexed by box number) --
 * multipole_to_local (:148-166), form_locals (:190-212). */
int bt_fmm_csr_sum(bt_context *ctx, int64_t nrows, const int32_t *starts, const int32_t *lists,
                   const double *box_values, const int32_t *row_boxes, double *out,
                   int scatter_add);

/* pot[box_starts[b] : +box_counts[b]] (+)= v for the boxes b = row_boxes[i], with
 * v = row_values[i] (row_values != NULL) or box_values[b]: eval_direct (:144),
 * eval_multipoles (:186), eval_locals (:225-235). */
int bt_fmm_box_to_particles(bt_context *ctx, int64_t nrows, const int32_t *row_boxes,
                            const int32_t *box_starts, const int32_t *box_counts,
                            const double *row_values, const double *box_values, double *pot,
                            int accumulate);

/* One level of an upward (child_ids != NULL: box += its children, coarsen_multipoles
 * :99-123) or downward (parent_ids: box += its parent, refine_locals :214-223) sweep
 * over the listed boxes. */
int bt_fmm_tree_sweep(bt_context *ctx, int64_t n, const int32_t *boxes, const int32_t *child_ids,
                      int64_t aligned_nboxes, int nchildren, const int32_t *parent_ids,
                      double *box_values);

/* TranslationClassesBuilder.compute_translation_classes (translation_classes.py:325-378;
 * kernel :62-189), shared by RotationClassesBuilder (rotation_classes.py:166-177):
 * classes[i] = class of the translation from list-2 entry i to its target box,
 * class_is_used[nclasses] (zeroed here) marks the classes that occur; nclasses =
 * (4n+3)^d, times nlevels if per_level.  *error != 0: a pair of boxes on different
 * levels or further apart than 2n+1 boxes (ValueError upstream). */
int bt_translation_classes(bt_context *ctx, int dims, int coord_kind, int64_t n_entries,
                           const int32_t *from_sep_siblings_lists,
                           const int32_t *from_sep_siblings_starts,
                           const int32_t *target_or_target_parent_boxes, int64_t nttp,
                           const void *box_centers, int64_t aligned_nboxes, double root_extent,
                           const uint8_t *box_levels, int well_sep_is_n_away, int per_level,
                           int nclasses, int32_t *classes, int32_t *class_is_used, int32_t *error);

/* ---- multi-GPU exchange helpers (no counterpart in the reference, which never
 *      builds the tree in parallel: boxtree/distributed/__init__.py:183-199) ---- */

/* Level-`level` Morton cell of every point (same float expression as the key
 * kernel) into cells_out[n]; hist_inout[2^(dims*level)] (device, int32) is
 * incremented per point. */
int bt_morton_cells(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                    int64_t n, const double *bbox_min, const double *bbox_max, int level,
                    uint32_t *cells_out, int32_t *hist_inout);

/* perm_out[n]: original indices grouped by owner_of_cell[cells[i]] (ascending
 * owner, original order inside a group) -- the send order of the all-to-all. */
int bt_bucket_permutation(bt_context *ctx, const uint32_t *cells, int64_t n,
                          const int32_t *owner_of_cell, int nranks, uint32_t *perm_out);

/* The same send buffer in ONE sweep over the coordinates (no permutation, no gather): the
 * particles are partitioned by owner_of_cell[cells[i]], stably, and their coordinates are
 * written interleaved (out[k*dims + ax]) in owner order into `send` -- except the segment
 * of self_rank, which goes straight to `recv` at record offset self_recv_offset (its send
 * offset, self_send_offset, is the number of particles of the lower ranks).  Reads are
 * sequential and writes go to nranks advancing runs, whereas bucket permutation + gather
 * reads the coordinate arrays in nranks interleaved strided passes. */
int bt_partition_pack(bt_context *ctx, int dims, int elem_size, const void *const *in,
                      const uint32_t *cells, int64_t n, const int32_t *owner_of_cell, int nranks,
                      int self_rank, int64_t self_send_offset, int64_t self_recv_offset,
                      void *send, void *recv);

/* out[i] = in[perm[i]] for 4- or 8-byte elements */
int bt_gather(bt_context *ctx, int elem_size, const void *in, const uint32_t *perm, int64_t n,
              void *out);

/* send buffer of the all-to-all: out[i*dims + ax] = in[ax][perm[i]] (one message
 * per peer carries all coordinates), and its inverse on the receiving side:
 * out[ax][i] = in[i*dims + ax]. */
int bt_gather_pack(bt_context *ctx, int dims, int elem_size, const void *const *in,
                   const uint32_t *perm, int64_t n, void *out);
int bt_unpack(bt_context *ctx, int dims, int elem_size, const void *in, int64_t n,
              void *const *out);

/* ---- multi-GPU entries: a sharded build and the lists of a rank's own boxes, one process
 *      (or thread) per GPU (no counterpart in the reference, which
 *      builds on one rank: boxtree/distributed/__init__.py:183-199) ---- */

/* A communicator: an RCCL communicator the caller owns (ncclComm_t of nranks ranks with
 * this process at `rank`, created on the context's device), or -- so that the multi-rank
 * logic can run on a box with one GPU, which RCCL refuses to share between ranks -- ranks
 * that are threads of one process: bt_mgpu_local_group_create makes the rendezvous object
 * all of them pass to bt_mgpu_comm_local, and every rank calls the collective entries
 * below from its own thread with its own bt_context.  Collectives are enqueued on the
 * context's stream. */
typedef struct bt_mgpu_comm bt_mgpu_comm;
int bt_mgpu_comm_rccl(void *nccl_comm, int rank, int nranks, bt_mgpu_comm **out);
int bt_mgpu_local_group_create(int nranks, void **group);
void bt_mgpu_local_group_destroy(void *group);
int bt_mgpu_comm_local(void *group, int rank, bt_mgpu_comm **out);
/* Ranks as PROCESSES that share a GPU (RCCL refuses that; threads share an address space): the
 * collectives are staged through the POSIX shared-memory segment `name` ("/..."; every rank of
 * the job opens the same fresh name, the last one to destroy its communicator unlinks it) in
 * slots of slot_bytes per rank (<= 0: 64 MiB; larger messages go in rounds); a rank that waits
 * longer than timeout_s (<= 0: 120 s) for a peer fails the group.  Every byte crosses the host:
 * a vehicle for running the N-rank code as N real processes on a box with one GPU (tests,
 * `bench.py --gpus 2` there), not a transport to measure. */
int bt_mgpu_comm_shm(const char *name, int rank, int nranks, int64_t slot_bytes, double timeout_s,
                     bt_mgpu_comm **out);
void bt_mgpu_comm_destroy(bt_mgpu_comm *comm);
/* The library binds the few RCCL entry points it uses at run time (no link-time dependency).
 * A communicator must be used through the image of RCCL that made it: name that image here
 * (before the first bt_mgpu_comm_rccl) when the process could hold more than one, e.g. the
 * copy inside a torch wheel next to /opt/rocm/lib.  Without the call an image already loaded
 * is preferred over loading one. */
int bt_mgpu_use_rccl_library(const char *path);
/* Test switch for boxes with one GPU (environment: BT_MGPU_SELF_LOOPBACK=1 at communicator
 * creation): a rank's messages to ITSELF -- its own segment of the particle all-to-all-v, and
 * an echo of the halo records in bt_mgpu_let_build -- travel through ncclSend / ncclRecv
 * (512-MiB rounds and all) instead of a device copy, so that the point-to-point branch runs
 * with a world of one rank.  Results are unchanged. */
int bt_mgpu_comm_set_self_loopback(bt_mgpu_comm *comm, int on);

typedef struct {
    int32_t dims, coord_kind;
    int64_t n;                         /* particles of this rank's chunk              */
    const void *coords[BT_MAX_DIMS];   /* device, [n] each (sources == targets, points) */
    int32_t top_level;                 /* level of the ownership cells; 0: default (5 in 3D) */
    int64_t max_particles_in_box;      /* > 0: derive the top of the global tree, keep its  */
                                       /* leaves on one rank (kind "adaptive", unit weights) */
    bt_alloc_fn alloc;                 /* NULL: `points` below is owned by the context;     */
    void *alloc_user;                  /* else the receive buffer comes from the caller     */
    int64_t ntargets;                  /* separate point targets of this rank's chunk, or 0 */
    const void *targets[BT_MAX_DIMS];  /* (sources are the targets); exchanged by the same  */
                                       /* owners, cells counted over sources AND targets     */
    /* targets with extents (boxtree's target_radii; every rank passes them or none does):
     * a target that sticks out of the boxes of the shared top levels stays in one of them
     * (tree_build_kernels.py:388-428) and travels to the owner of that box's first cell; the
     * received records are [x.. , radius] (target_record_len = dims + 1) */
    const void *target_radii;          /* device [ntargets] or NULL                          */
    double stick_out_factor;
    int32_t extent_norm;               /* BT_NORM_LINF / BT_NORM_L2 with target_radii        */
    /* refine weights (boxtree's refine_weights / max_leaf_refine_weight; the limit is a parameter
     * of the job, > 0 on every rank or on none): a box splits iff the weight bound for its
     * children exceeds the limit (tree_build_kernels.py:569-591), so the exchange sums weights
     * per cell next to the counts; the weights travel with the particles.  A NULL array on a
     * rank of a weighted job means unit weights. */
    int32_t max_leaf_refine_weight;
    const int32_t *source_refine_weights;   /* device [n] or NULL           */
    const int32_t *target_refine_weights;   /* device [ntargets] or NULL    */
} bt_mgpu_params;

typedef struct {
    int64_t n_owned;                   /* particles this rank owns after the exchange  */
    void *points;                      /* device, interleaved [n_owned][dims]; owned by the */
                                       /* context until the next exchange / bt_destroy  */
                                       /* unless params.alloc provided it               */
    double bbox_min[BT_MAX_DIMS], bbox_max[BT_MAX_DIMS], root_extent;   /* global root box */
    int32_t top_level;
    const int64_t *top_cell_prefix;    /* device [2^(dims*top_level) + 1] or NULL        */
    int64_t bytes_sent;                /* payload bytes that left this GPU              */
    int32_t rounds;                    /* point-to-point rounds of the all-to-all       */
    float a2a_ms;                      /* device time of the payload all-to-all-v (HIP   */
                                       /* events on the context's stream); -1 on a       */
                                       /* stream-ordered context: bt_mgpu_exchange_time  */
    int64_t n_owned_targets;           /* separate targets: the ones this rank owns, laid */
    void *target_points;               /* out like `points` (second allocation)          */
    int32_t sep_targets;               /* 1: some rank passed separate targets, so every  */
                                       /* rank exchanged two sets (n_owned_targets may be 0) */
    int32_t target_record_len;         /* values per received target: dims, or dims + 1 with */
                                       /* radii (the radius last): bt_tree_params.target_stride */
    void *target_radii;                /* ... and the radii once more as a dense array        */
                                       /* [n_owned_targets] (third allocation), else NULL     */
    int32_t source_record_len;         /* values per received source: dims (+ 1: its refine   */
                                       /* weight, in the low 32 bits of the last value)      */
    const int32_t *refine_weights;     /* weighted jobs: [n_owned + n_owned_targets] weights  */
                                       /* of the received sources, then targets (the order of */
                                       /* bt_tree_params.refine_weights); owned by the context */
    const int64_t *top_box_arrive, *top_box_stay;   /* device tables for bt_tree_params with   */
                                       /* extents (levels 0..top_level), else NULL           */
    /* particle identity: global user ids number the ranks' chunks in rank order (the input */
    /* a single GPU would be given is their concatenation); particle i of this rank's chunk  */
    /* is global particle source_chunk_offset + i (targets likewise)                         */
    int64_t source_chunk_offset, target_chunk_offset;
    int64_t n_global_sources, n_global_targets;
    int64_t n_sent_sources, n_sent_targets;   /* particles of the chunk that went to another rank: */
                                       /* a bt_mgpu_route of elem_size bytes moves that many   */
                                       /* elements off this GPU                                */
} bt_mgpu_shard;

/* the one-sweep partition (bt_partition_pack) keeps one run per owner in LDS: at most this
 * many ranks; bt_mgpu_exchange returns BT_ERR_UNSUPPORTED beyond (one node has 8) */
#define BT_MGPU_MAX_RANKS 256

/* Steps 1-3: global root box, ownership cells, the particles this rank owns.  The call
 * returns when the shard is complete -- on a stream-ordered context (bt_set_stream_ordered)
 * when the payload exchange is QUEUED on the context's stream: sizes, root box and pointers
 * are final, the received particles are there for whatever is queued on that stream next
 * (the host waits once, for the all-reduced cell histogram).  Feed the shard to bt_tree_build with sources[ax] =
 * (char *) points + ax * sizeof(coord), source_stride = dims, the root box, top_level and
 * top_cell_prefix.  The context remembers the plan (the top of the global tree, the owner
 * of every cell) for bt_mgpu_number and bt_mgpu_let_build. */
int bt_mgpu_exchange(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_params *params,
                     bt_mgpu_shard *out);

/* Particle identity across the exchange (what the reference keeps as
 * src_idx / tgt_idx, boxtree/distributed/__init__.py:238-248, and uses to hand out source
 * weights and collect potentials, distributed/calculation.py:86-142).  The received records carry
 * no ids: the exchange keeps its stable send plan instead -- the owner of every chunk particle
 * (one byte) and the scanned per-tile counts, from which the partition kernel's own ranking
 * reproduces every particle's place -- and bt_mgpu_route moves ANY per-particle array of 4- or
 * 8-byte elements over that plan (one all-to-all-v of elem_size bytes per particle that changes
 * rank; nothing rides on the coordinate exchange):
 *   BT_ROUTE_TO_OWNERS   in: [n] in the order of this rank's chunk (bt_mgpu_params.coords /
 *                        .targets)  ->  out: [n_owned] in the order of the receive buffer, the
 *                        order bt_tree_arrays.user_source_ids / sorted_target_ids of the
 *                        rank's tree refer to;
 *   BT_ROUTE_TO_CALLERS  the inverse: [n_owned] -> [n].
 * particle_set: 0 sources, 1 separate targets.  Collective: every rank of the exchange calls it
 * with the same particle_set, direction and elem_size, after bt_mgpu_exchange and before the
 * next one on this context.  Source weights given in the caller's order reach tree order by
 * TO_OWNERS and a gather through user_source_ids; potentials in tree order go back by a gather
 * through sorted_target_ids and TO_CALLERS. */
#define BT_ROUTE_TO_OWNERS 0
#define BT_ROUTE_TO_CALLERS 1
int bt_mgpu_route(bt_context *ctx, bt_mgpu_comm *comm, int particle_set, int direction, int elem_size,
                  const void *in, void *out);
/* The global user id (bt_mgpu_shard.source_chunk_offset + index in its chunk) of every OWNED
 * particle, in receive-buffer order: ids [n_owned] of id_size 4 (int32, the reference's
 * particle_id_t; BT_ERR_UNSUPPORTED past 2^31 - 1 particles) or 8 bytes.  ids[user_source_ids[j]]
 * of the rank's tree is entry source_offset + j of the single-GPU tree's user_source_ids
 * (bt_mgpu_numbering).  Same collective rules as bt_mgpu_route (it is one). */
int bt_mgpu_global_ids(bt_context *ctx, bt_mgpu_comm *comm, int particle_set, int id_size, void *ids);

/* Device time of the last exchange's payload all-to-all-v on this context, in milliseconds
 * (waits for it if it is still running). */
int bt_mgpu_exchange_time(bt_context *ctx, float *a2a_ms);

/* The host part of the exchange, a pure function of the all-reduced level-top_level
 * cell histogram: owner rank of every cell (contiguous Morton ranges balanced by
 * particle count, leaves of the global top tree kept whole) and the exclusive prefix
 * sums of the histogram [ncells + 1] (may be NULL). */
int bt_mgpu_plan(int dims, int top_level, int64_t max_particles_in_box, int nranks,
                 const int64_t *global_hist, int32_t *owner_of_cell, int64_t *cell_prefix);

/* The same for particles with extents: stay_table counts, per box of levels 0..top_level (box
 * (level, Morton path) at index (C^level - 1) / (C - 1) + path, C = 2^dims), the particles that
 * stay in it; global_hist counts such a particle at the first cell under its box.  A box splits
 * iff arrivals - stayers > max_particles_in_box (tree_build_kernels.py:569-591).  box_arrive
 * (same indexing) and box_split (bit 0: the box exists, bit 1: it splits) may be NULL. */
int bt_mgpu_plan_ext(int dims, int top_level, int64_t max_particles_in_box, int nranks,
                     const int64_t *global_hist, const int64_t *stay_table, int32_t *owner_of_cell,
                     int64_t *cell_prefix, int64_t *box_arrive, uint8_t *box_split);

/* the tree a rank built from its shard (bt_tree_sizes / bt_tree_arrays of that build) */
typedef struct {
    int32_t dims, coord_kind;
    int64_t nboxes, aligned_nboxes;
    int32_t nlevels;
    const int32_t *level_start_box_nrs;   /* HOST [nlevels + 1]                     */
    const void *box_centers;              /* device [dims][aligned_nboxes]          */
    const uint8_t *box_levels;            /* device [nboxes]                        */
    const uint8_t *box_flags;             /* device [nboxes] (bt_mgpu_let_build)    */
    int64_t nsources, ntargets;
    /* targets with extents (bt_mgpu_let_build; NULL otherwise): what the traversal reads
     * of a tree whose targets have extents, laid out as in the tree export */
    const void *box_target_bounding_box_min, *box_target_bounding_box_max;  /* [dims][aligned_nboxes] */
    const int32_t *box_source_counts_cumul;                                  /* [nboxes] */
    /* Optional: bt_tree_arrays.box_subtree_sizes of the local tree ([nboxes]).  If EVERY rank
     * passes it, the LET comes with the subtree sizes bt_trav_params.box_subtree_sizes takes
     * (own boxes keep theirs, the owners send those of the halo boxes, the shared top boxes are
     * summed) and the traversal saves its sweep over the levels. */
    const int32_t *box_subtree_sizes;
} bt_mgpu_local_tree;

/* Step 5: where the rank's tree sits in the global one -- the tree a single GPU builds
 * from the concatenated input.  Boxes of levels <= top_level are shared between ranks and
 * numbered by Morton path (from the plan); deeper levels are the concatenation of the
 * ranks' level slices (ranks own ascending Morton ranges).  One all-gather of the
 * per-level box counts.  box_ids [nboxes] (device, out): global number of every local
 * box.  Renumbered with these, the per-rank arrays are slices of the single-GPU tree
 * (tests/test_gpu_mgpu.py). */
typedef struct {
    int32_t nlevels;                                   /* of the global tree          */
    int32_t level_start_box_nrs[BT_MAX_LEVELS + 2];     /* global, [nlevels + 1] valid  */
    int32_t deep_base[BT_MAX_LEVELS + 1];   /* global number of this rank's first box of a level > top_level */
    int64_t nboxes, nsources, ntargets;                /* global totals               */
    int64_t source_offset, target_offset;   /* global tree-order index of local particle 0 */
} bt_mgpu_numbering;
int bt_mgpu_number(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                   int32_t *box_ids, bt_mgpu_numbering *out);

/* Step 6: the local essential tree -- the boxes a rank needs for the interaction lists
 * of its own boxes: the shared top levels, its own subtrees, and the subtrees of other
 * ranks' cells within well_sep_is_n_away cells of its own (one all-to-all-v of 16-byte
 * box records between neighbours).  Boxes are level-major, Morton order within a level;
 * the result is the global tree restricted to these boxes.  _build returns the sizes,
 * _export fills caller-allocated arrays laid out as in the tree export: box_child_ids
 * [2^d][aligned], box_centers [d][aligned]; global_box_ids [nboxes] and
 * target_boxes_mask [nboxes] (1: this rank builds the lists of the box) may be NULL.
 * Hand the arrays to bt_traversal_build with target_boxes_mask and active_level_ranges. */
typedef struct {
    int64_t nboxes, aligned_nboxes;
    int32_t nlevels;
    int32_t level_start_box_nrs[BT_MAX_LEVELS + 2];
    int32_t active_level_ranges[BT_MAX_LEVELS + 1][2];  /* per level: [begin, end) of the rank's boxes */
    int64_t halo_boxes_sent, halo_boxes_received;
    int64_t loopback_records, loopback_mismatches;      /* self-loopback echo of the halo records */
    int32_t has_subtree_sizes;            /* every rank passed box_subtree_sizes: _export has them */
} bt_mgpu_let_sizes;
typedef struct {
    void *box_centers;
    int32_t *box_parent_ids, *box_child_ids;
    uint8_t *box_levels, *box_flags;
    int32_t *global_box_ids;
    int8_t *target_boxes_mask;            /* 1: this rank builds the lists of the box; 2 (targets  */
                                          /* with extents): a shared internal box whose own        */
                                          /* targets another rank holds -- the lists it has as a   */
                                          /* parent of target boxes, not those of its own targets  */
    /* targets with extents (the local tree had the three arrays): of every box of the LET --
     * the owners send them with the halo boxes, the shared top boxes get the union over the
     * ranks (an all-reduce) and their source counts from the plan */
    void *box_target_bounding_box_min, *box_target_bounding_box_max;   /* [dims][aligned_nboxes] */
    int32_t *box_source_counts_cumul;                                   /* [nboxes] */
    int32_t *box_subtree_sizes;           /* [nboxes] or NULL; filled if has_subtree_sizes */
} bt_mgpu_let_arrays;
int bt_mgpu_let_build(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                      const int32_t *box_ids, const bt_mgpu_numbering *numbering,
                      int well_sep_is_n_away, bt_mgpu_let_sizes *out);
int bt_mgpu_let_export(bt_context *ctx, const bt_mgpu_let_arrays *arrays);

/* Morton path (x most significant in every digit, level*dims bits) of every box,
 * from its centre and level. */
int bt_box_morton_paths(bt_context *ctx, int dims, int coord_kind, int64_t nboxes,
                        int64_t aligned_nboxes, const void *box_centers, const uint8_t *box_levels,
                        const double *bbox_min, double root_extent, uint64_t *paths);

/* Links a tree given as Morton paths: boxes are level-major (level_start_box_nrs,
 * HOST, [nlevels+1]) and ascending by path within a level.  Writes box_parent_ids
 * [nboxes], box_child_ids [2^d][aligned] (0 where the child is not in the set) and
 * box_centers [d][aligned] by the builder's own arithmetic (root centre, then
 * parent centre +/- half the child's size).  BT_ERR_INVALID if a box has no parent. */
int bt_let_build(bt_context *ctx, int dims, int coord_kind, int nlevels,
                 const int32_t *level_start_box_nrs, const uint64_t *paths, int64_t aligned_nboxes,
                 const double *bbox_min, const double *bbox_max, double root_extent,
                 int32_t *box_parent_ids, int32_t *box_child_ids, void *box_centers);

/* ---- work partition and local trees of the distributed FMM evaluation
 *      (boxtree/distributed/partition.py, local_tree.py, calculation.py) ---- */

/* get_box_ids_dfs_order (distributed/partition.py:39-57): preorder in which the
 * reference's explicit stack visits the boxes, i.e. children in DESCENDING Morton
 * child number.  level_start_box_nrs is HOST memory [nlevels+1]; dfs_order [nboxes]. */
int bt_dfs_order(bt_context *ctx, int nchildren, int nlevels, const int32_t *level_start_box_nrs,
                 int64_t nboxes, int64_t aligned_nboxes, const int32_t *box_child_ids,
                 int32_t *dfs_order);

/* partition_work (distributed/partition.py:60-121): cuts the depth-first order
 * into nranks consecutive segments of about equal cost.  Segment s ends at the
 * first box whose running cost exceeds (s+1)*total/nranks, one box ends at most
 * one segment, the last rank takes the rest.  cost_per_box is f64 [nboxes] on the
 * device; segments is HOST memory [nranks][2] = [start, end) in depth-first
 * positions (ranks the loop never reaches get [nboxes, nboxes)). */
int bt_partition_work(bt_context *ctx, int64_t nboxes, const int32_t *dfs_order,
                      const double *cost_per_box, int nranks, int32_t *segments);

/* get_ancestor_boxes_mask (distributed/partition.py:167-188): ancestors[b] = 1 iff
 * b is a proper ancestor of a box in the mask.  Both [nboxes], int8. */
int bt_ancestor_mask(bt_context *ctx, int64_t nboxes, const int32_t *box_parent_ids,
                     const int8_t *boxes_mask, int8_t *ancestors);

/* add_interaction_list_boxes_kernel (distributed/partition.py:134-164): for every
 * row i with (mask_a | mask_b)[box_list[i]] set, out_mask[lists[j]] = 1 for the
 * row's entries.  mask_b may be NULL; out_mask is updated, not cleared. */
int bt_mark_list_boxes(bt_context *ctx, int64_t nrows, const int32_t *box_list,
                       const int8_t *mask_a, const int8_t *mask_b, const int32_t *starts,
                       const int32_t *lists, int8_t *out_mask);

/* construct_local_particles_and_lists (distributed/local_tree.py:198-283): the
 * particles a rank keeps are those owned (counts_nonchild) by boxes in box_mask.
 * Outputs: local starts / counts_nonchild / counts_cumul [nboxes], particle_idx
 * (capacity nparticles, *nlocal valid, ascending global tree-order indices). */
int bt_local_particles(bt_context *ctx, int64_t nboxes, int64_t nparticles, const int8_t *box_mask,
                       const int32_t *box_particle_starts,
                       const int32_t *box_particle_counts_nonchild,
                       const int32_t *box_particle_counts_cumul, int32_t *local_starts,
                       int32_t *local_counts_nonchild, int32_t *local_counts_cumul,
                       int32_t *particle_idx, int64_t *nlocal);

/* modify_target_flags_kernel (distributed/local_tree.py:155-185): rebuilds
 * BT_BOX_IS_TARGET_BOX / BT_BOX_HAS_TARGET_CHILD_BOXES from the local counts. */
int bt_modify_target_flags(bt_context *ctx, int64_t nboxes, const int32_t *counts_nonchild,
                           const int32_t *counts_cumul, uint8_t *box_flags);

/* box_to_user_rank CSR (distributed/local_tree.py:368-399): masks is [nranks][nboxes]
 * int8 (rank-major, as gathered); starts [nboxes+1]; with lists == NULL only starts
 * and *nentries are produced, otherwise lists [*nentries] holds each box's user
 * ranks ascending. */
int bt_box_to_user_ranks(bt_context *ctx, int nranks, int64_t nboxes, const int8_t *masks,
                         int32_t *starts, int32_t *lists, int64_t *nentries);

/* find_boxes_used_by_subrange + compaction (distributed/calculation.py:191-262,
 * 335-352): ascending list of the boxes with contributing[b] != 0 that have a user
 * rank in [rank_lo, rank_hi).  boxes has capacity nboxes. */
int bt_boxes_used_by_ranks(bt_context *ctx, int64_t nboxes, const int8_t *contributing,
                           int rank_lo, int rank_hi, const int32_t *box_to_user_rank_starts,
                           const int32_t *box_to_user_rank_lists, int32_t *boxes, int64_t *n);

#ifdef __cplusplus
}
#endif
#endif /* BOXTREE_HIP_H */
