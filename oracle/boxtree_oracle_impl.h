/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle: a literal restatement of inducer/boxtree's tree-build
 * and traversal kernels (reference = /root/reference, version 2024.10).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED at the bit level: the reference's OpenCL kernels cannot be
 * imported or compiled in this environment (pyopencl/arraycontext/pytools/mako
 * absent, no OpenCL device) and its test-suite holds no golden vectors, so box
 * numbering, particle order within a leaf and list order follow the cited
 * reference lines, not reference output.  What the reference CAN do here it does:
 * its own test functions run against this oracle (tests/test_reference_suite.py,
 * 93 parameterisations) and its pure-Python modules (drive_fmm + constant-one
 * wrangler, cost-model loops, partition, ...) run on this oracle's output
 * (tests/golden/make_reference_vectors.py, tests/test_reference_vectors.py).
 *
 * This header is a "template": it is included once per coordinate type with
 *   COORD_T   float | double
 *   SFX(name) name##_f32 | name##_f64
 *   COORD_SQRT sqrtf | sqrt
 *   COORD_EPS  FLT_EPSILON | DBL_EPSILON
 * Compile with -ffp-contract=off: every a*b+c below is two roundings.
 */

/* ------------------------------------------------------------------------ */
/* data types                                                               */
/* ------------------------------------------------------------------------ */

#ifndef ORC_COMMON_DEFS
#define ORC_COMMON_DEFS

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include <float.h>

#define ORC_MAXDIM 3
#define ORC_MAXC 8

/* The same source builds the sequential oracle (liboracle.so) and, with -fopenmp,
 * the all-core CPU baseline (liboracle_omp.so).  Loops over particles / boxes /
 * list objects are split into one contiguous chunk per thread; chunk results are
 * combined in chunk order, so both builds produce identical arrays
 * (tests/test_oracle_openmp.py). */
#ifdef _OPENMP
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#define ORC_NTHREADS() omp_get_max_threads()
#define ORC_TID() omp_get_thread_num()
/* the team a parallel region actually got (inside the region) */
#define ORC_TEAM() omp_get_num_threads()
#define ORC_PRAGMA(x) _Pragma(#x)
/* Regions that size per-thread arrays by the REQUESTED team (num_threads(nth)) must get it:
 * with a smaller team (OMP_THREAD_LIMIT, cgroup limits, nested regions; dynamic teams are
 * switched off in orc_set_num_threads and at load time) the chunks of the missing threads
 * would silently never be processed.  Fail loudly instead. */
#define ORC_CHECK_TEAM(nth)                                                                   \
    do {                                                                                      \
        if (omp_get_num_threads() != (nth)) {                                                 \
            fprintf(stderr, "boxtree oracle: asked for %d OpenMP threads, the runtime gave %d\n", \
                    (int) (nth), omp_get_num_threads());                                      \
            abort();                                                                          \
        }                                                                                     \
    } while (0)
#else
#define ORC_NTHREADS() 1
#define ORC_TID() 0
#define ORC_TEAM() 1
#define ORC_PRAGMA(x)
#define ORC_CHECK_TEAM(nth) do { } while (0)
#endif
/* chunk [lo_, hi_) of [0, n) of thread t out of nth */
#define ORC_CHUNK(n, t, nth, lo_, hi_) \
    const int64_t lo_ = (int64_t) (n) * (t) / (nth), hi_ = (int64_t) (n) * ((t) + 1) / (nth)

enum { ORC_OK = 0, ORC_ERR_MAX_LEVELS = 1, ORC_ERR_ALLOC = 2, ORC_ERR_INTERNAL = 3 };
enum { ORC_NORM_NONE = 0, ORC_NORM_LINF = 1, ORC_NORM_L2 = 2 };
enum { ORC_KIND_ADAPTIVE = 0, ORC_KIND_ADAPTIVE_LEVEL_RESTRICTED = 1, ORC_KIND_NON_ADAPTIVE = 2 };
enum { ORC_CRIT_STATIC_LINF = 0, ORC_CRIT_PRECISE_LINF = 1, ORC_CRIT_STATIC_L2 = 2 };

/* box flags: boxtree/tree.py:109-145 */
#define BOX_IS_SOURCE_BOX 1
#define BOX_IS_TARGET_BOX 2
#define BOX_HAS_SOURCE_CHILD_BOXES 4
#define BOX_HAS_TARGET_CHILD_BOXES 8
#define BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES 12

/* morton_counts_t: boxtree/tree_build_kernels.py:158-189
 * (nonchild_srcntgts is only present in the reference struct with extents; it
 * is simply kept at zero here otherwise) */
typedef struct {
    int32_t nonchild_srcntgts;
    int32_t pcnt[ORC_MAXC];
    int32_t pwt[ORC_MAXC];
} orc_mc_t;

/* my_add_sat: tree_build_kernels.py:270-274 */
static inline int32_t orc_add_sat(int32_t a, int32_t b)
{
    int64_t r = (int64_t) a + b;
    return (r > INT_MAX) ? INT_MAX : (int32_t) r;
}

/* growable int32 vector used to emulate pyopencl's ListOfListsBuilder */
typedef struct { int32_t *data; int64_t n, cap; } orc_ivec;

static inline int orc_ivec_push(orc_ivec *v, int32_t x)
{
    if (v->n == v->cap) {
        int64_t nc = v->cap ? 2 * v->cap : 1024;
        int32_t *nd = (int32_t *) realloc(v->data, (size_t) nc * sizeof(int32_t));
        if (!nd) return 1;
        v->data = nd; v->cap = nc;
    }
    v->data[v->n++] = x;
    return 0;
}

/* CSR list as produced by pyopencl.algorithm.ListOfListsBuilder ("BuiltList").
 * With eliminate_empty: starts has num_nonempty_lists+1 entries and
 * nonempty_indices / compressed_indices are filled
 * (boxtree/array_context.py:222-238, traversal.py:2211-2212). */
typedef struct {
    int64_t n_objects;
    int64_t count;              /* total number of list entries */
    int32_t *starts;
    int32_t *lists;
    int64_t num_nonempty_lists; /* -1 if not compressed */
    int32_t *nonempty_indices;
    int32_t *compressed_indices;
} orc_built_list;

void orc_free(void *p);

#endif /* ORC_COMMON_DEFS */

/* ------------------------------------------------------------------------ */
/* per-coordinate-type section                                              */
/* ------------------------------------------------------------------------ */

typedef struct {
    int32_t dims;
    int32_t sources_are_targets;
    int64_t nsrcntgts;
    int64_t nsources;          /* == nsrcntgts if sources_are_targets */
    const COORD_T *srcntgts[ORC_MAXDIM];  /* merged, tree_build.py:328-388 */
    const COORD_T *srcntgt_radii;         /* NULL unless extents */
    int32_t sources_have_extent, targets_have_extent;
    const int32_t *refine_weights;        /* [nsrcntgts] */
    int32_t max_leaf_refine_weight;
    COORD_T bbox_min[ORC_MAXDIM];         /* bbox struct, after tree_build.py:462-476 */
    COORD_T bbox_max[ORC_MAXDIM];
    COORD_T root_extent;
    COORD_T stick_out_factor;
    int32_t extent_norm;                  /* ORC_NORM_* ; NONE if no extents */
    int32_t kind;
    int32_t skip_prune;
    int32_t nlevels_max;                  /* tree_build.py:622 */
} SFX(orc_tree_in);

typedef struct {
    int32_t status;
    int32_t nlevels;
    int64_t nboxes;
    int64_t aligned_nboxes;
    int32_t *level_start_box_nrs;         /* [nlevels+1] */

    int32_t *user_source_ids;             /* [nsources] */
    int32_t *sorted_target_ids;           /* [ntargets] */
    COORD_T *sources[ORC_MAXDIM];
    COORD_T *targets[ORC_MAXDIM];         /* == sources if sources_are_targets */
    COORD_T *source_radii, *target_radii; /* NULL unless extents */

    int32_t *box_source_starts, *box_source_counts_nonchild, *box_source_counts_cumul;
    int32_t *box_target_starts, *box_target_counts_nonchild, *box_target_counts_cumul;
    int32_t *box_parent_ids;              /* [nboxes] */
    int32_t *box_child_ids;               /* [2^d, aligned_nboxes] */
    COORD_T *box_centers;                 /* [d, aligned_nboxes] */
    uint8_t *box_levels;                  /* [nboxes] */
    uint8_t *box_flags;                   /* [nboxes] */
    COORD_T *box_source_bounding_box_min, *box_source_bounding_box_max; /* [d, aligned] */
    COORD_T *box_target_bounding_box_min, *box_target_bounding_box_max;
} SFX(orc_tree_out);

/* ---- bounding box: boxtree/bounding_box.py:54-122 ----------------------- */

void SFX(orc_bbox)(int dims, int64_t n, const COORD_T *const *coords,
                   const COORD_T *radii, COORD_T *out_min, COORD_T *out_max)
{
    for (int d = 0; d < dims; ++d) {
        COORD_T mn = COORD_MAX, mx = -COORD_MAX;   /* bbox_neutral() :66-75 */
        ORC_PRAGMA(omp parallel)
        {
            COORD_T tmn = COORD_MAX, tmx = -COORD_MAX;
            ORC_CHUNK(n, ORC_TID(), ORC_TEAM(), lo_, hi_);
            for (int64_t i = lo_; i < hi_; ++i) {
                COORD_T r = radii ? radii[i] : 0;
                COORD_T lo = coords[d][i] - r, hi = coords[d][i] + r;  /* :77-90 */
                tmn = (lo < tmn) ? lo : tmn;                           /* :92-99 */
                tmx = (hi > tmx) ? hi : tmx;
            }
            ORC_PRAGMA(omp critical)
            { mn = (tmn < mn) ? tmn : mn; mx = (tmx > mx) ? tmx : mx; }
        }
        out_min[d] = mn; out_max[d] = mx;
    }
}

/* ---- K3: scan_t_from_particle, tree_build_kernels.py:308-470 ------------ */

static inline orc_mc_t SFX(orc_scan_t_from_particle)(
        const SFX(orc_tree_in) *in, int64_t i, int particle_level,
        int8_t *morton_nrs, const int32_t *user_srcntgt_ids)
{
    const int dims = in->dims;
    const int have_extent = in->extent_norm != ORC_NORM_NONE;
    int32_t user_srcntgt_id = user_srcntgt_ids[i];

    /* :328-329 */
    COORD_T next_level_box_size_factor =
        ((COORD_T) 1) / ((COORD_T) (1U << (1 + particle_level)));

    int stop_srcntgt_descent = 0;
    COORD_T srcntgt_radius = 0;
    if (have_extent)
        srcntgt_radius = in->srcntgt_radii[user_srcntgt_id];

    const COORD_T one_half = ((COORD_T) 1) / 2;
    /* :342-346  "(1. + stick_out_factor) * one_half" -- the literal is double */
    const COORD_T box_radius_factor = (COORD_T) (
        (1. + (have_extent ? (double) in->stick_out_factor : 0.)) * (double) one_half);

    COORD_T global_extent[ORC_MAXDIM], srcntgt[ORC_MAXDIM], next_center[ORC_MAXDIM];
    unsigned bits[ORC_MAXDIM];

    for (int ax = 0; ax < dims; ++ax) {
        COORD_T global_min = in->bbox_min[ax];                         /* :358 */
        global_extent[ax] = in->bbox_max[ax] - global_min;             /* :359 */
        srcntgt[ax] = in->srcntgts[ax][user_srcntgt_id];               /* :360 */

        /* :374-376 */
        bits[ax] = (unsigned) (
            ((srcntgt[ax] - global_min) / global_extent[ax])
            * (COORD_T) (1U << (1 + particle_level)));

        /* :380-384 */
        next_center[ax] =
            global_min
            + global_extent[ax]
            * ((COORD_T) bits[ax] + one_half)
            * next_level_box_size_factor;
    }

    if (in->extent_norm == ORC_NORM_LINF) {
        for (int ax = 0; ax < dims; ++ax) {
            /* :390-393 */
            const COORD_T sor = box_radius_factor * global_extent[ax]
                * next_level_box_size_factor;
            /* :396-403 */
            stop_srcntgt_descent = stop_srcntgt_descent ||
                (srcntgt[ax] + srcntgt_radius >= next_center[ax] + sor);
            stop_srcntgt_descent = stop_srcntgt_descent ||
                (srcntgt[ax] - srcntgt_radius < next_center[ax] - sor);
        }
    } else if (in->extent_norm == ORC_NORM_L2) {
        /* :408-428 */
        COORD_T sor = box_radius_factor * global_extent[0] * next_level_box_size_factor;
        COORD_T sumsq = 0;
        for (int ax = 0; ax < dims; ++ax) {
            COORD_T t = (srcntgt[ax] - next_center[ax]) * (srcntgt[ax] - next_center[ax]);
            sumsq = (ax == 0) ? t : sumsq + t;
        }
        COORD_T dist = COORD_SQRT(sumsq) + srcntgt_radius;
        stop_srcntgt_descent = stop_srcntgt_descent ||
            (dist * dist >= dims * sor * sor);
    }

    /* :441-445 */
    int level_morton_number = 0;
    for (int iax = 0; iax < dims; ++iax)
        level_morton_number |= (int) (bits[iax] & 1U) << (dims - 1 - iax);

    if (have_extent && stop_srcntgt_descent)
        level_morton_number = -1;                                      /* :448-451 */

    orc_mc_t result;
    memset(&result, 0, sizeof(result));
    if (have_extent)
        result.nonchild_srcntgts = (level_morton_number == -1);        /* :456 */
    for (int mnr = 0; mnr < (1 << dims); ++mnr) {
        result.pcnt[mnr] = (level_morton_number == mnr);               /* :460 */
        result.pwt[mnr] = (level_morton_number == mnr)
            ? in->refine_weights[user_srcntgt_id] : 0;                 /* :464-465 */
    }
    morton_nrs[i] = (int8_t) level_morton_number;                      /* :467 */
    return result;
}

/* scan_t_add: tree_build_kernels.py:277-302 */
static inline orc_mc_t SFX(orc_scan_t_add)(orc_mc_t a, orc_mc_t b, int across, int C)
{
    if (!across) {
        b.nonchild_srcntgts += a.nonchild_srcntgts;
        for (int m = 0; m < C; ++m) b.pcnt[m] = a.pcnt[m] + b.pcnt[m];
        for (int m = 0; m < C; ++m) b.pwt[m] = orc_add_sat(a.pwt[m], b.pwt[m]);
    }
    return b;
}

/* get_count: tree_build_kernels.py:732-739 */
static inline int32_t SFX(orc_get_count)(const orc_mc_t *c, int morton_nr)
{
    if (morton_nr == -1) return c->nonchild_srcntgts;
    return c->pcnt[morton_nr];
}

#define ORC_GROW(ptr, type, oldn, newn) do { \
        type *np_ = (type *) calloc((size_t) (newn) ? (size_t) (newn) : 1, sizeof(type)); \
        if (!np_) { status = ORC_ERR_ALLOC; goto done; } \
        if (ptr) { memcpy(np_, ptr, (size_t) (oldn) * sizeof(type)); free(ptr); } \
        ptr = np_; } while (0)

/* is_adjacent_or_overlapping as instantiated for LEVEL_RESTRICT_TPL
 * (traversal.py:279-318 via tbk:960-965; target = walk box, source = my box) */
static inline int SFX(orc_lr_is_adj)(int dims, COORD_T root_extent,
        const COORD_T *target_center, int target_level,
        const COORD_T *source_center, int source_level)
{
    COORD_T target_rad = (root_extent * 1 / (COORD_T) (1 << (target_level + 1)));
    COORD_T source_rad = (root_extent * 1 / (COORD_T) (1 << (source_level + 1)));
    COORD_T rad_sum = ((2 * ((COORD_T) 1 - 1) + 1) * target_rad + source_rad);
    COORD_T slack = rad_sum + ((target_rad < source_rad) ? target_rad : source_rad);
    COORD_T l_inf_dist = 0;
    for (int i = 0; i < dims; ++i) {
        COORD_T d = target_center[i] - source_center[i];
        d = (d < 0) ? -d : d;
        l_inf_dist = (d > l_inf_dist) ? d : l_inf_dist;
    }
    return l_inf_dist <= slack;
}

/* ---- TreeBuilder.__call__ from "allocate data" on: tree_build.py:512-1878 - */

int SFX(orc_tree_build)(const SFX(orc_tree_in) *in, SFX(orc_tree_out) *out)
{
    int status = ORC_OK;
    const int dims = in->dims;
    const int C = 1 << dims;
    const int64_t N = in->nsrcntgts;
    const int have_extent = in->extent_norm != ORC_NORM_NONE;
    const int adaptive = in->kind != ORC_KIND_NON_ADAPTIVE;
    const int level_restrict = in->kind == ORC_KIND_ADAPTIVE_LEVEL_RESTRICTED;  /* :606-611 */
    const int32_t max_w = in->max_leaf_refine_weight;

    memset(out, 0, sizeof(*out));

    /* per-particle state, tree_build.py:516-532 */
    orc_mc_t *morton_bin_counts = NULL;
    int8_t *morton_nrs = NULL, *box_start_flags = NULL;
    int32_t *srcntgt_box_ids = NULL, *user_srcntgt_ids = NULL;
    int32_t *new_user_srcntgt_ids = NULL, *new_srcntgt_box_ids = NULL;

    /* per-box state, tree_build.py:557-611 */
    int64_t nboxes_alloc = 0;
    int32_t *split_box_ids = NULL, *box_srcntgt_starts = NULL, *box_parent_ids = NULL;
    int32_t *box_srcntgt_counts_cumul = NULL, *box_has_children = NULL;
    int32_t *force_split_box = NULL;     /* :606-611 (level restriction only) */
    int32_t *box_child_ids[ORC_MAXC] = {0};
    COORD_T *box_centers[ORC_MAXDIM] = {0};
    uint8_t *box_levels = NULL;
    orc_mc_t *box_morton_bin_counts = NULL;
    int32_t *box_srcntgt_counts_nonchild = NULL;

    int32_t *level_start_box_nrs = NULL, *level_used_box_counts = NULL;
    int32_t *new_level_used_box_counts = NULL;
    int32_t *src_box_id = NULL, *dst_box_id = NULL;
    int32_t *source_numbers = NULL, *srcntgt_target_ids = NULL;

    morton_bin_counts = (orc_mc_t *) calloc((size_t) (N ? N : 1), sizeof(orc_mc_t));
    morton_nrs = (int8_t *) calloc((size_t) (N ? N : 1), 1);
    box_start_flags = (int8_t *) calloc((size_t) (N ? N : 1), 1);
    srcntgt_box_ids = (int32_t *) calloc((size_t) (N ? N : 1), 4);
    user_srcntgt_ids = (int32_t *) calloc((size_t) (N ? N : 1), 4);
    new_user_srcntgt_ids = (int32_t *) calloc((size_t) (N ? N : 1), 4);
    new_srcntgt_box_ids = (int32_t *) calloc((size_t) (N ? N : 1), 4);
    level_start_box_nrs = (int32_t *) calloc((size_t) in->nlevels_max + 2, 4);
    level_used_box_counts = (int32_t *) calloc((size_t) in->nlevels_max + 2, 4);
    new_level_used_box_counts = (int32_t *) calloc((size_t) in->nlevels_max + 2, 4);
    if (!morton_bin_counts || !morton_nrs || !box_start_flags || !srcntgt_box_ids
            || !user_srcntgt_ids || !new_user_srcntgt_ids || !new_srcntgt_box_ids
            || !level_start_box_nrs || !level_used_box_counts
            || !new_level_used_box_counts) {
        status = ORC_ERR_ALLOC; goto done;
    }

    for (int64_t i = 0; i < N; ++i) user_srcntgt_ids[i] = (int32_t) i;  /* :395 */

    int64_t total_refine_weight = 0;                                   /* :446 */
    for (int64_t i = 0; i < N; ++i) total_refine_weight += in->refine_weights[i];

#define ORC_ENSURE_BOXES(need) do { \
        if ((need) > nboxes_alloc) { \
            int64_t na_ = nboxes_alloc ? nboxes_alloc : 64; \
            while (na_ < (need)) na_ *= 2; \
            ORC_GROW(split_box_ids, int32_t, nboxes_alloc, na_); \
            ORC_GROW(box_srcntgt_starts, int32_t, nboxes_alloc, na_); \
            ORC_GROW(box_parent_ids, int32_t, nboxes_alloc, na_); \
            ORC_GROW(box_srcntgt_counts_cumul, int32_t, nboxes_alloc, na_); \
            ORC_GROW(box_has_children, int32_t, nboxes_alloc, na_); \
            ORC_GROW(force_split_box, int32_t, nboxes_alloc, na_); \
            ORC_GROW(box_levels, uint8_t, nboxes_alloc, na_); \
            ORC_GROW(box_morton_bin_counts, orc_mc_t, nboxes_alloc, na_); \
            for (int m_ = 0; m_ < C; ++m_) \
                ORC_GROW(box_child_ids[m_], int32_t, nboxes_alloc, na_); \
            for (int d_ = 0; d_ < dims; ++d_) \
                ORC_GROW(box_centers[d_], COORD_T, nboxes_alloc, na_); \
            nboxes_alloc = na_; \
        } } while (0)

    ORC_ENSURE_BOXES(64);

    /* root box: tree_build.py:585-618 */
    for (int d = 0; d < dims; ++d)
        box_centers[d][0] = in->bbox_min[d] + (in->bbox_max[d] - in->bbox_min[d]) / 2;
    box_srcntgt_counts_cumul[0] = (int32_t) N;
    box_parent_ids[0] = 0;

    /* level loop: tree_build.py:653-1276 */
    int nlev_starts = 2;                 /* len(level_start_box_nrs) */
    level_start_box_nrs[0] = 0; level_start_box_nrs[1] = 1;
    int nlev_used = 1;                   /* len(level_used_box_counts) */
    level_used_box_counts[0] = 1;
    int32_t have_oversize_split_box = 0;

    int level = (total_refine_weight > max_w) ? 1 : 0;                 /* :676 */
    int final_level_restrict_iteration = 0;                            /* :695 */

    while (level) {
        if (level + 1 >= in->nlevels_max) {                            /* :705-709 */
            status = ORC_ERR_MAX_LEVELS; goto done;
        }

        /* K3 morton_count_scan over all particles (:732, tbk:1555-1572),
         * segmented by box_start_flags.  One chunk per thread: a local scan from the
         * chunk's start, then the carry of the preceding chunks is added to the items
         * before the chunk's first segment boundary (scan_t_add is associative). */
        {
            const int nth = ORC_NTHREADS();
            orc_mc_t *chunk_end = (orc_mc_t *) calloc((size_t) nth, sizeof(orc_mc_t));
            int64_t *first_boundary = (int64_t *) calloc((size_t) nth, sizeof(int64_t));
            if (!chunk_end || !first_boundary) { status = ORC_ERR_ALLOC; goto done; }
            ORC_PRAGMA(omp parallel num_threads(nth))
            {
                ORC_CHECK_TEAM(nth);
                const int t = ORC_TID();
                ORC_CHUNK(N, t, nth, lo_, hi_);
                orc_mc_t acc; memset(&acc, 0, sizeof(acc));
                int64_t fb = hi_;                      /* first boundary in the chunk */
                for (int64_t i = lo_; i < hi_; ++i) {
                    orc_mc_t item = SFX(orc_scan_t_from_particle)(
                        in, i, box_levels[srcntgt_box_ids[i]], morton_nrs, user_srcntgt_ids);
                    int seg_start = (i == 0) || box_start_flags[i];
                    if (seg_start && fb == hi_) fb = i;
                    acc = SFX(orc_scan_t_add)(acc, item, seg_start, C);
                    morton_bin_counts[i] = acc;
                }
                chunk_end[t] = acc;
                first_boundary[t] = fb;
                ORC_PRAGMA(omp barrier)
                /* carry into this chunk: the running value at the end of chunk t-1 */
                orc_mc_t carry; memset(&carry, 0, sizeof(carry));
                for (int u = 0; u < t; ++u) {
                    ORC_CHUNK(N, u, nth, ulo, uhi);
                    int had_boundary = first_boundary[u] < uhi;
                    (void) ulo;
                    carry = SFX(orc_scan_t_add)(carry, chunk_end[u], had_boundary, C);
                }
                for (int64_t i = lo_; i < fb; ++i)
                    morton_bin_counts[i] = SFX(orc_scan_t_add)(carry, morton_bin_counts[i], 0, C);
                ORC_PRAGMA(omp barrier)
                /* output statement, tbk:480-508 */
                for (int64_t i = lo_; i < hi_; ++i) {
                    const orc_mc_t a2 = morton_bin_counts[i];
                    int32_t my_id_in_my_box = -1 + a2.nonchild_srcntgts;
                    for (int m = 0; m < C; ++m) my_id_in_my_box += a2.pcnt[m];
                    int32_t current_box_id = srcntgt_box_ids[i];
                    int32_t box_srcntgt_count = box_srcntgt_counts_cumul[current_box_id];
                    if (my_id_in_my_box + 1 == box_srcntgt_count)
                        box_morton_bin_counts[current_box_id] = a2;
                }
            }
            free(chunk_end); free(first_boundary);
        }

        /* K4 split_box_id_scan over boxes above the new level (:740-759,
         * tbk:514-640), segmented by box level */
        {
            int64_t nscan = level_start_box_nrs[level];
            int32_t acc = 0;
            for (int64_t i = 0; i < nscan; ++i) {
                int blevel = box_levels[i];
                int32_t result = 0;
                if (i == level_start_box_nrs[blevel]) {                /* tbk:555-559 */
                    result += level_start_box_nrs[blevel + 1];
                    result += level_used_box_counts[blevel + 1];
                }
                const int32_t nonchild_srcntgts_in_box =
                    have_extent ? box_morton_bin_counts[i].nonchild_srcntgts : 0;
                int32_t box_refine_weight = 0;                         /* tbk:569-573 */
                for (int m = 0; m < C; ++m)
                    box_refine_weight = orc_add_sat(box_refine_weight,
                            box_morton_bin_counts[i].pwt[m]);
                int do_split;
                if (adaptive)
                    do_split = (blevel + 1 == level) && (box_refine_weight > max_w);
                else
                    do_split = (blevel + 1 == level)
                        && (box_srcntgt_counts_cumul[i] - nonchild_srcntgts_in_box >= 0);
                if (level_restrict) do_split = do_split || force_split_box[i];  /* tbk:593-595 */
                if (do_split) {                                        /* tbk:596-611 */
                    result += C;
                    box_has_children[i] = 1;
                    int32_t max_subbox = 0;
                    for (int m = 0; m < C; ++m)
                        if (box_morton_bin_counts[i].pwt[m] > max_subbox)
                            max_subbox = box_morton_bin_counts[i].pwt[m];
                    if (max_subbox > max_w) have_oversize_split_box = 1;
                }
                int seg_start = (i == 0) || (box_levels[i] != box_levels[i - 1]);
                acc = seg_start ? result : acc + result;               /* tbk:632-634 */
                split_box_ids[i] = acc;                                /* tbk:638 */
            }
        }

        /* :762-786 new_level_used_box_counts */
        new_level_used_box_counts[0] = 1;
        for (int l = 1; l < nlev_starts; ++l) {
            int32_t last_box_on_prev_level = level_start_box_nrs[l] - 1;
            new_level_used_box_counts[l] =
                split_box_ids[last_box_on_prev_level] - level_start_box_nrs[l];
        }
        /* :788-1005.  Only level restriction makes upper levels grow (children of
         * force-split boxes are appended to their level).  The reference pads the
         * levels (lr_lookbehind) so that this rarely needs a renumbering; padding
         * changes how often boxes are renumbered, never their relative order, so
         * this restatement uses none: whenever a level is too short for its new
         * used-box count all boxes are renumbered order-preservingly (:826-900,
         * :912-1005) and the level iteration is restarted (:1005 "continue"). */
        int needs_renumbering = 0;
        for (int l = 0; l + 1 < nlev_starts; ++l)
            if (new_level_used_box_counts[l]
                    > level_start_box_nrs[l + 1] - level_start_box_nrs[l])
                needs_renumbering = 1;
        if (needs_renumbering) {
            if (!level_restrict) { status = ORC_ERR_INTERNAL; goto done; }   /* :832 */
            const int nl = nlev_starts - 1;          /* == level: existing levels */
            int32_t *new_starts = (int32_t *) calloc((size_t) nl + 2, 4);
            for (int l = 0; l < nl; ++l) {
                int32_t cur = level_start_box_nrs[l + 1] - level_start_box_nrs[l];
                int32_t need = new_level_used_box_counts[l];
                new_starts[l + 1] = new_starts[l] + (need > cur ? need : cur);
            }
            const int64_t old_box_count = level_start_box_nrs[nl];
            const int64_t new_box_count = new_starts[nl];
            int32_t *dst = (int32_t *) calloc((size_t) old_box_count + 1, 4);
            for (int l = 0; l < nl; ++l) {                              /* :861-869 */
                int32_t len = level_start_box_nrs[l + 1] - level_start_box_nrs[l];
                for (int32_t j = 0; j < len; ++j)
                    dst[level_start_box_nrs[l] + j] = new_starts[l] + j;
            }
            const int64_t old_alloc = nboxes_alloc;
            int64_t na = nboxes_alloc;
            while (na < new_box_count + 1) na *= 2;
#define ORC_REMAP(arr, type, mapvals) do { \
                type *n_ = (type *) calloc((size_t) na, sizeof(type)); \
                if (!n_) { status = ORC_ERR_ALLOC; goto done; } \
                for (int64_t i_ = 0; i_ < old_box_count; ++i_) { \
                    type v_ = arr[i_]; \
                    if (mapvals) v_ = (type) dst[(int64_t) v_]; \
                    n_[dst[i_]] = v_; } \
                free(arr); arr = n_; } while (0)
            ORC_REMAP(split_box_ids, int32_t, 0);
            /* orc_mc_t is a struct: plain copy */
            {
                orc_mc_t *n_ = (orc_mc_t *) calloc((size_t) na, sizeof(orc_mc_t));
                if (!n_) { status = ORC_ERR_ALLOC; goto done; }
                for (int64_t i_ = 0; i_ < old_box_count; ++i_) n_[dst[i_]] = box_morton_bin_counts[i_];
                free(box_morton_bin_counts); box_morton_bin_counts = n_;
            }
            ORC_REMAP(force_split_box, int32_t, 0);
            ORC_REMAP(box_srcntgt_starts, int32_t, 0);
            ORC_REMAP(box_srcntgt_counts_cumul, int32_t, 0);
            ORC_REMAP(box_has_children, int32_t, 0);
            for (int d_ = 0; d_ < dims; ++d_) ORC_REMAP(box_centers[d_], COORD_T, 0);
            for (int m_ = 0; m_ < C; ++m_) ORC_REMAP(box_child_ids[m_], int32_t, 1);
            ORC_REMAP(box_parent_ids, int32_t, 1);
            {                                                           /* :975-983 */
                uint8_t *n_ = (uint8_t *) calloc((size_t) na, 1);
                if (!n_) { status = ORC_ERR_ALLOC; goto done; }
                for (int l = 0; l < nl; ++l)
                    for (int32_t j = new_starts[l]; j < new_starts[l + 1]; ++j) n_[j] = (uint8_t) l;
                free(box_levels); box_levels = n_;
            }
#undef ORC_REMAP
            (void) old_alloc;
            nboxes_alloc = na;
            for (int64_t i = 0; i < N; ++i)                             /* :985-987 */
                srcntgt_box_ids[i] = dst[srcntgt_box_ids[i]];
            for (int l = 0; l <= nl; ++l) level_start_box_nrs[l] = new_starts[l];
            free(new_starts); free(dst);
            continue;                                                   /* :1005 */
        }
        int64_t nboxes_new = (int64_t) level_start_box_nrs[nlev_starts - 1]
            + new_level_used_box_counts[nlev_starts - 1];

        ORC_ENSURE_BOXES(nboxes_new);      /* :912-1005 (contents preserved) */

        if (level_start_box_nrs[nlev_starts - 1] == nboxes_new) {      /* :1016-1025 */
            if (have_extent && !final_level_restrict_iteration) { level -= 1; break; }
            if (!final_level_restrict_iteration) { status = ORC_ERR_INTERNAL; goto done; }
        }

        /* :1029-1038 */
        level_start_box_nrs[nlev_starts++] = (int32_t) nboxes_new;
        for (int l = 0; l <= level; ++l)
            level_used_box_counts[l] = new_level_used_box_counts[l];
        nlev_used = level + 1;

        /* K5 box_splitter over all boxes (:1072, tbk:646-711) */
        ORC_PRAGMA(omp parallel for schedule(static))
        for (int64_t ibox = 0; ibox < nboxes_new; ++ibox) {
            int do_split_box = box_has_children[ibox] && (box_levels[ibox] + 1 == level);
            if (level_restrict) do_split_box = do_split_box || force_split_box[ibox];  /* tbk:651-653 */
            if (!do_split_box) continue;
            orc_mc_t bmc = box_morton_bin_counts[ibox];
            for (int mnr = 0; mnr < C; ++mnr) {
                int32_t new_box_id = split_box_ids[ibox] - C + mnr;    /* tbk:667 */
                box_parent_ids[new_box_id] = (int32_t) ibox;
                box_child_ids[mnr][ibox] = new_box_id;
                int new_level = box_levels[ibox] + 1;
                box_levels[new_box_id] = (uint8_t) new_level;
                int32_t new_count = bmc.pcnt[mnr];
                box_srcntgt_counts_cumul[new_box_id] = new_count;
                if (new_count > 0) {                                   /* tbk:682-695 */
                    int32_t new_box_start = box_srcntgt_starts[ibox];
                    if (have_extent) new_box_start += bmc.nonchild_srcntgts;
                    for (int sub = 0; sub < mnr; ++sub) new_box_start += bmc.pcnt[sub];
                    box_start_flags[new_box_start] = 1;
                    box_srcntgt_starts[new_box_id] = new_box_start;
                }
                /* tbk:698-705 */
                COORD_T radius = (in->root_extent * 1 / (COORD_T) (1 << (1 + new_level)));
                for (int idim = 0; idim < dims; ++idim) {
                    int has_bit = mnr & (1 << (dims - 1 - idim));
                    box_centers[idim][new_box_id] = has_bit
                        ? box_centers[idim][ibox] + radius
                        : box_centers[idim][ibox] - radius;
                }
            }
        }

        /* K6 renumber_particles (:1111, tbk:744-819) */
        ORC_PRAGMA(omp parallel for schedule(static))
        for (int64_t i = 0; i < N; ++i) {
            int32_t ibox = srcntgt_box_ids[i];
            int do_split_box = box_has_children[ibox] && (box_levels[ibox] + 1 == level);
            if (level_restrict) do_split_box = do_split_box || force_split_box[ibox];  /* tbk:748-750 */
            if (!do_split_box) {
                new_user_srcntgt_ids[i] = user_srcntgt_ids[i];
                new_srcntgt_box_ids[i] = ibox;
                continue;
            }
            int my_morton_nr = morton_nrs[i];
            const orc_mc_t *mybox = &box_morton_bin_counts[ibox];
            int32_t my_count = SFX(orc_get_count)(&morton_bin_counts[i], my_morton_nr);
            int32_t tgt = box_srcntgt_starts[ibox] + my_count - 1;     /* tbk:776-777 */
            if (have_extent)
                tgt += (my_morton_nr >= 0) ? mybox->nonchild_srcntgts : 0;
            for (int mnr = 0; mnr < C; ++mnr)
                tgt += (my_morton_nr > mnr) ? mybox->pcnt[mnr] : 0;    /* tbk:784-790 */
            new_user_srcntgt_ids[tgt] = user_srcntgt_ids[i];
            int32_t new_box_id = split_box_ids[ibox] - C + my_morton_nr;
            if (have_extent && my_morton_nr == -1) new_box_id = ibox;  /* tbk:804-811 */
            new_srcntgt_box_ids[tgt] = new_box_id;
        }
        { int32_t *t = user_srcntgt_ids; user_srcntgt_ids = new_user_srcntgt_ids;
          new_user_srcntgt_ids = t; }
        { int32_t *t = srcntgt_box_ids; srcntgt_box_ids = new_srcntgt_box_ids;
          new_srcntgt_box_ids = t; }

        /* enforce level restriction on upper levels: :1125-1224 */
        if (final_level_restrict_iteration) {                          /* :1127-1143 */
            if (have_oversize_split_box || level_used_box_counts[nlev_used - 1] != 0) {
                status = ORC_ERR_INTERNAL; goto done;
            }
            nlev_used -= 1;
            nlev_starts -= 1;
            level -= 1;
            break;
        }
        if (level_restrict) {
            for (int64_t i = 0; i < nboxes_new; ++i) force_split_box[i] = 0;   /* :1155 */
            int did_upper_level_split = 0;
            /* upper_level = level-2 .. 1 (:1163-1172): the parent level already has
             * a 2-to-1 ratio with the level just built */
            for (int upper_level = level - 2; upper_level >= 1; --upper_level) {
                const int32_t upper_level_start = level_start_box_nrs[upper_level];
                const int32_t upper_level_box_count = level_used_box_counts[upper_level];
                int have_upper_level_split_box = 0;
                /* LEVEL_RESTRICT_TPL: tbk:825-913 */
                for (int32_t box_id = upper_level_start;
                        box_id < upper_level_start + upper_level_box_count; ++box_id) {
                    if (box_has_children[box_id]) continue;
                    int32_t walk_box_stack[128]; int walk_morton_nr_stack[128];
                    int walk_stack_size = 0; int32_t walk_parent_box_id = 0;
                    int walk_morton_nr = 0; int continue_walk = 1;
                    while (continue_walk) {
                        int32_t child_box_id = box_child_ids[walk_morton_nr][walk_parent_box_id];
                        if (child_box_id) {
                            int child_level = walk_stack_size + 1;
                            int is_adjacent;
                            if (child_box_id == box_id) {
                                is_adjacent = 0;
                            } else {
                                COORD_T bc[ORC_MAXDIM], cc[ORC_MAXDIM];
                                for (int d = 0; d < dims; ++d) {
                                    bc[d] = box_centers[d][box_id];
                                    cc[d] = box_centers[d][child_box_id];
                                }
                                is_adjacent = SFX(orc_lr_is_adj)(dims, in->root_extent,
                                        cc, child_level, bc, upper_level);
                            }
                            if (is_adjacent) {
                                if (box_has_children[child_box_id]) {
                                    if (child_level <= 1 + upper_level) {
                                        walk_box_stack[walk_stack_size] = walk_parent_box_id;
                                        walk_morton_nr_stack[walk_stack_size] = walk_morton_nr;
                                        ++walk_stack_size;
                                        walk_parent_box_id = child_box_id; walk_morton_nr = 0;
                                        continue;
                                    }
                                } else {
                                    if (child_level == 2 + upper_level || (
                                            child_level == 1 + upper_level
                                            && force_split_box[child_box_id])) {
                                        force_split_box[box_id] = 1;
                                        have_upper_level_split_box = 1;
                                        continue_walk = 0;
                                    }
                                }
                            }
                        }
                        while (1) {                                    /* walk_advance */
                            ++walk_morton_nr;
                            if (walk_morton_nr < C) break;
                            continue_walk = (walk_stack_size > 0);
                            if (continue_walk) {
                                --walk_stack_size;
                                walk_parent_box_id = walk_box_stack[walk_stack_size];
                                walk_morton_nr = walk_morton_nr_stack[walk_stack_size];
                            } else break;
                        }
                    }
                }
                if (!have_upper_level_split_box) break;                 /* :1201-1202 */
                did_upper_level_split = 1;
            }
            if (!have_oversize_split_box && did_upper_level_split) {    /* :1216-1224 */
                final_level_restrict_iteration = 1;
                level += 1;
                continue;
            }
        }

        if (!have_oversize_split_box) break;                           /* :1228-1230 */
        level += 1;
        have_oversize_split_box = 0;
    }

    int64_t nboxes = level_start_box_nrs[nlev_starts - 1];             /* :1278 */

    /* K8 nonchild extraction, tree_build.py:1288-1305, tbk:979-1007 */
    if (have_extent) {
        box_srcntgt_counts_nonchild = (int32_t *) calloc((size_t) nboxes, 4);
        if (!box_srcntgt_counts_nonchild) { status = ORC_ERR_ALLOC; goto done; }
        int32_t highest_possibly_split_box_nr = level_start_box_nrs[nlev_starts - 2];
        for (int64_t i = 0; i < nboxes; ++i) {
            if (i >= highest_possibly_split_box_nr) box_srcntgt_counts_nonchild[i] = 0;
            else if (box_srcntgt_counts_cumul[i] == 0) box_srcntgt_counts_nonchild[i] = 0;
            else box_srcntgt_counts_nonchild[i] = box_morton_bin_counts[i].nonchild_srcntgts;
        }
    }

    /* prune: tree_build.py:1328-1456 */
    int64_t nboxes_post_prune = nboxes;
    if (!in->skip_prune) {
        src_box_id = (int32_t *) calloc((size_t) nboxes, 4);
        dst_box_id = (int32_t *) calloc((size_t) nboxes, 4);
        if (!src_box_id || !dst_box_id) { status = ORC_ERR_ALLOC; goto done; }
        int32_t item = 0;                                  /* K9 tbk:1697-1718 */
        for (int64_t i = 0; i < nboxes; ++i) {
            item += (box_srcntgt_counts_cumul[i] != 0);
            if (box_srcntgt_counts_cumul[i]) {
                dst_box_id[i] = item - 1;
                src_box_id[item - 1] = (int32_t) i;
            }
        }
        nboxes_post_prune = item;

        /* K10 gappy copies, tools.py:417-438 */
#define ORC_PRUNE(arr, type, mapvals) do { \
            type *na_ = (type *) calloc((size_t) nboxes_post_prune + 1, sizeof(type)); \
            if (!na_) { status = ORC_ERR_ALLOC; goto done; } \
            for (int64_t i_ = 0; i_ < nboxes_post_prune; ++i_) { \
                type v_ = arr[src_box_id[i_]]; \
                if (mapvals) v_ = (type) dst_box_id[(int64_t) v_]; \
                na_[i_] = v_; } \
            free(arr); arr = na_; } while (0)
        ORC_PRUNE(box_srcntgt_starts, int32_t, 0);
        ORC_PRUNE(box_srcntgt_counts_cumul, int32_t, 0);
        ORC_PRAGMA(omp parallel for schedule(static))
        for (int64_t i = 0; i < N; ++i)                     /* :1406 map_values */
            srcntgt_box_ids[i] = dst_box_id[srcntgt_box_ids[i]];
        ORC_PRUNE(box_parent_ids, int32_t, 1);
        ORC_PRUNE(box_levels, uint8_t, 0);
        if (have_extent) ORC_PRUNE(box_srcntgt_counts_nonchild, int32_t, 0);
        ORC_PRUNE(box_has_children, int32_t, 0);
        for (int m = 0; m < C; ++m) ORC_PRUNE(box_child_ids[m], int32_t, 1);
        for (int d = 0; d < dims; ++d) ORC_PRUNE(box_centers[d], COORD_T, 0);
        nboxes_alloc = nboxes_post_prune + 1;

        /* K11 find_level_box_counts, tbk:1724-1742; :1440-1445 */
        {
            int32_t acc = 0;
            for (int64_t i = 0; i < nboxes_post_prune; ++i) {
                int seg_start = (i == 0) || (box_levels[i] != box_levels[i - 1]);
                acc = seg_start ? 1 : acc + 1;
                if (i + 1 == nboxes_post_prune || box_levels[i] != box_levels[i + 1])
                    level_used_box_counts[box_levels[i]] = acc;
            }
            int nlevels_ = nlev_used;
            level_start_box_nrs[0] = 0;
            for (int l = 0; l < nlevels_; ++l)
                level_start_box_nrs[l + 1] = level_start_box_nrs[l] + level_used_box_counts[l];
            nlev_starts = nlevels_ + 1;
        }
    }
    const int64_t B = nboxes_post_prune;
    const int nlevels = nlev_starts - 1;                               /* :1628 */
    if (level + 1 != nlevels) { status = ORC_ERR_INTERNAL; goto done; } /* :1631 */
    const int64_t aligned = ((B + 31) / 32) * 32;                      /* :1641 */

    out->nlevels = nlevels;
    out->nboxes = B;
    out->aligned_nboxes = aligned;
    out->level_start_box_nrs = (int32_t *) calloc((size_t) nlevels + 1, 4);
    for (int l = 0; l <= nlevels; ++l) out->level_start_box_nrs[l] = level_start_box_nrs[l];

    /* sources/targets: tree_build.py:1462-1567 */
    const int64_t nsources = in->nsources;
    const int64_t ntargets = in->sources_are_targets ? N : N - nsources;
    int32_t *box_source_starts = NULL, *box_source_counts_cumul = NULL;
    int32_t *box_source_counts_nonchild = NULL;
    int32_t *box_target_starts = NULL, *box_target_counts_cumul = NULL;
    int32_t *box_target_counts_nonchild = NULL;
    int own_st_arrays = 0;   /* 1 if box_{source,target}_* are separate allocations */

    if (in->sources_are_targets) {
        out->user_source_ids = (int32_t *) malloc((size_t) (N ? N : 1) * 4);
        out->sorted_target_ids = (int32_t *) calloc((size_t) (N ? N : 1), 4);
        memcpy(out->user_source_ids, user_srcntgt_ids, (size_t) N * 4);
        ORC_PRAGMA(omp parallel for schedule(static))
        for (int64_t i = 0; i < N; ++i)                  /* K17 tools.py:81-109 */
            out->sorted_target_ids[user_srcntgt_ids[i]] = (int32_t) i;
        /* :1469-1474 */
        box_source_starts = box_target_starts = box_srcntgt_starts;
        box_source_counts_cumul = box_target_counts_cumul = box_srcntgt_counts_cumul;
        if (have_extent)
            box_source_counts_nonchild = box_target_counts_nonchild
                = box_srcntgt_counts_nonchild;
    } else {
        own_st_arrays = 1;
        source_numbers = (int32_t *) calloc((size_t) (N ? N : 1), 4);
        srcntgt_target_ids = (int32_t *) calloc((size_t) (ntargets ? ntargets : 1), 4);
        out->user_source_ids = (int32_t *) calloc((size_t) (nsources ? nsources : 1), 4);
        out->sorted_target_ids = (int32_t *) calloc((size_t) (ntargets ? ntargets : 1), 4);
        box_source_starts = (int32_t *) calloc((size_t) B + 1, 4);
        box_source_counts_cumul = (int32_t *) calloc((size_t) B + 1, 4);
        box_target_starts = (int32_t *) calloc((size_t) B + 1, 4);
        box_target_counts_cumul = (int32_t *) calloc((size_t) B + 1, 4);
        if (have_extent) {
            box_source_counts_nonchild = (int32_t *) calloc((size_t) B + 1, 4);
            box_target_counts_nonchild = (int32_t *) calloc((size_t) B + 1, 4);
        }
        /* K12 source_counter, tbk:1770-1782 (exclusive scan) */
        {
            int32_t acc = 0;
            for (int64_t i = 0; i < N; ++i) {
                source_numbers[i] = acc;
                acc += (user_srcntgt_ids[i] < nsources) ? 1 : 0;
            }
        }
        /* K13 find_source_and_target_indices, tbk:1013-1164 */
        ORC_PRAGMA(omp parallel for schedule(static))
        for (int64_t i = 0; i < N; ++i) {
            int32_t sorted_srcntgt_id = (int32_t) i;
            int32_t source_nr = source_numbers[i];
            int32_t target_nr = (int32_t) i - source_nr;
            int32_t box_id = srcntgt_box_ids[i];
            int32_t box_start = box_srcntgt_starts[box_id];
            int32_t box_count = box_srcntgt_counts_cumul[box_id];
            int32_t user_srcntgt_id = user_srcntgt_ids[i];
            int is_source = user_srcntgt_id < nsources;
            {
                int32_t walk_box_start = box_start, walk_box_id = box_id;
                while (sorted_srcntgt_id == walk_box_start) {
                    box_source_starts[walk_box_id] = source_nr;
                    box_target_starts[walk_box_id] = target_nr;
                    int32_t new_box_id = box_parent_ids[walk_box_id];
                    if (new_box_id == walk_box_id) break;
                    walk_box_id = new_box_id;
                    walk_box_start = box_srcntgt_starts[walk_box_id];
                }
            }
            if (have_extent) {
                int32_t box_nonchild_count = box_srcntgt_counts_nonchild[box_id];
                if (sorted_srcntgt_id + 1 == box_start + box_nonchild_count) {
                    int32_t bs_src = source_numbers[box_start];
                    int32_t bs_tgt = box_start - bs_src;
                    box_source_counts_nonchild[box_id] = source_nr + is_source - bs_src;
                    box_target_counts_nonchild[box_id] = target_nr + 1 - is_source - bs_tgt;
                }
            }
            {
                int32_t walk_box_start = box_start, walk_box_count = box_count;
                int32_t walk_box_id = box_id;
                while (sorted_srcntgt_id + 1 == walk_box_start + walk_box_count) {
                    int32_t bs_src = source_numbers[walk_box_start];
                    int32_t bs_tgt = walk_box_start - bs_src;
                    box_source_counts_cumul[walk_box_id] = source_nr + is_source - bs_src;
                    box_target_counts_cumul[walk_box_id] = target_nr + 1 - is_source - bs_tgt;
                    int32_t new_box_id = box_parent_ids[walk_box_id];
                    if (new_box_id == walk_box_id) break;
                    walk_box_id = new_box_id;
                    walk_box_start = box_srcntgt_starts[walk_box_id];
                    walk_box_count = box_srcntgt_counts_cumul[walk_box_id];
                }
            }
            if (is_source) {
                out->user_source_ids[source_nr] = user_srcntgt_id;
            } else {
                srcntgt_target_ids[target_nr] = user_srcntgt_id;
                out->sorted_target_ids[user_srcntgt_id - nsources] = target_nr;
            }
        }
    }

    /* K14 permute, tree_build.py:1569-1622, tbk:1170-1186 */
    for (int d = 0; d < dims; ++d) {
        if (in->sources_are_targets) {
            out->sources[d] = (COORD_T *) malloc((size_t) (N ? N : 1) * sizeof(COORD_T));
            ORC_PRAGMA(omp parallel for schedule(static))
            for (int64_t i = 0; i < N; ++i)
                out->sources[d][i] = in->srcntgts[d][user_srcntgt_ids[i]];
            out->targets[d] = out->sources[d];
        } else {
            out->sources[d] = (COORD_T *) malloc((size_t) (nsources ? nsources : 1) * sizeof(COORD_T));
            out->targets[d] = (COORD_T *) malloc((size_t) (ntargets ? ntargets : 1) * sizeof(COORD_T));
            ORC_PRAGMA(omp parallel for schedule(static))
            for (int64_t i = 0; i < nsources; ++i)
                out->sources[d][i] = in->srcntgts[d][out->user_source_ids[i]];
            ORC_PRAGMA(omp parallel for schedule(static))
            for (int64_t i = 0; i < ntargets; ++i)
                out->targets[d][i] = in->srcntgts[d][srcntgt_target_ids[i]];
        }
    }
    if (in->srcntgt_radii && !in->sources_are_targets) {               /* :1606-1616 */
        out->source_radii = (COORD_T *) malloc((size_t) (nsources ? nsources : 1) * sizeof(COORD_T));
        out->target_radii = (COORD_T *) malloc((size_t) (ntargets ? ntargets : 1) * sizeof(COORD_T));
        for (int64_t i = 0; i < nsources; ++i)
            out->source_radii[i] = in->srcntgt_radii[out->user_source_ids[i]];
        for (int64_t i = 0; i < ntargets; ++i)
            out->target_radii[i] = in->srcntgt_radii[srcntgt_target_ids[i]];
    }

    /* repack: tree_build.py:1636-1664 */
    out->box_child_ids = (int32_t *) calloc((size_t) (C * aligned) + 1, 4);
    out->box_centers = (COORD_T *) calloc((size_t) (dims * aligned) + 1, sizeof(COORD_T));
    for (int m = 0; m < C; ++m)
        for (int64_t i = 0; i < B; ++i) out->box_child_ids[m * aligned + i] = box_child_ids[m][i];
    for (int d = 0; d < dims; ++d)
        for (int64_t i = 0; i < B; ++i) out->box_centers[d * aligned + i] = box_centers[d][i];

    /* K15 box_info: tree_build.py:1666-1723, tbk:1192-1305 */
    out->box_flags = (uint8_t *) calloc((size_t) B + 1, 1);
    int32_t *nonchild_alloc_s = NULL, *nonchild_alloc_t = NULL;
    if (!have_extent) {                                                /* :1697-1706 */
        nonchild_alloc_s = box_source_counts_nonchild = (int32_t *) calloc((size_t) B + 1, 4);
        if (in->sources_are_targets) box_target_counts_nonchild = box_source_counts_nonchild;
        else nonchild_alloc_t = box_target_counts_nonchild = (int32_t *) calloc((size_t) B + 1, 4);
    }
    ORC_PRAGMA(omp parallel for schedule(static))
    for (int64_t box_id = 0; box_id < B; ++box_id) {
        int32_t particle_count = box_srcntgt_counts_cumul[box_id];
        int32_t nonchild_source_count = have_extent ? box_source_counts_nonchild[box_id] : 0;
        int32_t nonchild_target_count = have_extent ? box_target_counts_nonchild[box_id] : 0;
        /* NB (sources_are_targets && have_extent cannot happen: tree_build.py:253) */
        int32_t nonchild_srcntgt_count = nonchild_source_count + nonchild_target_count;
        uint8_t my_box_flags = 0;
        if (box_has_children[box_id]) {
            my_box_flags |= BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES;      /* tbk:1256 */
            if (in->sources_are_targets) {
                if (particle_count - nonchild_srcntgt_count)
                    my_box_flags |= BOX_HAS_SOURCE_CHILD_BOXES | BOX_HAS_TARGET_CHILD_BOXES;
            } else {
                int32_t source_count = box_source_counts_cumul[box_id];
                int32_t target_count = box_target_counts_cumul[box_id];
                if (source_count - nonchild_source_count)
                    my_box_flags |= BOX_HAS_SOURCE_CHILD_BOXES;
                if (target_count - nonchild_target_count)
                    my_box_flags |= BOX_HAS_TARGET_CHILD_BOXES;
            }
            if (nonchild_source_count) my_box_flags |= BOX_IS_SOURCE_BOX;
            if (nonchild_target_count) my_box_flags |= BOX_IS_TARGET_BOX;
        } else {
            if (in->sources_are_targets) {
                if (particle_count)
                    my_box_flags |= BOX_IS_SOURCE_BOX | BOX_IS_TARGET_BOX;
                box_source_counts_nonchild[box_id] = particle_count;
            } else {
                int32_t my_source_count = box_source_counts_cumul[box_id];
                int32_t my_target_count = particle_count - my_source_count;
                if (my_source_count) my_box_flags |= BOX_IS_SOURCE_BOX;
                if (my_target_count) my_box_flags |= BOX_IS_TARGET_BOX;
                box_source_counts_nonchild[box_id] = my_source_count;
                box_target_counts_nonchild[box_id] = my_target_count;
            }
        }
        out->box_flags[box_id] = my_box_flags;
    }

    /* K16 box extents: tree_build.py:1730-1806, tbk:1311-1399 */
    {
        size_t sz = (size_t) (dims * aligned) + 1;
        out->box_source_bounding_box_min = (COORD_T *) calloc(sz, sizeof(COORD_T));
        out->box_source_bounding_box_max = (COORD_T *) calloc(sz, sizeof(COORD_T));
        if (in->sources_are_targets) {
            out->box_target_bounding_box_min = out->box_source_bounding_box_min;
            out->box_target_bounding_box_max = out->box_source_bounding_box_max;
        } else {
            out->box_target_bounding_box_min = (COORD_T *) calloc(sz, sizeof(COORD_T));
            out->box_target_bounding_box_max = (COORD_T *) calloc(sz, sizeof(COORD_T));
        }
        for (int lev = nlevels - 1; lev >= 0; --lev) {
            int64_t start = level_start_box_nrs[lev], stop = level_start_box_nrs[lev + 1];
            for (int round = 0; round < 2; ++round) {
                if (round == 1 && in->sources_are_targets) continue;
                int enable_radii = round == 0 ? in->sources_have_extent : in->targets_have_extent;
                COORD_T *bmin = round == 0 ? out->box_source_bounding_box_min
                                           : out->box_target_bounding_box_min;
                COORD_T *bmax = round == 0 ? out->box_source_bounding_box_max
                                           : out->box_target_bounding_box_max;
                const int32_t *pstarts = round == 0 ? box_source_starts : box_target_starts;
                const int32_t *pcounts = round == 0 ? box_source_counts_nonchild
                                                    : box_target_counts_nonchild;
                COORD_T *const *particles = round == 0 ? out->sources : out->targets;
                const COORD_T *pradii = round == 0 ? out->source_radii : out->target_radii;
                ORC_PRAGMA(omp parallel for schedule(dynamic, 256))
                for (int64_t ibox = start; ibox < stop; ++ibox) {
                    COORD_T mn[ORC_MAXDIM], mx[ORC_MAXDIM];
                    for (int d = 0; d < dims; ++d)
                        mn[d] = mx[d] = out->box_centers[d * aligned + ibox];
                    int32_t pstart = pstarts[ibox], pstop = pstart + pcounts[ibox];
                    for (int32_t ip = pstart; ip < pstop; ++ip) {
                        COORD_T rad = 0;
                        if (have_extent && enable_radii) rad = pradii[ip];
                        for (int d = 0; d < dims; ++d) {
                            COORD_T c = particles[d][ip];
                            COORD_T lo = c - rad, hi = c + rad;
                            mn[d] = (lo < mn[d]) ? lo : mn[d];
                            mx[d] = (hi > mx[d]) ? hi : mx[d];
                        }
                    }
                    for (int m = 0; m < C; ++m) {
                        int32_t child_id = out->box_child_ids[m * aligned + ibox];
                        if (child_id == 0) continue;
                        for (int d = 0; d < dims; ++d) {
                            COORD_T lo = bmin[d * aligned + child_id];
                            COORD_T hi = bmax[d * aligned + child_id];
                            mn[d] = (lo < mn[d]) ? lo : mn[d];
                            mx[d] = (hi > mx[d]) ? hi : mx[d];
                        }
                    }
                    for (int d = 0; d < dims; ++d) {
                        bmin[d * aligned + ibox] = mn[d];
                        bmax[d * aligned + ibox] = mx[d];
                    }
                }
            }
        }
    }

    /* hand over copies of the per-box arrays */
#define ORC_DUP(dst, src, n, type) do { \
        dst = (type *) malloc(((size_t) (n) + 1) * sizeof(type)); \
        memcpy(dst, src, (size_t) (n) * sizeof(type)); } while (0)
    ORC_DUP(out->box_source_starts, box_source_starts, B, int32_t);
    ORC_DUP(out->box_source_counts_cumul, box_source_counts_cumul, B, int32_t);
    ORC_DUP(out->box_source_counts_nonchild, box_source_counts_nonchild, B, int32_t);
    if (in->sources_are_targets) {
        out->box_target_starts = out->box_source_starts;
        out->box_target_counts_cumul = out->box_source_counts_cumul;
        out->box_target_counts_nonchild = out->box_source_counts_nonchild;
    } else {
        ORC_DUP(out->box_target_starts, box_target_starts, B, int32_t);
        ORC_DUP(out->box_target_counts_cumul, box_target_counts_cumul, B, int32_t);
        ORC_DUP(out->box_target_counts_nonchild, box_target_counts_nonchild, B, int32_t);
    }
    ORC_DUP(out->box_parent_ids, box_parent_ids, B, int32_t);
    ORC_DUP(out->box_levels, box_levels, B, uint8_t);
#undef ORC_DUP
    free(nonchild_alloc_s); free(nonchild_alloc_t);
    if (own_st_arrays) {
        free(box_source_starts); free(box_source_counts_cumul);
        free(box_target_starts); free(box_target_counts_cumul);
        if (have_extent) { free(box_source_counts_nonchild); free(box_target_counts_nonchild); }
    }

done:
    out->status = status;
    free(morton_bin_counts); free(morton_nrs); free(box_start_flags);
    free(srcntgt_box_ids); free(user_srcntgt_ids);
    free(new_user_srcntgt_ids); free(new_srcntgt_box_ids);
    free(split_box_ids); free(box_srcntgt_starts); free(box_parent_ids); free(force_split_box);
    free(box_srcntgt_counts_cumul); free(box_has_children);
    for (int m = 0; m < ORC_MAXC; ++m) free(box_child_ids[m]);
    for (int d = 0; d < ORC_MAXDIM; ++d) free(box_centers[d]);
    free(box_levels); free(box_morton_bin_counts); free(box_srcntgt_counts_nonchild);
    free(level_start_box_nrs); free(level_used_box_counts); free(new_level_used_box_counts);
    free(src_box_id); free(dst_box_id); free(source_numbers); free(srcntgt_target_ids);
    return status;
}

#undef ORC_PRUNE
#undef ORC_ENSURE_BOXES
#undef ORC_GROW
