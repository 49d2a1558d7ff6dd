/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See boxtree_oracle_impl.h header.
 *
 * Literal restatement of boxtree/traversal.py (FMMTraversalBuilder).  The list
 * generators run over contiguous chunks of their objects, one chunk per thread when
 * built with -fopenmp; the chunks' lists are concatenated in order.
 * PARITY UNPINNED (see boxtree_oracle_impl.h).
 * Included once per coordinate type (COORD_T / SFX / COORD_SQRT / COORD_EPS).
 */

typedef struct {
    int32_t dims;
    int32_t nlevels;
    int64_t nboxes;
    int64_t aligned_nboxes;
    COORD_T root_extent;
    const COORD_T *box_centers;          /* [d, aligned] */
    const uint8_t *box_levels;
    const int32_t *box_child_ids;        /* [C, aligned] */
    const uint8_t *box_flags;
    const int32_t *box_parent_ids;
    const int32_t *level_start_box_nrs;  /* [nlevels+1] */
    int32_t sources_are_targets;
    int32_t sources_have_extent, targets_have_extent;
    COORD_T stick_out_factor;
    const COORD_T *box_target_bounding_box_min, *box_target_bounding_box_max;
    const int32_t *box_source_counts_cumul;
    int32_t well_sep_is_n_away;
    int32_t from_sep_smaller_crit;       /* ORC_CRIT_* */
    int32_t from_sep_smaller_min_nsources_cumul;
    const int8_t *source_boxes_mask;         /* optional */
    const int8_t *source_parent_boxes_mask;  /* optional */
} SFX(orc_trav_in);

typedef struct {
    int32_t status;
    int64_t nsource_boxes, ntarget_boxes, nsource_parent_boxes,
            ntarget_or_target_parent_boxes;
    int32_t *source_boxes, *target_boxes, *source_parent_boxes,
            *target_or_target_parent_boxes;
    int32_t *level_start_source_box_nrs, *level_start_target_box_nrs,
            *level_start_source_parent_box_nrs,
            *level_start_target_or_target_parent_box_nrs;   /* [nlevels+1] */
    orc_built_list same_level_non_well_sep_boxes;
    orc_built_list neighbor_source_boxes;
    orc_built_list from_sep_siblings;
    orc_built_list *from_sep_smaller_by_level;               /* [nlevels] */
    orc_built_list from_sep_close_smaller;                   /* starts==NULL if none */
    orc_built_list from_sep_bigger;
    orc_built_list from_sep_close_bigger;                    /* starts==NULL if none */
} SFX(orc_trav_out);

/* LEVEL_TO_RAD: traversal.py:234-235 */
#define ORC_LEVEL_TO_RAD(root_extent, level) \
    ((root_extent) * 1 / (COORD_T) (1 << ((level) + 1)))

/* is_adjacent_or_overlapping_with_neighborhood: traversal.py:279-305 */
static inline int SFX(orc_is_adj_nbhd)(int dims, COORD_T root_extent,
        const COORD_T *target_center, int target_level,
        COORD_T target_box_neighborhood_size,
        const COORD_T *source_center, int source_level)
{
    COORD_T target_rad = ORC_LEVEL_TO_RAD(root_extent, target_level);
    COORD_T source_rad = ORC_LEVEL_TO_RAD(root_extent, source_level);
    COORD_T rad_sum = (
        (2 * (target_box_neighborhood_size - 1) + 1) * target_rad
        + source_rad);
    COORD_T slack = rad_sum + ((target_rad < source_rad) ? target_rad : source_rad);
    COORD_T l_inf_dist = 0;
    for (int i = 0; i < dims; ++i) {
        COORD_T d = target_center[i] - source_center[i];
        d = (d < 0) ? -d : d;
        l_inf_dist = (d > l_inf_dist) ? d : l_inf_dist;
    }
    return l_inf_dist <= slack;
}

/* is_adjacent_or_overlapping: traversal.py:307-318 */
static inline int SFX(orc_is_adj)(int dims, COORD_T root_extent,
        const COORD_T *tc, int tl, const COORD_T *sc, int sl)
{
    return SFX(orc_is_adj_nbhd)(dims, root_extent, tc, tl, 1, sc, sl);
}

#define ORC_LOAD_CENTER(name, box_id) \
    COORD_T name[ORC_MAXDIM]; \
    for (int d_ = 0; d_ < dims; ++d_) name[d_] = in->box_centers[aligned * d_ + (box_id)]

/* walk machinery: traversal.py:98-160 */
#define ORC_WALK_DECL \
    int32_t walk_box_stack[128]; int walk_morton_nr_stack[128]; \
    int walk_stack_size; int32_t walk_parent_box_id; int walk_morton_nr; int continue_walk
#define ORC_WALK_INIT(start_box_id) \
    walk_stack_size = 0; walk_parent_box_id = (start_box_id); walk_morton_nr = 0; \
    continue_walk = 1
#define ORC_WALK_GET_BOX_ID \
    int32_t walk_box_id = in->box_child_ids[walk_morton_nr * aligned + walk_parent_box_id]
#define ORC_WALK_ADVANCE \
    while (1) { \
        ++walk_morton_nr; \
        if (walk_morton_nr < C) break; \
        continue_walk = (walk_stack_size > 0); \
        if (continue_walk) { \
            --walk_stack_size; \
            walk_parent_box_id = walk_box_stack[walk_stack_size]; \
            walk_morton_nr = walk_morton_nr_stack[walk_stack_size]; \
        } else break; \
    }
#define ORC_WALK_PUSH(new_box) \
    walk_box_stack[walk_stack_size] = walk_parent_box_id; \
    walk_morton_nr_stack[walk_stack_size] = walk_morton_nr; \
    ++walk_stack_size; \
    walk_parent_box_id = (new_box); walk_morton_nr = 0

/* per-thread list pieces -> one list (in chunk order) */
#ifndef ORC_CONCAT_DEFINED
#define ORC_CONCAT_DEFINED
static void orc_concat_parts(orc_ivec *dst, orc_ivec *parts, int nparts)
{
    int64_t total = 0;
    for (int t = 0; t < nparts; ++t) total += parts[t].n;
    dst->data = (int32_t *) malloc((size_t) (total ? total : 1) * sizeof(int32_t));
    dst->n = 0; dst->cap = total ? total : 1;
    for (int t = 0; t < nparts; ++t) {
        if (parts[t].n) memcpy(dst->data + dst->n, parts[t].data, (size_t) parts[t].n * 4);
        dst->n += parts[t].n;
        free(parts[t].data);
    }
    free(parts);
}
#endif

/* finish a CSR list built object-by-object */
static int SFX(orc_finish_list)(orc_built_list *bl, int64_t n_objects,
        const int32_t *counts, orc_ivec *lists, int eliminate_empty)
{
    bl->n_objects = n_objects;
    bl->count = lists->n;
    bl->lists = lists->data ? lists->data : (int32_t *) calloc(1, 4);
    lists->data = NULL;
    if (!eliminate_empty) {
        bl->num_nonempty_lists = -1;
        bl->starts = (int32_t *) calloc((size_t) n_objects + 1, 4);
        int32_t acc = 0;
        for (int64_t i = 0; i < n_objects; ++i) { bl->starts[i] = acc; acc += counts[i]; }
        bl->starts[n_objects] = acc;
    } else {
        int64_t nne = 0;
        for (int64_t i = 0; i < n_objects; ++i) nne += counts[i] != 0;
        bl->num_nonempty_lists = nne;
        bl->starts = (int32_t *) calloc((size_t) nne + 1, 4);
        bl->nonempty_indices = (int32_t *) calloc((size_t) nne + 1, 4);
        bl->compressed_indices = (int32_t *) calloc((size_t) n_objects + 1, 4);
        int32_t acc = 0; int32_t k = 0;
        for (int64_t i = 0; i < n_objects; ++i) {
            bl->compressed_indices[i] = k;
            if (counts[i]) {
                bl->nonempty_indices[k] = (int32_t) i;
                bl->starts[k] = acc;
                ++k;
            }
            acc += counts[i];
        }
        bl->compressed_indices[n_objects] = k;
        bl->starts[nne] = acc;
    }
    return 0;
}

/* extract_level_start_box_nrs: traversal.py:361-392 + host post-pass :2073-2098 */
static int32_t *SFX(orc_level_starts)(const SFX(orc_trav_in) *in,
        const int32_t *box_list, int64_t n)
{
    int nlevels = in->nlevels;
    int32_t *result = (int32_t *) malloc(((size_t) nlevels + 1) * 4);
    for (int l = 0; l <= nlevels; ++l) result[l] = (int32_t) n;
    for (int64_t i = 0; i < n; ++i) {
        int32_t my_box_id = box_list[i];
        int my_level = in->box_levels[my_box_id];
        int is_level_leading_box;
        if (i == 0) is_level_leading_box = 1;
        else {
            int32_t prev_box_id = box_list[i - 1];
            int32_t my_level_start = in->level_start_box_nrs[my_level];
            is_level_leading_box = (prev_box_id < my_level_start
                    && my_level_start <= my_box_id);
        }
        if (is_level_leading_box) result[my_level] = (int32_t) i;
    }
    int32_t prev_start = (int32_t) n;
    for (int ilev = nlevels - 1; ilev >= 0; --ilev) {
        int32_t v = result[ilev] < prev_start ? result[ilev] : prev_start;
        result[ilev] = prev_start = v;
    }
    return result;
}

/* from_sep_smaller generate(): traversal.py:607-875.
 * Appends to `main` (from_sep_smaller) and `close` (from_sep_close_smaller). */
static void SFX(orc_gen_from_sep_smaller)(const SFX(orc_trav_in) *in,
        const int32_t *target_boxes, const orc_built_list *slnws,
        int32_t target_box_number, int from_sep_smaller_source_level,
        orc_ivec *mainl, orc_ivec *closel)
{
    const int dims = in->dims; const int C = 1 << dims;
    const int64_t aligned = in->aligned_nboxes;
    const COORD_T root_extent = in->root_extent;
    const int close_lists_exist = in->sources_have_extent || in->targets_have_extent;
    ORC_WALK_DECL;

    int32_t tgt_box_id = target_boxes[target_box_number];
    ORC_LOAD_CENTER(tgt_center, tgt_box_id);
    int tgt_level = in->box_levels[tgt_box_id];

    COORD_T tgt_stickout_l_inf_rad = 0;
    COORD_T tgt_ext_center[ORC_MAXDIM] = {0}, tgt_radii_vec[ORC_MAXDIM] = {0};
    if (in->targets_have_extent) {
        if (in->from_sep_smaller_crit == ORC_CRIT_STATIC_LINF
                || in->from_sep_smaller_crit == ORC_CRIT_STATIC_L2) {
            tgt_stickout_l_inf_rad =
                (1 + in->stick_out_factor) * ORC_LEVEL_TO_RAD(root_extent, tgt_level);
        } else {
            /* load_true_box_extent: traversal.py:177-198 */
            for (int d = 0; d < dims; ++d) {
                COORD_T mn = in->box_target_bounding_box_min[d * aligned + tgt_box_id];
                COORD_T mx = in->box_target_bounding_box_max[d * aligned + tgt_box_id];
                tgt_ext_center[d] = ((COORD_T) 0.5) * (mn + mx);
                tgt_radii_vec[d] = ((COORD_T) 0.5) * (mx - mn);
            }
        }
    }

    int32_t slnws_start = slnws->starts[tgt_box_id];
    int32_t slnws_stop = slnws->starts[tgt_box_id + 1];

    for (int32_t i = slnws_start; i < slnws_stop; ++i) {
        int32_t same_lev_nws_box = slnws->lists[i];
        if (same_lev_nws_box == tgt_box_id) continue;

        ORC_WALK_INIT(same_lev_nws_box);
        while (continue_walk) {
            ORC_WALK_GET_BOX_ID;
            uint8_t child_box_flags = in->box_flags[walk_box_id];

            if (walk_box_id && (child_box_flags
                        & (BOX_IS_SOURCE_BOX | BOX_HAS_SOURCE_CHILD_BOXES))) {
                ORC_LOAD_CENTER(walk_center, walk_box_id);
                int walk_level = in->box_levels[walk_box_id];
                int in_list_1 = SFX(orc_is_adj)(dims, root_extent,
                        tgt_center, tgt_level, walk_center, walk_level);
                if (in_list_1) {
                    if (child_box_flags & BOX_HAS_SOURCE_CHILD_BOXES) {
                        if (walk_level <= from_sep_smaller_source_level
                                || from_sep_smaller_source_level == -1) {
                            ORC_WALK_PUSH(walk_box_id);
                            continue;
                        }
                    }
                } else {
                    int meets_sep_crit;
                    if (!in->targets_have_extent) {
                        meets_sep_crit = 1;
                    } else if (in->from_sep_smaller_crit == ORC_CRIT_STATIC_LINF) {
                        COORD_T source_rad = ORC_LEVEL_TO_RAD(root_extent, walk_level);
                        COORD_T l_inf_dist = 0;
                        for (int d = 0; d < dims; ++d) {
                            COORD_T a = tgt_center[d] - walk_center[d];
                            a = (a < 0) ? -a : a;
                            COORD_T v = a - tgt_stickout_l_inf_rad - source_rad;
                            l_inf_dist = (v > l_inf_dist) ? v : l_inf_dist;
                        }
                        meets_sep_crit = l_inf_dist >= (2 - 8 * COORD_EPS) * source_rad;
                    } else if (in->from_sep_smaller_crit == ORC_CRIT_PRECISE_LINF) {
                        COORD_T source_rad = ORC_LEVEL_TO_RAD(root_extent, walk_level);
                        COORD_T l_inf_dist = 0;
                        for (int d = 0; d < dims; ++d) {
                            COORD_T a = tgt_ext_center[d] - walk_center[d];
                            a = (a < 0) ? -a : a;
                            COORD_T v = a - tgt_radii_vec[d] - source_rad;
                            l_inf_dist = (v > l_inf_dist) ? v : l_inf_dist;
                        }
                        meets_sep_crit = l_inf_dist >= (2 - 8 * COORD_EPS) * source_rad;
                    } else {
                        COORD_T source_l_inf_rad = ORC_LEVEL_TO_RAD(root_extent, walk_level);
                        COORD_T l2sq = 0;
                        for (int d = 0; d < dims; ++d) {
                            COORD_T a = tgt_center[d] - walk_center[d];
                            l2sq = l2sq + a * a;
                        }
                        COORD_T rhs = COORD_SQRT(l2sq)
                            - COORD_SQRT((COORD_T) dims) * tgt_stickout_l_inf_rad
                            - source_l_inf_rad;
                        meets_sep_crit = ((2 - 8 * COORD_EPS) * source_l_inf_rad <= rhs);
                    }

                    int force_close = close_lists_exist
                        && (in->box_source_counts_cumul[walk_box_id]
                                < in->from_sep_smaller_min_nsources_cumul);

                    if (meets_sep_crit && !force_close) {
                        if (from_sep_smaller_source_level == walk_level)
                            orc_ivec_push(mainl, walk_box_id);
                    } else if (close_lists_exist) {
                        if ((child_box_flags & BOX_IS_SOURCE_BOX)
                                && from_sep_smaller_source_level == -1)
                            orc_ivec_push(closel, walk_box_id);
                        if (child_box_flags & BOX_HAS_SOURCE_CHILD_BOXES) {
                            ORC_WALK_PUSH(walk_box_id);
                            continue;
                        }
                    }
                }
            }
            ORC_WALK_ADVANCE
        }
    }
}

/* meets_sep_bigger_criterion: traversal.py:933-972 */
static inline int SFX(orc_meets_sep_bigger)(int dims, COORD_T root_extent,
        const COORD_T *target_center, int target_level,
        const COORD_T *source_center, int source_level, COORD_T stick_out_factor)
{
    COORD_T target_rad = ORC_LEVEL_TO_RAD(root_extent, target_level);
    COORD_T source_rad = ORC_LEVEL_TO_RAD(root_extent, source_level);
    COORD_T max_allowed = (3 * (1 + stick_out_factor) * target_rad + source_rad);
    COORD_T l_inf_dist = 0;
    for (int i = 0; i < dims; ++i) {
        COORD_T d = target_center[i] - source_center[i];
        d = (d < 0) ? -d : d;
        l_inf_dist = (d > l_inf_dist) ? d : l_inf_dist;
    }
    return l_inf_dist >= max_allowed * (1 - 8 * COORD_EPS);
}

int SFX(orc_trav_build)(const SFX(orc_trav_in) *in, SFX(orc_trav_out) *out)
{
    const int dims = in->dims; const int C = 1 << dims;
    const int64_t aligned = in->aligned_nboxes;
    const int64_t B = in->nboxes;
    const COORD_T root_extent = in->root_extent;
    const int nway = in->well_sep_is_n_away;
    const int with_extent = in->sources_have_extent || in->targets_have_extent;
    memset(out, 0, sizeof(*out));

    /* T1 sources_parents_and_targets: traversal.py:326-355, 2043-2067 */
    orc_ivec sb = {0}, spb = {0}, tb = {0}, ttpb = {0};
    for (int64_t box_id = 0; box_id < B; ++box_id) {
        uint8_t flags = in->box_flags[box_id];
        if ((flags & BOX_IS_SOURCE_BOX)
                && (!in->source_boxes_mask || in->source_boxes_mask[box_id]))
            orc_ivec_push(&sb, (int32_t) box_id);
        if ((flags & BOX_HAS_SOURCE_CHILD_BOXES)
                && (!in->source_parent_boxes_mask || in->source_parent_boxes_mask[box_id]))
            orc_ivec_push(&spb, (int32_t) box_id);
        if (!in->sources_are_targets && (flags & BOX_IS_TARGET_BOX))
            orc_ivec_push(&tb, (int32_t) box_id);
        if (flags & (BOX_HAS_TARGET_CHILD_BOXES | BOX_IS_TARGET_BOX))
            orc_ivec_push(&ttpb, (int32_t) box_id);
    }
    if (!sb.data) sb.data = (int32_t *) calloc(1, 4);
    if (!spb.data) spb.data = (int32_t *) calloc(1, 4);
    if (!tb.data) tb.data = (int32_t *) calloc(1, 4);
    if (!ttpb.data) ttpb.data = (int32_t *) calloc(1, 4);
    out->source_boxes = sb.data; out->nsource_boxes = sb.n;
    out->source_parent_boxes = spb.data; out->nsource_parent_boxes = spb.n;
    out->target_or_target_parent_boxes = ttpb.data;
    out->ntarget_or_target_parent_boxes = ttpb.n;
    if (in->sources_are_targets) {
        free(tb.data);
        out->target_boxes = out->source_boxes; out->ntarget_boxes = sb.n;
    } else {
        out->target_boxes = tb.data; out->ntarget_boxes = tb.n;
    }
    const int32_t *target_boxes = out->target_boxes;
    const int64_t ntarget_boxes = out->ntarget_boxes;
    const int32_t *ttp_boxes = out->target_or_target_parent_boxes;
    const int64_t nttp = out->ntarget_or_target_parent_boxes;

    /* T2 level starts */
    out->level_start_source_box_nrs = SFX(orc_level_starts)(in, out->source_boxes, out->nsource_boxes);
    out->level_start_source_parent_box_nrs = SFX(orc_level_starts)(in, out->source_parent_boxes, out->nsource_parent_boxes);
    out->level_start_target_box_nrs = SFX(orc_level_starts)(in, target_boxes, ntarget_boxes);
    out->level_start_target_or_target_parent_box_nrs = SFX(orc_level_starts)(in, ttp_boxes, nttp);

    /* T3 same_level_non_well_sep_boxes: traversal.py:398-464 */
    {
        orc_ivec l = {0};
        int32_t *counts = (int32_t *) calloc((size_t) B + 1, 4);
        const int nth = ORC_NTHREADS();
        orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
        ORC_PRAGMA(omp parallel num_threads(nth))
        {
            ORC_CHECK_TEAM(nth);
        ORC_WALK_DECL;
        orc_ivec *pl = &parts[ORC_TID()];
        ORC_CHUNK(B, ORC_TID(), nth, lo_, hi_);
        for (int64_t box_id = lo_; box_id < hi_; ++box_id) {
            int64_t n0 = pl->n;
            ORC_LOAD_CENTER(center, box_id);
            if (box_id != 0) {
                int level = in->box_levels[box_id];
                ORC_WALK_INIT(0);
                while (continue_walk) {
                    ORC_WALK_GET_BOX_ID;
                    if (walk_box_id) {
                        ORC_LOAD_CENTER(walk_center, walk_box_id);
                        int a_or_o = SFX(orc_is_adj_nbhd)(dims, root_extent,
                                center, level, (COORD_T) nway,
                                walk_center, in->box_levels[walk_box_id]);
                        if (a_or_o) {
                            if (walk_stack_size + 1 == level && walk_box_id != box_id) {
                                orc_ivec_push(pl, walk_box_id);
                            } else {
                                ORC_WALK_PUSH(walk_box_id);
                                continue;
                            }
                        }
                    }
                    ORC_WALK_ADVANCE
                }
            }
            counts[box_id] = (int32_t) (pl->n - n0);
        }
        }
        orc_concat_parts(&l, parts, nth);
        SFX(orc_finish_list)(&out->same_level_non_well_sep_boxes, B, counts, &l, 0);
        free(counts);
    }
    const orc_built_list *slnws = &out->same_level_non_well_sep_boxes;

    /* T4 neighbor_source_boxes (list 1): traversal.py:470-550 */
    {
        orc_ivec l = {0};
        int32_t *counts = (int32_t *) calloc((size_t) ntarget_boxes + 1, 4);
        const int nth = ORC_NTHREADS();
        orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
        ORC_PRAGMA(omp parallel num_threads(nth))
        {
            ORC_CHECK_TEAM(nth);
        ORC_WALK_DECL;
        orc_ivec *pl = &parts[ORC_TID()];
        ORC_CHUNK(ntarget_boxes, ORC_TID(), nth, lo_, hi_);
        for (int64_t tbn = lo_; tbn < hi_; ++tbn) {
            int64_t n0 = pl->n;
            int32_t box_id = target_boxes[tbn];
            ORC_LOAD_CENTER(center, box_id);
            int level = in->box_levels[box_id];
            if (in->box_flags[0] & BOX_IS_SOURCE_BOX) orc_ivec_push(pl, 0);
            ORC_WALK_INIT(0);
            while (continue_walk) {
                ORC_WALK_GET_BOX_ID;
                if (walk_box_id) {
                    ORC_LOAD_CENTER(walk_center, walk_box_id);
                    int a_or_o = SFX(orc_is_adj)(dims, root_extent, center, level,
                            walk_center, in->box_levels[walk_box_id]);
                    if (a_or_o) {
                        uint8_t flags = in->box_flags[walk_box_id];
                        if (flags & BOX_IS_SOURCE_BOX) orc_ivec_push(pl, walk_box_id);
                        if (flags & BOX_HAS_SOURCE_CHILD_BOXES) {
                            ORC_WALK_PUSH(walk_box_id);
                            continue;
                        }
                    }
                }
                ORC_WALK_ADVANCE
            }
            counts[tbn] = (int32_t) (pl->n - n0);
        }
        }
        orc_concat_parts(&l, parts, nth);
        SFX(orc_finish_list)(&out->neighbor_source_boxes, ntarget_boxes, counts, &l, 0);
        free(counts);
    }

    /* T5 from_sep_siblings (list 2): traversal.py:556-601 */
    {
        orc_ivec l = {0};
        int32_t *counts = (int32_t *) calloc((size_t) nttp + 1, 4);
        const int nth = ORC_NTHREADS();
        orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
        ORC_PRAGMA(omp parallel num_threads(nth))
        {
            ORC_CHECK_TEAM(nth);
        orc_ivec *pl = &parts[ORC_TID()];
        ORC_CHUNK(nttp, ORC_TID(), nth, lo_, hi_);
        for (int64_t it = lo_; it < hi_; ++it) {
            int64_t n0 = pl->n;
            int32_t box_id = ttp_boxes[it];
            ORC_LOAD_CENTER(center, box_id);
            int level = in->box_levels[box_id];
            int32_t parent = in->box_parent_ids[box_id];
            if (parent != box_id) {
                int32_t ps = slnws->starts[parent], pe = slnws->starts[parent + 1];
                for (int32_t i = ps; i < pe; ++i) {
                    int32_t parent_nf = slnws->lists[i];
                    for (int morton_nr = 0; morton_nr < C; ++morton_nr) {
                        int32_t sib_box_id = in->box_child_ids[morton_nr * aligned + parent_nf];
                        if (sib_box_id == 0) continue;
                        ORC_LOAD_CENTER(sib_center, sib_box_id);
                        int sep = !SFX(orc_is_adj_nbhd)(dims, root_extent, center, level,
                                (COORD_T) nway, sib_center, in->box_levels[sib_box_id]);
                        if (sep) orc_ivec_push(pl, sib_box_id);
                    }
                }
            }
            counts[it] = (int32_t) (pl->n - n0);
        }
        }
        orc_concat_parts(&l, parts, nth);
        SFX(orc_finish_list)(&out->from_sep_siblings, nttp, counts, &l, 0);
        free(counts);
    }

    /* T6 from_sep_smaller (list 3) per source level: traversal.py:2179-2233 */
    {
        out->from_sep_smaller_by_level = (orc_built_list *) calloc(
                (size_t) in->nlevels + 1, sizeof(orc_built_list));
        int32_t *counts = (int32_t *) calloc((size_t) ntarget_boxes + 1, 4);
        for (int ilevel = 0; ilevel < in->nlevels; ++ilevel) {
            orc_ivec l = {0};
            const int nth = ORC_NTHREADS();
            orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
            ORC_PRAGMA(omp parallel num_threads(nth))
            {
                ORC_CHECK_TEAM(nth);
                orc_ivec *pl = &parts[ORC_TID()];
                orc_ivec dummy = {0};
                ORC_CHUNK(ntarget_boxes, ORC_TID(), nth, lo_, hi_);
                for (int64_t tbn = lo_; tbn < hi_; ++tbn) {
                    int64_t n0 = pl->n;
                    SFX(orc_gen_from_sep_smaller)(in, target_boxes, slnws, (int32_t) tbn,
                            ilevel, pl, &dummy);
                    counts[tbn] = (int32_t) (pl->n - n0);
                }
                free(dummy.data);
            }
            orc_concat_parts(&l, parts, nth);
            SFX(orc_finish_list)(&out->from_sep_smaller_by_level[ilevel],
                    ntarget_boxes, counts, &l, 1);
        }
        if (with_extent) {
            orc_ivec l = {0};
            const int nth = ORC_NTHREADS();
            orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
            ORC_PRAGMA(omp parallel num_threads(nth))
            {
                ORC_CHECK_TEAM(nth);
                orc_ivec *pl = &parts[ORC_TID()];
                orc_ivec dummy = {0};
                ORC_CHUNK(ntarget_boxes, ORC_TID(), nth, lo_, hi_);
                for (int64_t tbn = lo_; tbn < hi_; ++tbn) {
                    int64_t n0 = pl->n;
                    SFX(orc_gen_from_sep_smaller)(in, target_boxes, slnws, (int32_t) tbn,
                            -1, &dummy, pl);
                    counts[tbn] = (int32_t) (pl->n - n0);
                }
                free(dummy.data);
            }
            orc_concat_parts(&l, parts, nth);
            SFX(orc_finish_list)(&out->from_sep_close_smaller, ntarget_boxes, counts, &l, 0);
        }
        free(counts);
    }

    /* T7 from_sep_bigger (list 4): traversal.py:975-1145 */
    {
        orc_ivec l = {0}, lc = {0};
        int32_t *counts = (int32_t *) calloc((size_t) nttp + 1, 4);
        int32_t *ccounts = (int32_t *) calloc((size_t) nttp + 1, 4);
        const int nth = ORC_NTHREADS();
        orc_ivec *parts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
        orc_ivec *cparts = (orc_ivec *) calloc((size_t) nth, sizeof(orc_ivec));
        ORC_PRAGMA(omp parallel num_threads(nth))
        {
            ORC_CHECK_TEAM(nth);
        orc_ivec *pl = &parts[ORC_TID()], *plc = &cparts[ORC_TID()];
        ORC_CHUNK(nttp, ORC_TID(), nth, lo_, hi_);
        for (int64_t it = lo_; it < hi_; ++it) {
            int64_t n0 = pl->n, nc0 = plc->n;
            int32_t tgt_ibox = ttp_boxes[it];
            ORC_LOAD_CENTER(tgt_box_center, tgt_ibox);
            int tgt_box_level = in->box_levels[tgt_ibox];
            if (tgt_box_level != 0) {
                int32_t tgt_parent_box_id = in->box_parent_ids[tgt_ibox];
                const int tgt_parent_level = tgt_box_level - 1;
                ORC_LOAD_CENTER(parent_center, tgt_parent_box_id);
                uint8_t tgt_box_flags = in->box_flags[tgt_ibox];
                int walk_level; int32_t cur;
                if (nway == 1) { walk_level = tgt_box_level - 1; cur = tgt_parent_box_id; }
                else { walk_level = tgt_box_level; cur = tgt_ibox; }
                for (; walk_level != 0; --walk_level, cur = in->box_parent_ids[cur]) {
                    int32_t s0 = slnws->starts[cur], s1 = slnws->starts[cur + 1];
                    for (int32_t i = s0; i < s1; ++i) {
                        int32_t slnws_box_id = slnws->lists[i];
                        if (!(in->box_flags[slnws_box_id] & BOX_IS_SOURCE_BOX)) continue;
                        ORC_LOAD_CENTER(slnws_center, slnws_box_id);
                        int in_list_1 = SFX(orc_is_adj)(dims, root_extent,
                                tgt_box_center, tgt_box_level, slnws_center, walk_level);
                        if (in_list_1) continue;
                        if (with_extent) {
                            int tgt_meets = SFX(orc_meets_sep_bigger)(dims, root_extent,
                                    tgt_box_center, tgt_box_level, slnws_center, walk_level,
                                    in->stick_out_factor);
                            if (!tgt_meets) {
                                if (tgt_box_flags & BOX_IS_TARGET_BOX)
                                    orc_ivec_push(plc, slnws_box_id);
                                continue;
                            }
                        }
                        int in_parent_list_1 = SFX(orc_is_adj)(dims, root_extent,
                                parent_center, tgt_parent_level, slnws_center, walk_level);
                        int would_be_in_parent_list_4 = !in_parent_list_1;
                        if (nway > 1)
                            would_be_in_parent_list_4 = would_be_in_parent_list_4
                                && (walk_level < tgt_box_level);
                        if (would_be_in_parent_list_4) {
                            if (with_extent) {
                                int parent_meets = SFX(orc_meets_sep_bigger)(dims, root_extent,
                                        parent_center, tgt_parent_level, slnws_center,
                                        walk_level, in->stick_out_factor);
                                if (!parent_meets) orc_ivec_push(pl, slnws_box_id);
                            }
                        } else {
                            orc_ivec_push(pl, slnws_box_id);
                        }
                    }
                }
            }
            counts[it] = (int32_t) (pl->n - n0);
            ccounts[it] = (int32_t) (plc->n - nc0);
        }
        }
        orc_concat_parts(&l, parts, nth);
        orc_concat_parts(&lc, cparts, nth);
        SFX(orc_finish_list)(&out->from_sep_bigger, nttp, counts, &l, 0);
        if (with_extent) {
            /* re-index close-bigger from target-or-target-parent to target
             * boxes: _ListMerger, traversal.py:1259-1344, 2255-2287 */
            orc_built_list raw; memset(&raw, 0, sizeof(raw));
            SFX(orc_finish_list)(&raw, nttp, ccounts, &lc, 0);
            int32_t *ttp_from_all = (int32_t *) calloc((size_t) B + 1, 4);
            for (int64_t i = 0; i < nttp; ++i) ttp_from_all[ttp_boxes[i]] = (int32_t) i;
            orc_ivec nl = {0};
            int32_t *ncounts = (int32_t *) calloc((size_t) ntarget_boxes + 1, 4);
            for (int64_t i = 0; i < ntarget_boxes; ++i) {
                int32_t ibox = ttp_from_all[target_boxes[i]];
                int32_t s0 = raw.starts[ibox], s1 = raw.starts[ibox + 1];
                for (int32_t j = s0; j < s1; ++j) orc_ivec_push(&nl, raw.lists[j]);
                ncounts[i] = s1 - s0;
            }
            SFX(orc_finish_list)(&out->from_sep_close_bigger, ntarget_boxes, ncounts, &nl, 0);
            free(ncounts); free(ttp_from_all); free(raw.starts); free(raw.lists);
        } else {
            free(lc.data);
        }
        free(counts); free(ccounts);
    }

    out->status = ORC_OK;
    return ORC_OK;
}

#undef ORC_LOAD_CENTER
#undef ORC_WALK_DECL
#undef ORC_WALK_INIT
#undef ORC_WALK_GET_BOX_ID
#undef ORC_WALK_ADVANCE
#undef ORC_WALK_PUSH
#undef ORC_LEVEL_TO_RAD
