/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See boxtree_oracle_impl.h header.
 *
 * Sequential literal restatement of boxtree/area_query.py: peer lists
 * (:393-475), guiding box (:172-292), area query walker (:295-366), space
 * invader query (:613-651).  PARITY UNPINNED (see boxtree_oracle_impl.h).
 */

typedef struct {
    int32_t dims;
    int32_t nlevels;
    int64_t nboxes, aligned_nboxes;
    COORD_T root_extent;
    COORD_T bbox_min[ORC_MAXDIM];
    const COORD_T *box_centers;
    const uint8_t *box_levels;
    const int32_t *box_child_ids;
    const uint8_t *box_flags;
} SFX(orc_aq_tree);

#define AQ_LEVEL_TO_RAD(root_extent, level) \
    ((root_extent) * 1 / (COORD_T) (1 << ((level) + 1)))

/* PEER_LIST_FINDER_TEMPLATE: area_query.py:393-475 */
int SFX(orc_peer_lists)(const SFX(orc_aq_tree) *t, orc_built_list *out)
{
    const int dims = t->dims; const int C = 1 << dims;
    const int64_t aligned = t->aligned_nboxes, B = t->nboxes;
    orc_ivec l = {0};
    int32_t *counts = (int32_t *) calloc((size_t) B + 1, 4);
    for (int64_t box_id = 0; box_id < B; ++box_id) {
        int64_t n0 = l.n;
        COORD_T center[ORC_MAXDIM];
        for (int d = 0; d < dims; ++d) center[d] = t->box_centers[aligned * d + box_id];
        if (box_id == 0) {
            orc_ivec_push(&l, 0);
        } else {
            int level = t->box_levels[box_id];
            int32_t stack_box[128]; int stack_mnr[128];
            int size = 0; int32_t parent = 0; int mnr = 0; int go = 1;
            while (go) {
                int32_t wb = t->box_child_ids[mnr * aligned + parent];
                if (wb) {
                    COORD_T wc[ORC_MAXDIM];
                    for (int d = 0; d < dims; ++d) wc[d] = t->box_centers[aligned * d + wb];
                    int a_or_o = SFX(orc_is_adj)(dims, t->root_extent, center, level, wc, size + 1);
                    if (a_or_o) {
                        if (size + 1 == level) {
                            orc_ivec_push(&l, wb);
                        } else if (!(t->box_flags[wb] & BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES)) {
                            orc_ivec_push(&l, wb);
                        } else {
                            int must_be_peer = 1;
                            for (int m = 0; must_be_peer && m < C; ++m) {
                                int32_t nc = t->box_child_ids[m * aligned + wb];
                                if (nc) {
                                    COORD_T nwc[ORC_MAXDIM];
                                    for (int d = 0; d < dims; ++d)
                                        nwc[d] = t->box_centers[aligned * d + nc];
                                    must_be_peer &= !SFX(orc_is_adj)(dims, t->root_extent,
                                            center, level, nwc, size + 2);
                                }
                            }
                            if (must_be_peer) {
                                orc_ivec_push(&l, wb);
                            } else {
                                stack_box[size] = parent; stack_mnr[size] = mnr; ++size;
                                parent = wb; mnr = 0;
                                continue;
                            }
                        }
                    }
                }
                while (1) {
                    ++mnr;
                    if (mnr < C) break;
                    go = size > 0;
                    if (go) { --size; parent = stack_box[size]; mnr = stack_mnr[size]; }
                    else break;
                }
            }
        }
        counts[box_id] = (int32_t) (l.n - n0);
    }
    SFX(orc_finish_list)(out, B, counts, &l, 0);
    free(counts);
    return ORC_OK;
}

/* check_l_infty_ball_overlap: traversal.py:200-214 */
static inline int SFX(orc_ball_overlap)(const SFX(orc_aq_tree) *t, int32_t box,
        COORD_T ball_radius, const COORD_T *ball_center)
{
    const int dims = t->dims;
    int box_level = t->box_levels[box];
    COORD_T size_sum = AQ_LEVEL_TO_RAD(t->root_extent, box_level) + ball_radius;
    COORD_T max_dist = 0;
    for (int i = 0; i < dims; ++i) {
        COORD_T d = ball_center[i] - t->box_centers[t->aligned_nboxes * i + box];
        d = (d < 0) ? -d : d;
        max_dist = (d > max_dist) ? d : max_dist;
    }
    return max_dist <= size_sum;
}

/* find_guiding_box: area_query.py:179-291 */
static int32_t SFX(orc_guiding_box)(const SFX(orc_aq_tree) *t, const COORD_T *ball_center,
        COORD_T ball_radius)
{
    const int dims = t->dims; const int C = 1 << dims;
    int32_t box = 0;
    COORD_T query_center[ORC_MAXDIM], bbox_max[ORC_MAXDIM];
    for (int d = 0; d < dims; ++d) {
        bbox_max[d] = t->bbox_min[d] + (COORD_T) (t->root_extent / (1 + 1e-4));
        COORD_T c = ball_center[d];
        c = (c > t->bbox_min[d]) ? c : t->bbox_min[d];
        query_center[d] = (bbox_max[d] < c) ? bbox_max[d] : c;
    }
    COORD_T query_radius = 0;
    for (int mnr = 0; mnr < C; ++mnr) {
        for (int d = 0; d < dims; ++d) {
            COORD_T off = ((1 << (dims - 1 - d)) & mnr) ? +ball_radius : -ball_radius;
            COORD_T corner = ball_center[d] + off;
            corner = (corner > t->bbox_min[d]) ? corner : t->bbox_min[d];
            corner = (bbox_max[d] < corner) ? bbox_max[d] : corner;
            COORD_T dist = corner - query_center[d];
            dist = (dist < 0) ? -dist : dist;
            query_radius = (dist > query_radius) ? dist : query_radius;
        }
    }
    if (AQ_LEVEL_TO_RAD(t->root_extent, 0) / 2 >= query_radius) {
        for (unsigned box_level = 0;; ++box_level) {
            if (!(t->box_flags[box] & BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES)
                    || (AQ_LEVEL_TO_RAD(t->root_extent, box_level) / 2 < query_radius
                        && query_radius <= AQ_LEVEL_TO_RAD(t->root_extent, box_level)))
                break;
            int morton = 0;
            for (int d = 0; d < dims; ++d) {
                COORD_T off_scaled = (query_center[d] - t->bbox_min[d]) / t->root_extent;
                unsigned bits = (unsigned) (off_scaled * (COORD_T) (1U << (1 + box_level)));
                morton |= (int) (bits & 1U) << (dims - 1 - d);
            }
            int32_t next_box = t->box_child_ids[morton * t->aligned_nboxes + box];
            if (next_box) box = next_box;
            else break;
        }
    }
    return box;
}

/* AREA_QUERY_WALKER_BODY: area_query.py:295-366.  mode 0: collect leaves;
 * mode 1: space invader (atomic max of float32 l-inf centre distance, :629-650) */
static void SFX(orc_aq_walk)(const SFX(orc_aq_tree) *t, const int32_t *pl_starts,
        const int32_t *pl_lists, const COORD_T *ball_center, COORD_T ball_radius,
        orc_ivec *leaves, float *invader)
{
    const int dims = t->dims; const int C = 1 << dims;
    const int64_t aligned = t->aligned_nboxes;
    int32_t guiding = SFX(orc_guiding_box)(t, ball_center, ball_radius);
#define AQ_FOUND(leaf) do { \
        if (leaves) orc_ivec_push(leaves, (leaf)); \
        else { \
            COORD_T md_ = 0; \
            for (int d_ = 0; d_ < dims; ++d_) { \
                COORD_T dd_ = ball_center[d_] - t->box_centers[aligned * d_ + (leaf)]; \
                dd_ = (dd_ < 0) ? -dd_ : dd_; \
                md_ = (dd_ > md_) ? dd_ : md_; } \
            float f_ = (float) md_; \
            if (f_ > invader[(leaf)]) invader[(leaf)] = f_; \
        } } while (0)
    for (int32_t pb_i = pl_starts[guiding]; pb_i < pl_starts[guiding + 1]; ++pb_i) {
        int32_t peer = pl_lists[pb_i];
        if (!(t->box_flags[peer] & BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES)) {
            if (SFX(orc_ball_overlap)(t, peer, ball_radius, ball_center)) AQ_FOUND(peer);
        } else {
            int32_t stack_box[128]; int stack_mnr[128];
            int size = 0; int32_t parent = peer; int mnr = 0; int go = 1;
            while (go) {
                int32_t wb = t->box_child_ids[mnr * aligned + parent];
                if (wb) {
                    if (!(t->box_flags[wb] & BOX_HAS_SOURCE_OR_TARGET_CHILD_BOXES)) {
                        if (SFX(orc_ball_overlap)(t, wb, ball_radius, ball_center)) AQ_FOUND(wb);
                    } else {
                        stack_box[size] = parent; stack_mnr[size] = mnr; ++size;
                        parent = wb; mnr = 0;
                        continue;
                    }
                }
                while (1) {
                    ++mnr;
                    if (mnr < C) break;
                    go = size > 0;
                    if (go) { --size; parent = stack_box[size]; mnr = stack_mnr[size]; }
                    else break;
                }
            }
        }
    }
#undef AQ_FOUND
}

int SFX(orc_area_query)(const SFX(orc_aq_tree) *t, const int32_t *pl_starts,
        const int32_t *pl_lists, int64_t nballs, const COORD_T *const *ball_centers,
        const COORD_T *ball_radii, orc_built_list *out)
{
    orc_ivec l = {0};
    int32_t *counts = (int32_t *) calloc((size_t) nballs + 1, 4);
    for (int64_t i = 0; i < nballs; ++i) {
        int64_t n0 = l.n;
        COORD_T c[ORC_MAXDIM];
        for (int d = 0; d < t->dims; ++d) c[d] = ball_centers[d][i];
        SFX(orc_aq_walk)(t, pl_starts, pl_lists, c, ball_radii[i], &l, NULL);
        counts[i] = (int32_t) (l.n - n0);
    }
    SFX(orc_finish_list)(out, nballs, counts, &l, 0);
    free(counts);
    return ORC_OK;
}

int SFX(orc_space_invader)(const SFX(orc_aq_tree) *t, const int32_t *pl_starts,
        const int32_t *pl_lists, int64_t nballs, const COORD_T *const *ball_centers,
        const COORD_T *ball_radii, float *out /* [nboxes], zero-initialised */)
{
    for (int64_t i = 0; i < nballs; ++i) {
        COORD_T c[ORC_MAXDIM];
        for (int d = 0; d < t->dims; ++d) c[d] = ball_centers[d][i];
        SFX(orc_aq_walk)(t, pl_starts, pl_lists, c, ball_radii[i], NULL, out);
    }
    return ORC_OK;
}

#undef AQ_LEVEL_TO_RAD
