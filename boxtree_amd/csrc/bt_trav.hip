// FMM interaction-list generation for gfx950 (boxtree/traversal.py:1969-2345).
//
// Every list is produced count -> exclusive scan -> fill, like pyopencl's
// ListOfListsBuilder, by kernels that evaluate the reference's box predicates
// (traversal.py:255-320, 933-972) with identical floating-point expressions
// (-ffp-contract=off).  List 3 ("from_sep_smaller") is generated for all source
// levels in ONE walk per target box and bucketed by level afterwards, instead
// of nlevels relaunches (traversal.py:2203-2216).
#include "bt_common.hpp"
#include "bt_prims.hpp"
#include "bt_geom.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <vector>

using namespace bt;

namespace {


template <class T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920928955078125e-07f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

template <class T, int D>
struct TravArgs {
    const Node<T, D> *nodes;      // [nboxes]
    const int32_t *child_t;       // [nboxes][C]  (children of a box contiguous)
    const int32_t *parent;
    const T *tgt_bbox_min, *tgt_bbox_max;
    const int32_t *src_counts_cumul;
    int64_t aligned;
    int32_t nboxes;
    T root_extent;
    T stick_out_factor;
    int nway;
    int crit;
    int32_t min_nsources_cumul;
    int targets_have_extent;
    int close_lists_exist;
    int fast;                     // structure verified by check_pack_kernel
    // lists built earlier
    const int32_t *target_boxes; int32_t ntarget_boxes;
    const int32_t *ttp_boxes; int32_t nttp;
    const int32_t *coll_starts, *coll_lists;
    // colleagues that are source boxes, as fixed-stride rows (single-pass colleague
    // kernel only; srccoll_stride == 0 otherwise)
    const int32_t *srccoll_rows, *srccoll_cnt;
    int srccoll_stride;
    uint32_t srccoll_id_mask;       // row entries carry lattice codes above these bits
    uint32_t srccoll_src_bit;       // the rows hold all colleagues, source boxes carry this bit (or 0)
    const int8_t *target_mask;      // sharded traversals: boxes whose lists are wanted
    const int32_t *dfs_rank;        // preorder rank (parent-colleague kernels)
};

template <class T, int D>
__device__ __forceinline__ int box_level(const TravArgs<T, D> &a, int32_t box)
{
    return (int) (a.nodes[box].lf & 0xffu);
}

template <class T, int D>
__device__ __forceinline__ uint8_t box_flags(const TravArgs<T, D> &a, int32_t box)
{
    return (uint8_t) (a.nodes[box].lf >> 8);
}

template <int D, class T>
__device__ __forceinline__ int32_t child_of(const TravArgs<T, D> &a, int32_t box, int m)
{
    return (int32_t) ((uint32_t) a.child_t[(int64_t) box * (1 << D) + m] & CH_ID_MASK);
}

template <class T, int D>
__device__ __forceinline__ void load_center(const TravArgs<T, D> &a, int32_t box, T *c)
{
#pragma unroll
    for (int i = 0; i < D; ++i) c[i] = a.nodes[box].c[i];
}

// ---- T3 colleagues: traversal.py:398-464 ---------------------------------------

template <class T, int D, class E>
__device__ __forceinline__ void gen_colleagues(const TravArgs<T, D> &a, int32_t box_id, E &emit)
{
    constexpr int C = 1 << D;
    if (box_id == 0) return;
    T center[D];
    load_center(a, box_id, center);
    const int level = box_level(a, box_id);
    Walk w(s_walk_lds + threadIdx.x);
    w.init(0);
    while (w.go) {
        const int32_t wb = child_of<D>(a, w.parent, w.mnr);
        if (wb) {
            T wc[D];
            load_center(a, wb, wc);
            const bool a_or_o = adj_nbhd<T, D>(a.root_extent, center, level, (T) a.nway, wc,
                                               box_level(a, wb));
            if (a_or_o) {
                // The reference (traversal.py:438-452) pushes box_id itself and then
                // walks its whole subtree without ever emitting (nothing below
                // `level` can be on `level`); stopping at `level` gives the same list.
                if (w.size + 1 == level) {
                    if (wb != box_id) emit(wb);
                } else {
                    w.push(wb);
                    continue;
                }
            }
        }
        w.template advance<C>();
    }
}

// ---- T4 list 1: traversal.py:470-550 ---------------------------------------------

template <class T, int D, class E>
__device__ __forceinline__ void gen_list1(const TravArgs<T, D> &a, int32_t tbn, E &emit)
{
    constexpr int C = 1 << D;
    const int32_t box_id = a.target_boxes[tbn];
    T center[D];
    load_center(a, box_id, center);
    const int level = box_level(a, box_id);
    if (box_flags(a, 0) & BT_BOX_IS_SOURCE_BOX) emit(0);
    Walk w(s_walk_lds + threadIdx.x);
    w.init(0);
    while (w.go) {
        const int32_t wb = child_of<D>(a, w.parent, w.mnr);
        if (wb) {
            T wc[D];
            load_center(a, wb, wc);
            if (adj<T, D>(a.root_extent, center, level, wc, box_level(a, wb))) {
                const uint8_t fl = box_flags(a, wb);
                if (fl & BT_BOX_IS_SOURCE_BOX) emit(wb);
                if (fl & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
                    w.push(wb);
                    continue;
                }
            }
        }
        w.template advance<C>();
    }
}

// ---- T5 list 2: traversal.py:556-601 ----------------------------------------------

template <class T, int D, class E>
__device__ __forceinline__ void gen_list2(const TravArgs<T, D> &a, int32_t it, E &emit)
{
    constexpr int C = 1 << D;
    const int32_t box_id = a.ttp_boxes[it];
    T center[D];
    load_center(a, box_id, center);
    const int level = box_level(a, box_id);
    const int32_t parent = a.parent[box_id];
    if (parent == box_id) return;
    const int32_t ps = a.coll_starts[parent], pe = a.coll_starts[parent + 1];
    for (int32_t i = ps; i < pe; ++i) {
        const int32_t pnf = a.coll_lists[i];
        for (int m = 0; m < C; ++m) {
            const int32_t sib = child_of<D>(a, pnf, m);
            if (sib == 0) continue;
            T sc[D];
            load_center(a, sib, sc);
            const bool sep = !adj_nbhd<T, D>(a.root_extent, center, level, (T) a.nway, sc,
                                             box_level(a, sib));
            if (sep) emit(sib);
        }
    }
}

// ---- T6 list 3 (+ close): traversal.py:607-875, all source levels in one walk ------

struct NoL1 {
    static constexpr bool active = false;
    __device__ __forceinline__ void operator()(int32_t) {}
};

// E1 (optional): receives the adjacent source boxes met on the way -- the part of
// list 1 that lies at or below the colleagues' level (the walk of
// traversal.py:501-547 restricted to the colleagues' subtrees visits exactly the
// boxes this walk descends through).
template <class T, int D, class EM, class EC, class E1>
__device__ __forceinline__ void gen_list3(const TravArgs<T, D> &a, int32_t tbn, EM &emit_main,
                                          EC &emit_close, E1 &emit_l1,
                                          int32_t coll_first = 0, int32_t coll_count = -1)
{
    constexpr int C = 1 << D;
    const int32_t tgt = a.target_boxes[tbn];
    T tc[D];
    load_center(a, tgt, tc);
    const int tl = box_level(a, tgt);

    T stickout_rad = 0;
    T ext_center[D], radii_vec[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { ext_center[i] = 0; radii_vec[i] = 0; }
    if (a.targets_have_extent) {
        if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_STATIC_L2) {
            stickout_rad = (1 + a.stick_out_factor) * level_to_rad(a.root_extent, tl);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {          // load_true_box_extent, :177-198
                const T mn = a.tgt_bbox_min[i * a.aligned + tgt];
                const T mx = a.tgt_bbox_max[i * a.aligned + tgt];
                ext_center[i] = ((T) 0.5) * (mn + mx);
                radii_vec[i] = ((T) 0.5) * (mx - mn);
            }
        }
    }

    int32_t s0 = a.coll_starts[tgt], s1 = a.coll_starts[tgt + 1];
    if (coll_count >= 0) {          // only colleagues [coll_first, coll_first + coll_count)
        s0 += coll_first;
        s1 = (s0 + coll_count < s1) ? s0 + coll_count : s1;
    }
    for (int32_t i = s0; i < s1; ++i) {
        const int32_t nws = a.coll_lists[i];
        if (nws == tgt) continue;
        const uint8_t cfl = box_flags(a, nws);
        if (E1::active && (cfl & BT_BOX_IS_SOURCE_BOX)) {
            T cc[D];
            load_center(a, nws, cc);
            if (adj<T, D>(a.root_extent, tc, tl, cc, box_level(a, nws))) emit_l1(nws);
        }
        // nothing below a colleague without source children can be emitted
        // (flag consistency is part of the verified structure)
        if (a.fast && !(cfl & BT_BOX_HAS_SOURCE_CHILD_BOXES)) continue;
        Walk w(s_walk_lds + threadIdx.x);
        w.init(nws);
        while (w.go) {
            const int32_t wb = child_of<D>(a, w.parent, w.mnr);
            const uint8_t fl = box_flags(a, wb);
            if (wb && (fl & (BT_BOX_IS_SOURCE_BOX | BT_BOX_HAS_SOURCE_CHILD_BOXES))) {
                T wc[D];
                load_center(a, wb, wc);
                const int wl = box_level(a, wb);
                const bool in_list_1 = adj<T, D>(a.root_extent, tc, tl, wc, wl);
                if (in_list_1) {
                    if (E1::active && (fl & BT_BOX_IS_SOURCE_BOX)) emit_l1(wb);
                    if (fl & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
                        w.push(wb);
                        continue;
                    }
                } else {
                    bool meets;
                    if (!a.targets_have_extent) {
                        meets = true;
                    } else if (a.crit == BT_CRIT_STATIC_LINF) {
                        const T source_rad = level_to_rad(a.root_extent, wl);
                        T l_inf = 0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            T d = tc[k] - wc[k];
                            d = (d < 0) ? -d : d;
                            const T v = d - stickout_rad - source_rad;
                            l_inf = (v > l_inf) ? v : l_inf;
                        }
                        meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                    } else if (a.crit == BT_CRIT_PRECISE_LINF) {
                        const T source_rad = level_to_rad(a.root_extent, wl);
                        T l_inf = 0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            T d = ext_center[k] - wc[k];
                            d = (d < 0) ? -d : d;
                            const T v = d - radii_vec[k] - source_rad;
                            l_inf = (v > l_inf) ? v : l_inf;
                        }
                        meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                    } else {
                        const T source_rad = level_to_rad(a.root_extent, wl);
                        T l2sq = 0;
#pragma unroll
                        for (int k = 0; k < D; ++k) {
                            const T d = tc[k] - wc[k];
                            l2sq = l2sq + d * d;
                        }
                        const T rhs = sqrt(l2sq) - sqrt((T) D) * stickout_rad - source_rad;
                        meets = ((2 - 8 * Eps<T>::v) * source_rad <= rhs);
                    }
                    const bool force_close = a.close_lists_exist
                        && (a.src_counts_cumul[wb] < a.min_nsources_cumul);
                    if (meets && !force_close) {
                        emit_main(wl, wb);
                    } else if (a.close_lists_exist) {
                        if (fl & BT_BOX_IS_SOURCE_BOX) emit_close(wb);
                        if (fl & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
                            w.push(wb);
                            continue;
                        }
                    }
                }
            }
            w.template advance<C>();
        }
    }
}

// ---- T7 list 4 (+ close): traversal.py:931-1146 ---------------------------------------

template <class T, int D>
__device__ __forceinline__ bool meets_sep_bigger(T root_extent, const T *tc, int tl, const T *sc,
                                                 int sl, T stick_out_factor)
{
    const T target_rad = level_to_rad(root_extent, tl);
    const T source_rad = level_to_rad(root_extent, sl);
    const T max_allowed = (3 * (1 + stick_out_factor) * target_rad + source_rad);
    T l_inf = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T d = tc[i] - sc[i];
        d = (d < 0) ? -d : d;
        l_inf = (d > l_inf) ? d : l_inf;
    }
    return l_inf >= max_allowed * (1 - 8 * Eps<T>::v);
}

template <class T, int D, class EM, class EC>
__device__ __forceinline__ void gen_list4(const TravArgs<T, D> &a, int32_t it, EM &emit_main,
                                          EC &emit_close)
{
    const int32_t tgt = a.ttp_boxes[it];
    T tc[D];
    load_center(a, tgt, tc);
    const int tl = box_level(a, tgt);
    if (tl == 0) return;
    const int32_t tparent = a.parent[tgt];
    const int pl = tl - 1;
    T pc[D];
    load_center(a, tparent, pc);
    // (a sharded traversal marks with 2 a shared box whose own targets another rank holds: it has
    // lists as a parent of target boxes here, but it is not one of this rank's target boxes -- a
    // close list made for it would be dropped by the re-indexing to target boxes, and counted)
    const uint8_t tflags = (a.target_mask && a.target_mask[tgt] != 1)
        ? (uint8_t) (box_flags(a, tgt) & ~BT_BOX_IS_TARGET_BOX) : box_flags(a, tgt);
    int wl; int32_t cur;
    if (a.nway == 1) { wl = tl - 1; cur = tparent; }
    else { wl = tl; cur = tgt; }
    for (; wl != 0; --wl, cur = a.parent[cur]) {
        const bool rows = a.srccoll_stride != 0;
        const int64_t s0 = rows ? (int64_t) cur * a.srccoll_stride : a.coll_starts[cur];
        // (one row family: srccoll_cnt holds the mask of the row's source entries)
        const bool masked = rows && a.srccoll_src_bit;
        uint32_t smask = masked ? (uint32_t) a.srccoll_cnt[cur] : 0u;
        const int64_t s1 = masked ? s0 + __popc(smask) : rows ? s0 + a.srccoll_cnt[cur] : a.coll_starts[cur + 1];
        const int32_t *src = rows ? a.srccoll_rows : a.coll_lists;
        for (int64_t ii = s0; ii < s1; ++ii) {
            const int64_t i = masked ? s0 + __builtin_ctz(smask) : ii;
            smask &= smask - 1u;
            const int32_t sb = rows ? (int32_t) ((uint32_t) src[i] & a.srccoll_id_mask) : src[i];
            if (!rows && !(box_flags(a, sb) & BT_BOX_IS_SOURCE_BOX)) continue;
            T sc[D];
            load_center(a, sb, sc);
            if (adj<T, D>(a.root_extent, tc, tl, sc, wl)) continue;
            if (a.close_lists_exist) {
                if (!meets_sep_bigger<T, D>(a.root_extent, tc, tl, sc, wl, a.stick_out_factor)) {
                    if (tflags & BT_BOX_IS_TARGET_BOX) emit_close(sb);
                    continue;
                }
            }
            const bool in_parent_list_1 = adj<T, D>(a.root_extent, pc, pl, sc, wl);
            bool would = !in_parent_list_1;
            if (a.nway > 1) would = would && (wl < tl);
            if (would) {
                if (a.close_lists_exist) {
                    if (!meets_sep_bigger<T, D>(a.root_extent, pc, pl, sc, wl, a.stick_out_factor))
                        emit_main(sb);
                }
            } else {
                emit_main(sb);
            }
        }
    }
}

// ---- kernels --------------------------------------------------------------------------

enum { GEN_COLL = 0, GEN_L1 = 1, GEN_L2 = 2 };

template <class T, int D, int WHICH, bool FILL>
__global__ __launch_bounds__(256) void list_kernel(TravArgs<T, D> a, int32_t n,
        int32_t *counts_or_starts, int32_t *lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (!FILL) {
        CountEmit e;
        if (WHICH == GEN_COLL) gen_colleagues<T, D>(a, i, e);
        else if (WHICH == GEN_L1) gen_list1<T, D>(a, i, e);
        else gen_list2<T, D>(a, i, e);
        counts_or_starts[i] = e.n;
    } else {
        WriteEmit e{lists + counts_or_starts[i]};
        if (WHICH == GEN_COLL) gen_colleagues<T, D>(a, i, e);
        else if (WHICH == GEN_L1) gen_list1<T, D>(a, i, e);
        else gen_list2<T, D>(a, i, e);
    }
}

// per-level counters / cursors of list 3 live in LDS columns as well
struct L3CountMain {
    int32_t *c;        // LDS column, element l at c[l * WALK_THREADS]
    __device__ __forceinline__ void operator()(int lev, int32_t) { ++c[lev * WALK_THREADS]; }
};
struct L3WriteMain {
    int32_t *lists;
    int32_t *cur;      // LDS column
    __device__ __forceinline__ void operator()(int lev, int32_t b)
    {
        lists[cur[lev * WALK_THREADS]++] = b;
    }
};

// counts layout: [nlevels][ntb] (level-major) for the main list, [ntb] for close
template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void list3_kernel(TravArgs<T, D> a, int32_t ntb, int nlevels,
        int walk_cap, int32_t *main_cs, int32_t *main_lists, int32_t *close_cs, int32_t *close_lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ntb) return;
    int32_t *lvl = s_walk_lds + walk_cap * WALK_THREADS + threadIdx.x;
    if (!FILL) {
        L3CountMain em{lvl};
        for (int l = 0; l < nlevels; ++l) lvl[l * WALK_THREADS] = 0;
        CountEmit ec;
        NoL1 no1;
        gen_list3<T, D>(a, i, em, ec, no1);
        for (int l = 0; l < nlevels; ++l) main_cs[(int64_t) l * ntb + i] = lvl[l * WALK_THREADS];
        if (close_cs) close_cs[i] = ec.n;
    } else {
        L3WriteMain em{main_lists, lvl};
        for (int l = 0; l < nlevels; ++l) lvl[l * WALK_THREADS] = main_cs[(int64_t) l * ntb + i];
        WriteEmit ec{close_lists ? close_lists + close_cs[i] : nullptr};
        CountEmit dummy;
        NoL1 no1;
        if (close_lists) gen_list3<T, D>(a, i, em, ec, no1);
        else gen_list3<T, D>(a, i, em, dummy, no1);
    }
}

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void list4_kernel(TravArgs<T, D> a, int32_t n,
        int32_t *main_cs, int32_t *main_lists, int32_t *close_cs, int32_t *close_lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (!FILL) {
        CountEmit em, ec;
        gen_list4<T, D>(a, i, em, ec);
        main_cs[i] = em.n;
        if (close_cs) close_cs[i] = ec.n;
    } else {
        WriteEmit em{main_lists + main_cs[i]};
        WriteEmit ec{close_lists ? close_lists + close_cs[i] : nullptr};
        CountEmit dummy;
        if (close_lists) gen_list4<T, D>(a, i, em, ec);
        else gen_list4<T, D>(a, i, em, dummy);
    }
}

struct ScanI32 {
    const int32_t *p;
    __device__ int32_t operator()(int64_t i) const { return p[i]; }
};

struct HostWords { int32_t v[BT_MAX_LEVELS + 2]; };

// out[0 .. n) = w; zeros[0 .. nzeros) = 0
__global__ __launch_bounds__(128) void store_host_words_kernel(HostWords w, int n, int32_t *out,
                                                               int32_t *zeros, int nzeros)
{
    int32_t x = 0;
#pragma unroll
    for (int i = 0; i < BT_MAX_LEVELS + 2; ++i) x = ((int) threadIdx.x == i) ? w.v[i] : x;
    if ((int) threadIdx.x < n) out[threadIdx.x] = x;
    if ((int) threadIdx.x < nzeros) zeros[threadIdx.x] = 0;
}

// ---- box lists by flag (T1, traversal.py:326-355) ---------------------------------------

struct FlagPred {
    const uint8_t *flags;
    const int8_t *mask;     // optional
    uint8_t bits;
    uint8_t exact;          // the mask value must be 1 (a sharded traversal marks with 2 the shared
                            // boxes whose own targets another rank holds: parents of target boxes
                            // here, target boxes there)
    __device__ int32_t operator()(int64_t i) const
    {
        return ((flags[i] & bits) && (!mask || (exact ? mask[i] == 1 : mask[i] != 0))) ? 1 : 0;
    }
};

__global__ __launch_bounds__(256) void compact_kernel(FlagPred pr, int32_t n, const int32_t *pos,
                                                      int32_t *out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (pr(i)) out[pos[i]] = i;
}

// What the host reads in the first wait of a traversal, in one block ("mail"):
// rows[k][l] = pos[k][level_start_box_nrs[l]] for the five box lists (the number of listed
// boxes before the level's first box), the flags of check_pack_kernel (written there),
// and the root's centre -- one kernel and one read instead of five gathers, a row copy and
// five reads.
struct BoxListMarks {
    const int32_t *pos[5];        // scans of the list predicates, [nboxes + 1]
    int32_t *rows;                // [5][nlevels + 1]
    const void *centers;          // [d][aligned]
    int64_t aligned;
    void *root_center;            // [d] of the coordinate type
    int nlevels, dims, csize;
};

__global__ void box_list_marks_kernel(BoxListMarks m, const int32_t *level_start_box_nrs)
{
    const int t = threadIdx.x;
    const int n1 = m.nlevels + 1;
    for (int i = t; i < 5 * n1; i += blockDim.x) {
        const int k = i / n1, l = i % n1;
        m.rows[i] = m.pos[k][level_start_box_nrs[l]];
    }
    if (t < m.dims) {
        if (m.csize == 8) ((double *) m.root_center)[t] = ((const double *) m.centers)[(int64_t) t * m.aligned];
        else ((float *) m.root_center)[t] = ((const float *) m.centers)[(int64_t) t * m.aligned];
    }
}

// the box lists themselves: list k gets the boxes with pred k, in box order
struct BoxLists {
    FlagPred pred[5];
    const int32_t *pos[5];
    int32_t *out[5];              // null: not wanted (a list that shares another's array)
};

__global__ __launch_bounds__(256) void compact_lists_kernel(BoxLists bl, int32_t n)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (bl.out[k] && bl.pred[k](i)) bl.out[k][bl.pos[k][i]] = i;
}

// up to 40 array copies in one launch (blockIdx.y = the copy): the result arrays that
// were made in scratch before the caller's block existed
struct SpanCopies {
    enum { MAX = 40 };
    const int32_t *src[MAX];
    int32_t *dst[MAX];
    int64_t n[MAX];
    int count;
};

__global__ __launch_bounds__(256) void copy_spans_kernel(SpanCopies c)
{
    const int k = blockIdx.y;
    const int32_t *src = c.src[k];
    int32_t *dst = c.dst[k];
    const int64_t n = c.n[k];
    for (int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256)
        dst[i] = src[i];
}

// ---- list 3 per-level post-processing (BuiltList with eliminate_empty) --------------------

struct NonEmptyPred {
    const int32_t *starts;    // [ntb+1] slice of the level (global offsets)
    __device__ int32_t operator()(int64_t i) const { return starts[i + 1] > starts[i] ? 1 : 0; }
};

// values at the level boundaries of two flat [nlevels*ntb + 1] arrays -> small arrays
// (rows exist for the levels lev0 .. nlevels-1; the levels above them are empty)
__global__ void l3_level_marks_kernel(int nlevels, int64_t ntb, const int32_t *starts,
                                      const int32_t *scan, int32_t *out /* [2][nlevels+1] */, int lev0)
{
    const int l = threadIdx.x;
    if (l > nlevels) return;
    const int r = l > lev0 ? l - lev0 : 0;
    out[l] = starts[(int64_t) r * ntb];
    out[nlevels + 1 + l] = scan[(int64_t) r * ntb];
}

// all source levels in one launch (thread = (level, target box number))
struct L3CompressAll {
    int32_t *starts[BT_MAX_LEVELS], *nonempty[BT_MAX_LEVELS], *cidx[BT_MAX_LEVELS],
            *tboxes[BT_MAX_LEVELS];
    int32_t lev_base[BT_MAX_LEVELS], cidx_base[BT_MAX_LEVELS], lev_count[BT_MAX_LEVELS];
    int lev0;                      // TravState::l3_lo
};

__global__ __launch_bounds__(256) void l3_compress_all_kernel(int32_t ntb, int nlevels,
        const int32_t *l3_starts /* [nlevels*ntb + 1] */, const int32_t *l3_cidx,
        const int32_t *target_boxes, L3CompressAll o)
{
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int l = (int) (gid / (ntb + 1));
    const int32_t i = (int32_t) (gid % (ntb + 1));
    if (l >= nlevels) return;
    if (l < o.lev0) {
        // a source level without rows: nothing there
        if (o.cidx[l]) o.cidx[l][i] = 0;
        if (i == ntb) o.starts[l][0] = 0;
        return;
    }
    const int32_t *lev_starts = l3_starts + (int64_t) (l - o.lev0) * ntb;
    const int32_t *cidx = l3_cidx + (int64_t) (l - o.lev0) * ntb;
    if (o.cidx[l]) o.cidx[l][i] = cidx[i] - o.cidx_base[l];
    if (i == ntb) {
        o.starts[l][cidx[ntb] - o.cidx_base[l]] = o.lev_count[l];
        return;
    }
    if (lev_starts[i + 1] > lev_starts[i]) {
        const int32_t k = cidx[i] - o.cidx_base[l];
        o.starts[l][k] = lev_starts[i] - o.lev_base[l];
        o.nonempty[l][k] = i;
        o.tboxes[l][k] = target_boxes[i];
    }
}

// ---- close-bigger re-indexing (_ListMerger, traversal.py:1259-1344) --------------------------

__global__ __launch_bounds__(256) void reverse_index_kernel(const int32_t *list, int32_t n, int32_t *out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[list[i]] = i;
}

struct MergeCount {
    const int32_t *target_boxes, *ttp_from_all, *raw_starts;
    __device__ int32_t operator()(int64_t i) const
    {
        const int32_t ib = ttp_from_all[target_boxes[i]];
        return raw_starts[ib + 1] - raw_starts[ib];
    }
};

__global__ __launch_bounds__(256) void merge_copy_kernel(int32_t n, MergeCount mc,
        const int32_t *raw_lists, const int32_t *new_starts, int32_t *new_lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t ib = mc.ttp_from_all[mc.target_boxes[i]];
    const int32_t s = mc.raw_starts[ib], e = mc.raw_starts[ib + 1];
    int32_t o = new_starts[i];
    for (int32_t j = s; j < e; ++j) new_lists[o++] = raw_lists[j];
}

#include "bt_trav_fast.hpp"
#include "bt_trav_v2.hpp"

// ---- merging CSR lists row by row (_ListMerger, traversal.py:1153-1344) ----------

struct MergeRows {
    int nlists;
    const int32_t *starts[4];
    const int32_t *lists[4];
};

struct MergeRowCount {
    MergeRows m;
    __device__ int32_t operator()(int64_t i) const
    {
        int32_t c = 0;
        for (int k = 0; k < m.nlists; ++k) c += m.starts[k][i + 1] - m.starts[k][i];
        return c;
    }
};

__global__ __launch_bounds__(256) void merge_rows_kernel(int32_t n, MergeRows m,
        const int32_t *new_starts, int32_t *new_lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int32_t o = new_starts[i];
    for (int k = 0; k < m.nlists; ++k) {
        const int32_t s = m.starts[k][i], e = m.starts[k][i + 1];
        for (int32_t j = s; j < e; ++j) new_lists[o++] = m.lists[k][j];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------

struct CsrList {
    Buf<int32_t> starts, lists;
    int64_t n = 0, total = 0;
};

// list 2 of one level still in its scratch rows (single-pass colleague kernel): the
// compaction into the CSR runs at export time, straight into the caller's buffer
struct L2Pending {
    Buf<int32_t> rows, rel;
    int32_t nb = 0;
    int64_t base = 0;
    int stride = 0;
};

struct TravState {
    bt_trav_params p{};
    std::vector<L2Pending> l2_pending;
    int nlevels = 0;
    bool with_extent = false;
    Buf<int32_t> source_boxes, target_boxes_buf, source_parent_boxes, ttp_boxes;
    Buf<int32_t> srccoll_rows, srccoll_cnt;
    int64_t nsb = 0, ntb = 0, nspb = 0, nttp = 0;
    const int32_t *target_boxes = nullptr;
    Buf<int32_t> lev_starts;           // [4][nlevels+1]
    std::vector<int32_t> h_lev_starts; // host copy
    Buf<int32_t> parent_boxes;         // boxes with children, in box order (row 4 of lev_starts)
    int64_t nparents = 0;
    Buf<int32_t> d_level_start_box_nrs;
    CsrList coll, l1, l2, l4, close_smaller, close_bigger;
    // list 3: flat level-major
    int l3_lo = 0;                     // first source level with a row in the two arrays below
    Buf<int32_t> l3_starts;            // [(nlevels - l3_lo)*ntb + 1]
    Buf<int32_t> l3_lists;
    Buf<int32_t> l3_cidx;              // flat scan [nlevels*ntb + 1]
    std::vector<int64_t> l3_level_base, l3_level_count, l3_nonempty, l3_cidx_base;
    Buf<int32_t> subtree_size, dfs_rank, box_of_rank, src_rank_prefix, src_by_rank;
    Buf<unsigned char> nodes;          // packed Node<T, D>[nboxes]
    Buf<int32_t> child_t;              // [nboxes][C]
    Buf<uint64_t> child8;              // [nboxes] Kids (bt_trav_v2.hpp)
    // one-block output (bt_traversal_build_packed)
    bt_alloc_fn packed_alloc = nullptr;
    void *packed_user = nullptr;
    bt_trav_packed *packed = nullptr;
    int32_t *arena = nullptr;          // the caller's block once it exists
    bt_trav_sizes sizes{};
    bool fast = false;
    bool lattice = false;              // bt_trav_v2.hpp kernels apply
    bool has_blocks = true;            // some target box has source boxes below it
    Buf<unsigned char> cells;          // ICell[nboxes]
    std::vector<std::pair<const char *, hipEvent_t>> events;
    bool built = false;
};

void bt_free_trav_state(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    if (ctx->trav) {
        for (auto &e : ctx->trav->events) (void) hipEventDestroy(e.second);
        delete ctx->trav;
        ctx->trav = nullptr;
    }
}

int bt_trav_stage_times(bt_context *ctx, bt_stage_times *out, int n)
{
    bt::CallScope bt_call_scope_(ctx);
    TravState *st = ctx->trav;
    if (!st) return n;
    if (!st->events.empty()) (void) hipEventSynchronize(st->events.back().second);
    for (size_t i = 1; i < st->events.size() && n < BT_NUM_STAGES; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, st->events[i - 1].second, st->events[i].second) != hipSuccess) {
            (void) hipGetLastError();
            ms = -1.f;
        }
        out->ms[n] = ms;
        out->name[n] = st->events[i].first;
        ++n;
    }
    return n;
}

namespace {

inline unsigned nblk(int64_t n) { return (unsigned) std::max<int64_t>(1, div_up(n, 256)); }

int tmark(bt_context *ctx, TravState *st, const char *name)
{
    host_trace(name);
    if (!ctx->stage_timing) return BT_OK;
    hipEvent_t e;
    BT_HIP_CHECK(hipEventCreate(&e));
    BT_HIP_CHECK(hipEventRecord(e, ctx->stream));
    st->events.push_back({name, e});
    return BT_OK;
}

int read_i32(bt_context *ctx, const int32_t *d, int32_t *h)
{
    BT_CHECK(bt::d2h(ctx, h, d, 4));
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

// exclusive scan of list lengths into int32 starts (n+1 entries), accumulated in 64
// bits: the total comes back exactly and a list beyond the int32 CSR limit of the
// reference is an error instead of a wrapped count
template <class F>
int scan_list_counts(bt_context *ctx, F f, int64_t n, int32_t *starts, int64_t *total)
{
    Buf<int64_t> d_total;
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, f, n, starts, d_total.get(), true)));
    int64_t t = 0;
    BT_CHECK(bt::d2h(ctx, &t, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    if (t >= ((int64_t) 1 << 31)) {
        set_error("interaction list exceeds 2^31-1 entries (int32 CSR limit of the reference): "
                  "%lld", (long long) t);
        return BT_ERR_UNSUPPORTED;
    }
    *total = t;
    return BT_OK;
}

// counts (in `cs`, n entries) -> exclusive starts in place (n+1 entries) + total on host
int counts_to_starts(bt_context *ctx, Buf<int32_t> &cs, int64_t n, int64_t *total)
{
    Buf<int32_t> tmp;
    BT_CHECK(tmp.alloc(ctx->pool, n + 1));
    BT_CHECK(scan_list_counts(ctx, ScanI32{cs.get()}, n, tmp.get(), total));
    cs.swap(tmp);
    return BT_OK;
}

int compact_boxes(bt_context *ctx, const bt_trav_params &p, uint8_t bits, const int8_t *mask,
                  Buf<int32_t> &out, int64_t *n_out)
{
    const int64_t B = p.nboxes;
    FlagPred pr{p.box_flags, mask, bits};
    Buf<int32_t> pos;
    BT_CHECK(pos.alloc(ctx->pool, B + 1));
    BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, pr, B, pos.get(), (int32_t *) nullptr, true)));
    int32_t t = 0;
    BT_CHECK(read_i32(ctx, pos.get() + B, &t));
    BT_CHECK(out.alloc(ctx->pool, t));
    compact_kernel<<<nblk(B), 256, 0, ctx->stream>>>(pr, (int32_t) B, pos.get(), out.get());
    *n_out = t;
    return BT_OK;
}

// per-level bases and the compressed (non-empty) indexing of list 3
int l3_postprocess(bt_context *ctx, TravState *st)
{
    const int nlevels = st->nlevels;
    const int64_t ntb = st->ntb;
    const int64_t nflat = (int64_t) nlevels * ntb;
    // One exclusive scan of "list (level, target box) is non-empty" over the flat
    // [level][target box] layout; level l's compressed indices are the slice
    // [l*ntb, (l+1)*ntb] minus the value at its start.
    BT_CHECK(st->l3_cidx.alloc(ctx->pool, nflat + 1));
    NonEmptyPred ne{st->l3_starts.get()};
    BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, ne, nflat, st->l3_cidx.get(),
                                                      (int32_t *) nullptr, true)));
    Buf<int32_t> marks;
    BT_CHECK(marks.alloc(ctx->pool, 2 * (nlevels + 1)));
    l3_level_marks_kernel<<<1, 128, 0, ctx->stream>>>(nlevels, ntb, st->l3_starts.get(),
                                                     st->l3_cidx.get(), marks.get(), st->l3_lo);
    std::vector<int32_t> h((size_t) 2 * (nlevels + 1));
    BT_CHECK(bt::d2h(ctx, h.data(), marks.get(), h.size() * 4));
    BT_CHECK(bt::sync_stream(ctx));
    st->l3_level_base.assign((size_t) nlevels, 0);
    st->l3_level_count.assign((size_t) nlevels, 0);
    st->l3_nonempty.assign((size_t) nlevels, 0);
    st->l3_cidx_base.assign((size_t) nlevels, 0);
    for (int l = 0; l < nlevels; ++l) {
        st->l3_level_base[l] = h[l];
        st->l3_level_count[l] = h[l + 1] - h[l];
        st->l3_cidx_base[l] = h[nlevels + 1 + l];
        st->l3_nonempty[l] = h[nlevels + 1 + l + 1] - h[nlevels + 1 + l];
    }
    return BT_OK;
}

template <class U>
int grow_buf(bt_context *ctx, Buf<U> &buf, int64_t used, int64_t need)
{
    if (need <= buf.size()) return BT_OK;
    int64_t nc = std::max<int64_t>(buf.size() * 2, 1024);
    while (nc < need) nc *= 2;
    Buf<U> nb;
    BT_CHECK(nb.alloc(ctx->pool, nc));
    if (used > 0)
        BT_HIP_CHECK(hipMemcpyAsync(nb.get(), buf.get(), (size_t) used * sizeof(U),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    buf.swap(nb);
    return BT_OK;
}

template <class T, int D>
int coll_l2_two_pass(bt_context *ctx, TravState *st, TravArgs<T, D> &a, Buf<int32_t> &l2_by_box,
                     Buf<int32_t> &l2_lists, int64_t *l2_total_out)
{
    constexpr int C = 1 << D;
    const bt_trav_params &p = st->p;
    const int64_t B = p.nboxes;
    const int nlevels = p.nlevels;
    const int32_t *ls = p.level_start_box_nrs;       // host
    CsrList &coll = st->coll;
    BT_CHECK(coll.starts.alloc(ctx->pool, B + 1));
    BT_CHECK(l2_by_box.alloc(ctx->pool, B + 1));
    BT_HIP_CHECK(hipMemsetAsync(coll.starts.get(), 0, (size_t) (B + 1) * 4, ctx->stream));
    BT_HIP_CHECK(hipMemsetAsync(l2_by_box.get(), 0, (size_t) (B + 1) * 4, ctx->stream));
    BT_CHECK(grow_buf(ctx, coll.lists, 0, std::max<int64_t>(B * 10, 1024)));
    BT_CHECK(grow_buf(ctx, l2_lists, 0, std::max<int64_t>(B * 40, 1024)));
    int64_t coll_total = 0, l2_total = 0;
    Buf<int32_t> totals_d;
    BT_CHECK(totals_d.alloc(ctx->pool, 2));
    for (int lev = 1; lev < nlevels; ++lev) {
        int32_t b0 = ls[lev], nb = ls[lev + 1] - ls[lev];
        if (p.active_level_ranges) {        // sharded traversal: this rank's boxes only
            b0 = p.active_level_ranges[2 * lev];
            nb = p.active_level_ranges[2 * lev + 1] - b0;
        }
        // keep both CSR start arrays monotone over the boxes outside the range
        auto fill_outside = [&](int32_t lo, int32_t hi, int64_t cv, int64_t lv) {
            if (hi <= lo) return;
            fill_i32_kernel<<<nblk(hi - lo), 256, 0, ctx->stream>>>(hi - lo, (int32_t) cv,
                                                                   coll.starts.get() + lo);
            fill_i32_kernel<<<nblk(hi - lo), 256, 0, ctx->stream>>>(hi - lo, (int32_t) lv,
                                                                   l2_by_box.get() + lo);
        };
        if (nb <= 0) {
            fill_outside(ls[lev], ls[lev + 1] + (lev == nlevels - 1 ? 1 : 0), coll_total, l2_total);
            continue;
        }
        fill_outside(ls[lev], b0, coll_total, l2_total);
        Buf<int32_t> ccnt, lcnt, crel, lrel;
        BT_CHECK(ccnt.alloc(ctx->pool, nb));
        BT_CHECK(lcnt.alloc(ctx->pool, nb));
        BT_CHECK(crel.alloc(ctx->pool, nb + 1));
        BT_CHECK(lrel.alloc(ctx->pool, nb + 1));
        a.coll_starts = coll.starts.get();
        a.coll_lists = coll.lists.get();
        CollL2Out oc{ccnt.get(), lcnt.get(), nullptr, nullptr};
        coll_l2_kernel<T, D, false><<<nblk((int64_t) nb * C), 256, 0, ctx->stream>>>(a, b0, nb, oc);
        int64_t h_tot[2] = {0, 0};
        BT_CHECK(scan_list_counts(ctx, ScanI32{ccnt.get()}, nb, crel.get(), &h_tot[0]));
        BT_CHECK(scan_list_counts(ctx, ScanI32{lcnt.get()}, nb, lrel.get(), &h_tot[1]));
        if (coll_total + h_tot[0] >= ((int64_t) 1 << 31) || l2_total + h_tot[1] >= ((int64_t) 1 << 31)) {
            set_error("interaction list exceeds 2^31-1 entries (int32 CSR limit of the reference)");
            return BT_ERR_UNSUPPORTED;
        }
        BT_CHECK(grow_buf(ctx, coll.lists, coll_total, coll_total + h_tot[0]));
        BT_CHECK(grow_buf(ctx, l2_lists, l2_total, l2_total + h_tot[1]));
        add_base_kernel<<<nblk(nb + 1), 256, 0, ctx->stream>>>(nb + 1, crel.get(), (int32_t) coll_total,
                                                              coll.starts.get() + b0);
        add_base_kernel<<<nblk(nb + 1), 256, 0, ctx->stream>>>(nb + 1, lrel.get(), (int32_t) l2_total,
                                                              l2_by_box.get() + b0);
        a.coll_lists = coll.lists.get();
        CollL2Out of{coll.starts.get(), l2_by_box.get(), coll.lists.get(), l2_lists.get()};
        coll_l2_kernel<T, D, true><<<nblk((int64_t) nb * C), 256, 0, ctx->stream>>>(a, b0, nb, of);
        coll_total += h_tot[0];
        l2_total += h_tot[1];
        fill_outside(b0 + nb + 1, ls[lev + 1] + (lev == nlevels - 1 ? 1 : 0), coll_total, l2_total);
    }
    coll.total = coll_total;
    a.coll_starts = coll.starts.get();
    a.coll_lists = coll.lists.get();
    *l2_total_out = l2_total;
    return BT_OK;
}

// colleagues in fixed-stride rows, list 2 via per-level scratch rows: every test once
template <class T, int D>
int coll_l2_single_pass(bt_context *ctx, TravState *st, TravArgs<T, D> &a, Buf<int32_t> &l2_by_box,
                        Buf<int32_t> &l2_lists, int64_t *l2_total_out, Buf<int32_t> &srccoll_rows,
                        Buf<int32_t> &srccoll_cnt)
{
    constexpr int C = 1 << D;
    constexpr int P = (D == 1 ? 3 : D == 2 ? 9 : 27) - 1;
    constexpr int S = (D == 1 ? 6 : D == 2 ? 36 : 216) - (P + 1);
    const bt_trav_params &p = st->p;
    const int64_t B = p.nboxes;
    const int nlevels = p.nlevels;
    const int32_t *ls = p.level_start_box_nrs;       // host
    CsrList &coll = st->coll;
    Buf<int32_t> coll_rows, coll_cnt;
    BT_CHECK(coll_rows.alloc(ctx->pool, B * P));
    BT_CHECK(coll_cnt.alloc(ctx->pool, B));
    BT_HIP_CHECK(hipMemsetAsync(coll_cnt.get(), 0, (size_t) B * 4, ctx->stream));
    BT_CHECK(srccoll_rows.alloc(ctx->pool, B * P));
    BT_CHECK(srccoll_cnt.alloc(ctx->pool, B));
    BT_HIP_CHECK(hipMemsetAsync(srccoll_cnt.get(), 0, (size_t) B * 4, ctx->stream));
    Buf<int32_t> coll_ins;
    BT_CHECK(coll_ins.alloc(ctx->pool, B));
    BT_HIP_CHECK(hipMemsetAsync(coll_ins.get(), 0, (size_t) B * 4, ctx->stream));
    BT_CHECK(l2_by_box.alloc(ctx->pool, B + 1));
    BT_HIP_CHECK(hipMemsetAsync(l2_by_box.get(), 0, (size_t) (B + 1) * 4, ctx->stream));
    st->l2_pending.clear();
    int64_t l2_total = 0;
    Buf<int32_t> total_d;
    BT_CHECK(total_d.alloc(ctx->pool, 1));
    for (int lev = 1; lev < nlevels; ++lev) {
        int32_t b0 = ls[lev], nb = ls[lev + 1] - ls[lev];
        if (p.active_level_ranges) {        // sharded traversal: this rank's boxes only
            b0 = p.active_level_ranges[2 * lev];
            nb = p.active_level_ranges[2 * lev + 1] - b0;
        }
        const int32_t lev_end = ls[lev + 1] + (lev == nlevels - 1 ? 1 : 0);
        auto fill_outside = [&](int32_t lo, int32_t hi, int64_t lv) {
            if (hi > lo)
                fill_i32_kernel<<<nblk(hi - lo), 256, 0, ctx->stream>>>(hi - lo, (int32_t) lv,
                                                                       l2_by_box.get() + lo);
        };
        if (nb <= 0) { fill_outside(ls[lev], lev_end, l2_total); continue; }
        fill_outside(ls[lev], b0, l2_total);
        Buf<int32_t> l2_rows, l2_cnt, l2_rel;
        BT_CHECK(l2_rows.alloc(ctx->pool, (int64_t) nb * S));
        BT_CHECK(l2_cnt.alloc(ctx->pool, nb));
        BT_CHECK(l2_rel.alloc(ctx->pool, nb + 1));
        coll_l2_rows_kernel<T, D><<<nblk((int64_t) nb * C), 256, 0, ctx->stream>>>(
            a, b0, nb, coll_rows.get(), coll_cnt.get(), l2_rows.get(), l2_cnt.get(),
            srccoll_rows.get(), srccoll_cnt.get(), coll_ins.get());
        int64_t h_tot = 0;
        BT_CHECK(scan_list_counts(ctx, ScanI32{l2_cnt.get()}, nb, l2_rel.get(), &h_tot));
        if (l2_total + h_tot >= ((int64_t) 1 << 31)) {
            set_error("interaction list exceeds 2^31-1 entries (int32 CSR limit of the reference)");
            return BT_ERR_UNSUPPORTED;
        }
        add_base_kernel<<<nblk(nb + 1), 256, 0, ctx->stream>>>(nb + 1, l2_rel.get(), (int32_t) l2_total,
                                                              l2_by_box.get() + b0);
        {
            L2Pending pend;
            pend.rows.swap(l2_rows);
            pend.rel.swap(l2_rel);
            pend.nb = nb;
            pend.base = l2_total;
            pend.stride = S;
            st->l2_pending.push_back(std::move(pend));
        }
        l2_total += h_tot;
        fill_outside(b0 + nb + 1, lev_end, l2_total);
    }
    // colleague CSR from the rows
    BT_CHECK(coll.starts.alloc(ctx->pool, B + 1));
    {
        int64_t t = 0;
        BT_CHECK(scan_list_counts(ctx, ScanI32{coll_cnt.get()}, B, coll.starts.get(), &t));
        coll.total = t;
        BT_CHECK(coll.lists.alloc(ctx->pool, t));
        compact_strided_rows_kernel<8><<<nblk(B * 8), 256, 0, ctx->stream>>>(
            B, P, coll_rows.get(), coll.starts.get(), 0, coll.lists.get());
    }
    BT_HIP_CHECK(hipGetLastError());
    *l2_total_out = l2_total;
    return BT_OK;
}

// colleagues + list 2 (top-down from the parent's colleagues) and list 1 (from the
// ancestors' colleagues), see bt_trav_fast.hpp
template <class T, int D>
int fast_lists(bt_context *ctx, TravState *st, TravArgs<T, D> &a)
{
    (void) sizeof(int[1 << D]);
    const bt_trav_params &p = st->p;
    const int64_t B = p.nboxes;
    const int nlevels = p.nlevels;
    const int32_t *ls = p.level_start_box_nrs;       // host
    const int walk_cap = nlevels + 1;
    const size_t walk_lds = (size_t) walk_cap * WALK_THREADS * 4;
    const size_t lvl_lds = (size_t) nlevels * WALK_THREADS * 4;

    // depth-first preorder ranks
    const int32_t *sizes = p.box_subtree_sizes;      // from bt_tree_export, or counted here
    BT_CHECK(st->dfs_rank.alloc(ctx->pool, B));
    BT_CHECK(st->box_of_rank.alloc(ctx->pool, B));
    if (!sizes) {
        BT_CHECK(st->subtree_size.alloc(ctx->pool, B));
        for (int lev = nlevels - 1; lev >= 0; --lev)
            subtree_size_kernel<D><<<nblk(ls[lev + 1] - ls[lev]), 256, 0, ctx->stream>>>(
                ls[lev], ls[lev + 1] - ls[lev], p.aligned_nboxes, p.box_child_ids,
                st->subtree_size.get());
        sizes = st->subtree_size.get();
    }
    for (int lev = 0; lev < nlevels; ++lev)
        dfs_rank_kernel<D><<<nblk(ls[lev + 1] - ls[lev]), 256, 0, ctx->stream>>>(
            ls[lev], ls[lev + 1] - ls[lev], p.aligned_nboxes, p.box_child_ids,
            sizes, st->dfs_rank.get(), st->box_of_rank.get());
    // source boxes in depth-first order (+ prefix counts over ranks)
    BT_CHECK(st->src_rank_prefix.alloc(ctx->pool, B + 1));
    {
        SourceRankFlag<T, D> f{a.nodes, st->box_of_rank.get()};
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, f, B, st->src_rank_prefix.get(),
                                                          (int32_t *) nullptr, true)));
        // with a source_boxes_mask (local trees of the distributed FMM) st->nsb counts
        // the masked list only; the lists here refer to every box flagged as a source
        BT_CHECK(st->src_by_rank.alloc(ctx->pool, (p.source_boxes_mask ? B : st->nsb) + 1));
        compact_sources_by_rank_kernel<T, D><<<nblk(B), 256, 0, ctx->stream>>>(
            f, (int32_t) B, st->src_rank_prefix.get(), st->src_by_rank.get());
    }
    a.dfs_rank = st->dfs_rank.get();
    FastTree ft{st->dfs_rank.get(), st->box_of_rank.get(), sizes,
                st->src_rank_prefix.get(), st->src_by_rank.get()};

    // colleagues + list 2, level by level
    CsrList &coll = st->coll;
    coll.n = B;
    Buf<int32_t> l2_by_box;
    Buf<int32_t> l2_lists;
    int64_t l2_total = 0;
    Buf<int32_t> &srccoll_rows = st->srccoll_rows, &srccoll_cnt = st->srccoll_cnt;
    const bool single_pass = p.well_sep_is_n_away == 1;      // (else: the count + fill kernels)
    if (single_pass) {
        BT_CHECK((coll_l2_single_pass<T, D>(ctx, st, a, l2_by_box, l2_lists, &l2_total,
                                            srccoll_rows, srccoll_cnt)));
    } else {
        BT_CHECK((coll_l2_two_pass<T, D>(ctx, st, a, l2_by_box, l2_lists, &l2_total)));
    }
    a.coll_starts = coll.starts.get();
    a.coll_lists = coll.lists.get();
    BT_CHECK(tmark(ctx, st, "trav:colleagues+list2"));

    // list 2 is indexed by position in target_or_target_parent_boxes
    {
        CsrList &c = st->l2;
        c.n = st->nttp;
        c.total = l2_total;
        BT_CHECK(c.starts.alloc(ctx->pool, c.n + 1));
        gather_starts_kernel<<<nblk(c.n + 1), 256, 0, ctx->stream>>>(
            (int32_t) c.n, st->ttp_boxes.get(), l2_by_box.get(), (int32_t) l2_total, c.starts.get());
        c.lists.swap(l2_lists);
    }

    // source-box colleagues: what the ancestors contribute to list 1.  The single-pass
    // colleague kernel already produced them as fixed-stride rows.
    Buf<int32_t> lcoll_starts_buf, lcoll_lists_buf;
    const int32_t *lcoll_starts = nullptr, *lcoll_lists = nullptr;
    constexpr int COLL_STRIDE = (D == 1 ? 3 : D == 2 ? 9 : 27) - 1;
    int lcoll_stride = 0;
    if (single_pass) {
        lcoll_starts = srccoll_cnt.get();
        lcoll_lists = srccoll_rows.get();
        lcoll_stride = COLL_STRIDE;
        a.srccoll_rows = srccoll_rows.get();       // list 4 walks the same rows
        a.srccoll_cnt = srccoll_cnt.get();
        a.srccoll_stride = COLL_STRIDE;
        a.srccoll_id_mask = ~0u;
        a.srccoll_src_bit = 0;
    } else {
        BT_CHECK(lcoll_starts_buf.alloc(ctx->pool, B + 1));
        filter_source_colleagues_kernel<T, D, false><<<nblk(B), 256, 0, ctx->stream>>>(
            a, (int32_t) B, lcoll_starts_buf.get(), nullptr);
        int64_t tot = 0;
        BT_CHECK(counts_to_starts(ctx, lcoll_starts_buf, B, &tot));
        BT_CHECK(lcoll_lists_buf.alloc(ctx->pool, tot));
        filter_source_colleagues_kernel<T, D, true><<<nblk(B), 256, 0, ctx->stream>>>(
            a, (int32_t) B, lcoll_starts_buf.get(), lcoll_lists_buf.get());
        lcoll_starts = lcoll_starts_buf.get();
        lcoll_lists = lcoll_lists_buf.get();
    }

    // lists 1 and 3 (+ close smaller) in one walk per work item
    {
        const int64_t ntb = st->ntb;
        // work items: heavy target boxes (far above the leaf level) get one item per
        // colleague, see make_items_kernel
        constexpr int heavy_margin = 3;
        const int heavy_max_level = nlevels - heavy_margin;
        Buf<int32_t> first_item, item_tbn, item_slot;
        BT_CHECK(first_item.alloc(ctx->pool, ntb + 1));
        make_items_kernel<T, D, false><<<nblk(ntb), 256, 0, ctx->stream>>>(
            a, (int32_t) ntb, heavy_max_level, first_item.get(), nullptr, nullptr);
        int64_t nitems = 0;
        BT_CHECK(counts_to_starts(ctx, first_item, ntb, &nitems));
        BT_CHECK(item_tbn.alloc(ctx->pool, nitems));
        BT_CHECK(item_slot.alloc(ctx->pool, nitems));
        make_items_kernel<T, D, true><<<nblk(ntb), 256, 0, ctx->stream>>>(
            a, (int32_t) ntb, heavy_max_level, first_item.get(), item_tbn.get(), item_slot.get());

        const int64_t nflat = (int64_t) nlevels * nitems;
        if (nflat >= ((int64_t) 1 << 31)) {
            set_error("list 3 bookkeeping exceeds int32 range");
            return BT_ERR_UNSUPPORTED;
        }
        Buf<int32_t> l1_item, l3_item, close_item;
        BT_CHECK(l1_item.alloc(ctx->pool, nitems + 1));
        BT_CHECK(l3_item.alloc(ctx->pool, nflat + 1));
        if (st->with_extent) BT_CHECK(close_item.alloc(ctx->pool, nitems + 1));
        list13_kernel<T, D, false><<<nblk(nitems), 256, walk_lds + lvl_lds, ctx->stream>>>(
            a, ft, lcoll_starts, lcoll_lists, item_tbn.get(), item_slot.get(),
            (int32_t) nitems, nlevels, walk_cap, l1_item.get(), nullptr, l3_item.get(), nullptr,
            st->with_extent ? close_item.get() : nullptr, nullptr, lcoll_stride);

        CsrList &c1 = st->l1;
        c1.n = ntb;
        CsrList &cs = st->close_smaller;
        cs.n = ntb;
        int64_t total3 = 0;
        BT_CHECK(counts_to_starts(ctx, l1_item, nitems, &c1.total));
        BT_CHECK(counts_to_starts(ctx, l3_item, nflat, &total3));
        BT_CHECK(c1.lists.alloc(ctx->pool, c1.total));
        BT_CHECK(st->l3_lists.alloc(ctx->pool, total3));
        if (st->with_extent) {
            BT_CHECK(counts_to_starts(ctx, close_item, nitems, &cs.total));
            BT_CHECK(cs.lists.alloc(ctx->pool, cs.total));
        }
        list13_kernel<T, D, true><<<nblk(nitems), 256, walk_lds + lvl_lds, ctx->stream>>>(
            a, ft, lcoll_starts, lcoll_lists, item_tbn.get(), item_slot.get(),
            (int32_t) nitems, nlevels, walk_cap, l1_item.get(), c1.lists.get(), l3_item.get(),
            st->l3_lists.get(), st->with_extent ? close_item.get() : nullptr,
            st->with_extent ? cs.lists.get() : nullptr, lcoll_stride);

        // per-box starts from the per-item starts (items of a box are consecutive)
        BT_CHECK(c1.starts.alloc(ctx->pool, ntb + 1));
        gather_starts_kernel<<<nblk(ntb + 1), 256, 0, ctx->stream>>>(
            (int32_t) ntb, first_item.get(), l1_item.get(), (int32_t) c1.total, c1.starts.get());
        if (st->with_extent) {
            BT_CHECK(cs.starts.alloc(ctx->pool, ntb + 1));
            gather_starts_kernel<<<nblk(ntb + 1), 256, 0, ctx->stream>>>(
                (int32_t) ntb, first_item.get(), close_item.get(), (int32_t) cs.total,
                cs.starts.get());
        }
        const int64_t nflat_box = (int64_t) nlevels * ntb;
        BT_CHECK(st->l3_starts.alloc(ctx->pool, nflat_box + 1));
        l3_box_starts_kernel<<<nblk(nflat_box + 1), 256, 0, ctx->stream>>>(
            nflat_box, (int32_t) ntb, (int32_t) nitems, nlevels, first_item.get(), l3_item.get(),
            st->l3_starts.get());

        // list 1: order by depth-first rank, insert the own-subtree blocks
        Buf<int32_t> jobbuf;
        BT_CHECK(jobbuf.alloc(ctx->pool, 3 * ntb + 1));
        BlockJobs jobs{jobbuf.get(), jobbuf.get() + 1, jobbuf.get() + 1 + ntb,
                       jobbuf.get() + 1 + 2 * ntb, nullptr};
        BT_HIP_CHECK(hipMemsetAsync(jobs.len, 0, (size_t) ntb * 4, ctx->stream));
        Buf<uint8_t> tier;
        Buf<int32_t> tier_present;
        BT_CHECK(tier.alloc(ctx->pool, ntb));
        BT_CHECK(tier_present.alloc(ctx->pool, 2));
        BT_HIP_CHECK(hipMemsetAsync(tier.get(), 0, (size_t) ntb, ctx->stream));
        BT_HIP_CHECK(hipMemsetAsync(tier_present.get(), 0, 8, ctx->stream));
        l1_finalize32_kernel<T, D><<<nblk(ntb * 16), 256, 0, ctx->stream>>>(
            a, ft, (int32_t) ntb, c1.starts.get(), c1.lists.get(), jobs, tier.get(),
            tier_present.get(), 0);
        int32_t h_present[2] = {0, 0};
        BT_CHECK(bt::d2h(ctx, h_present, tier_present.get(), 8));
        BT_CHECK(bt::sync_stream(ctx));
        for (int which = 1; which <= 2; ++which) {
            if (!h_present[which - 1]) continue;
            TierIs pr{tier.get(), (uint8_t) which};
            Buf<int32_t> pos, list;
            BT_CHECK(pos.alloc(ctx->pool, ntb + 1));
            BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, pr, ntb, pos.get(),
                                                              (int32_t *) nullptr, true)));
            int32_t cnt = 0;
            BT_CHECK(read_i32(ctx, pos.get() + ntb, &cnt));
            if (cnt == 0) continue;
            BT_CHECK(list.alloc(ctx->pool, cnt));
            compact_tier_kernel<<<nblk(ntb), 256, 0, ctx->stream>>>((int32_t) ntb, pr, pos.get(),
                                                                   list.get());
            if (which == 1)
                l1_finalize_wave_kernel<T, D><<<(unsigned) div_up(cnt, 4), 256, 0, ctx->stream>>>(
                    a, ft, list.get(), pos.get() + ntb, c1.starts.get(), c1.lists.get(), jobs, 0);
            else
                l1_finalize_block_kernel<T, D><<<cnt, 256, 0, ctx->stream>>>(
                    a, ft, list.get(), pos.get() + ntb, c1.starts.get(), c1.lists.get(), jobs, 0);
            BT_CHECK(bt::sync_stream(ctx));    // `list` is freed on scope exit
        }
        if (st->has_blocks)
            copy_rank_blocks_kernel<<<(unsigned) std::min<int64_t>(ctx->num_cus * 8, std::max<int64_t>(1, div_up(ntb, 4))), 256, 0,
                                      ctx->stream>>>(
                (int32_t) ntb, jobs.dst, jobs.src, jobs.len, st->src_by_rank.get(), c1.lists.get());
    }
    BT_CHECK(tmark(ctx, st, "trav:list1+list3"));
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

// ---- one-block output ---------------------------------------------------------------------

void fill_sizes(TravState *st, bt_trav_sizes *out)
{
    const int nlevels = st->nlevels;
    memset(out, 0, sizeof(*out));
    out->nsource_boxes = st->nsb; out->ntarget_boxes = st->ntb;
    out->nsource_parent_boxes = st->nspb;
    out->ntarget_or_target_parent_boxes = st->nttp;
    out->n_same_level_non_well_sep = st->coll.total;
    out->n_neighbor_source = st->l1.total;
    out->n_from_sep_siblings = st->l2.total;
    out->n_from_sep_bigger = st->l4.total;
    out->n_from_sep_close_smaller = st->with_extent ? st->close_smaller.total : -1;
    out->n_from_sep_close_bigger = st->with_extent ? st->close_bigger.total : -1;
    for (int l = 0; l < nlevels; ++l) {
        out->n_from_sep_smaller[l] = st->l3_level_count[l];
        out->n_from_sep_smaller_nonempty[l] = st->l3_nonempty[l];
    }
}

// spans of all outputs in the caller's block (the list-3 lists of the levels are
// consecutive, like st->l3_lists); asks the caller for the block
int make_arena(bt_context *ctx, TravState *st)
{
    bt_trav_packed *pk = st->packed;
    const bt_trav_sizes &z = st->sizes;
    const int nl = st->nlevels;
    const int64_t B = st->p.nboxes;
    int64_t off = 0;
    auto take = [&](bt_span &sp, int64_t count) {
        sp.offset = off; sp.count = count;
        off += (std::max<int64_t>(count, 0) + 63) / 64 * 64;      // 256-byte steps
    };
    const bool shared_tb = st->p.sources_are_targets && !st->p.target_boxes_mask;
    take(pk->source_boxes, z.nsource_boxes);
    if (shared_tb) pk->target_boxes = pk->source_boxes;
    else take(pk->target_boxes, z.ntarget_boxes);
    take(pk->source_parent_boxes, z.nsource_parent_boxes);
    take(pk->target_or_target_parent_boxes, z.ntarget_or_target_parent_boxes);
    take(pk->same_level_non_well_sep_boxes_starts, B + 1);
    take(pk->same_level_non_well_sep_boxes_lists, z.n_same_level_non_well_sep);
    take(pk->neighbor_source_boxes_starts, z.ntarget_boxes + 1);
    take(pk->neighbor_source_boxes_lists, z.n_neighbor_source);
    take(pk->from_sep_siblings_starts, z.ntarget_or_target_parent_boxes + 1);
    take(pk->from_sep_siblings_lists, z.n_from_sep_siblings);
    take(pk->from_sep_bigger_starts, z.ntarget_or_target_parent_boxes + 1);
    take(pk->from_sep_bigger_lists, z.n_from_sep_bigger);
    if (st->with_extent) {
        take(pk->from_sep_close_smaller_starts, z.ntarget_boxes + 1);
        take(pk->from_sep_close_smaller_lists, z.n_from_sep_close_smaller);
        take(pk->from_sep_close_bigger_starts, z.ntarget_boxes + 1);
        take(pk->from_sep_close_bigger_lists, z.n_from_sep_close_bigger);
    } else {
        pk->from_sep_close_smaller_starts = pk->from_sep_close_smaller_lists = bt_span{0, -1};
        pk->from_sep_close_bigger_starts = pk->from_sep_close_bigger_lists = bt_span{0, -1};
    }
    // list 3: all levels' lists back to back, then the small per-level arrays
    for (int l = 0; l < nl; ++l) {
        pk->from_sep_smaller_lists[l].offset = off;
        pk->from_sep_smaller_lists[l].count = z.n_from_sep_smaller[l];
        off += z.n_from_sep_smaller[l];
    }
    off = (off + 63) / 64 * 64;
    for (int l = 0; l < nl; ++l) {
        take(pk->from_sep_smaller_starts[l], z.n_from_sep_smaller_nonempty[l] + 1);
        take(pk->from_sep_smaller_nonempty_indices[l], z.n_from_sep_smaller_nonempty[l]);
        take(pk->from_sep_smaller_compressed_indices[l], z.ntarget_boxes + 1);
        take(pk->target_boxes_sep_smaller[l], z.n_from_sep_smaller_nonempty[l]);
    }
    pk->total = off;
    pk->nlevels = nl;
    pk->sizes = z;
    pk->lattice_path = st->lattice ? 1 : 0;
    const int32_t *h = st->h_lev_starts.data();
    for (int l = 0; l <= nl; ++l) {
        pk->level_start_source_box_nrs[l] = h[0 * (nl + 1) + l];
        pk->level_start_target_box_nrs[l] = h[1 * (nl + 1) + l];
        pk->level_start_source_parent_box_nrs[l] = h[2 * (nl + 1) + l];
        pk->level_start_target_or_target_parent_box_nrs[l] = h[3 * (nl + 1) + l];
    }
    void *base = st->packed_alloc(st->packed_user, std::max<int64_t>(off, 64) * 4);
    if (!base) {
        set_error("bt_traversal_build_packed: the allocation callback returned NULL "
                  "(%lld bytes)", (long long) off * 4);
        return BT_ERR_ALLOC;
    }
    pk->base = base;
    st->arena = (int32_t *) base;
    return BT_OK;
}

inline int32_t *span_ptr(TravState *st, const bt_span &sp)
{
    return (st->arena && sp.count >= 0) ? st->arena + sp.offset : nullptr;
}

// a list's final storage: the span in the caller's block if there is one, the pool else
int place_list(bt_context *ctx, TravState *st, Buf<int32_t> &buf, int64_t count, const bt_span *sp)
{
    if (st->arena && sp) { buf.set_external(st->arena + sp->offset, count); return BT_OK; }
    return buf.alloc(ctx->pool, count);
}

// ---- lattice kernels (bt_trav_v2.hpp): colleagues, lists 1-3, close-smaller -------------
//
// Everything is counted first (rows with exact counts), the totals of all lists come
// back in ONE host synchronisation, then every list is written to its final place.

__global__ __launch_bounds__(256) void gather_i32_kernel(int32_t n, const int32_t *idx,
                                                         const int32_t *src, int32_t *dst)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// out[i] = starts_by_box[boxes[i]] (i < n), out[n] = starts_by_box[last]
__global__ __launch_bounds__(256) void gather_starts_dev_kernel(int32_t n, const int32_t *boxes,
        const int32_t *starts_by_box, int32_t last, int32_t *out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = starts_by_box[boxes[i]];
    if (i == n) out[n] = starts_by_box[last];
}

template <class T, int D>
int fast_lists_v2(bt_context *ctx, TravState *st, TravArgs<T, D> &a)
{
    constexpr int P = V2Dims<D>::P;
    const bt_trav_params &p = st->p;
    const int64_t B = p.nboxes;
    const int nlevels = p.nlevels;
    const int32_t *ls = p.level_start_box_nrs;       // host
    const int walk_cap = nlevels + 1;
    const size_t walk_lds = (size_t) walk_cap * WALK_THREADS * 4;
    const size_t lvl_lds = (size_t) nlevels * WALK_THREADS * 4;
    const int64_t ntb = st->ntb;
    const bool with_blocks = st->has_blocks;

    // ---- per-tree tables and colleague rows, level by level ---------------------------------
    const int32_t *sizes = p.box_subtree_sizes;      // from bt_tree_export, or counted here
    BT_CHECK(st->dfs_rank.alloc(ctx->pool, B));
    BT_CHECK(st->box_of_rank.alloc(ctx->pool, B));
    BT_CHECK(st->cells.alloc(ctx->pool, B * (int64_t) sizeof(ICell)));
    ICell *cells = (ICell *) st->cells.get();
    if (!sizes) {
        BT_CHECK(st->subtree_size.alloc(ctx->pool, B));
        for (int lev = nlevels - 1; lev >= 0; --lev)
            subtree_size_kernel<D><<<nblk(ls[lev + 1] - ls[lev]), 256, 0, ctx->stream>>>(
                ls[lev], ls[lev + 1] - ls[lev], p.aligned_nboxes, p.box_child_ids,
                st->subtree_size.get());
        sizes = st->subtree_size.get();
    }
    // One family of colleague rows with the source flag in the entry, and a mask per box of the
    // entries that carry it, where box numbers leave a bit for the flag (3D: every BASELINE
    // configuration; BT_ROW_FAMILIES=2 forces the other form), else all colleagues in one
    // family and the source boxes among them in a second one.
    // (read per call: a test builds the same lists in both forms)
    const bool two_env = [] { const char *e = getenv("BT_ROW_FAMILIES"); return e && atoi(e) == 2; }();
    const bool one_family = D == 3 && B < V2_ONE_FAMILY_MAX_BOXES && !two_env;
    const uint32_t row_id_mask = one_family ? (V2_SRC_BIT - 1u) : V2_ID_MASK;
    const uint32_t row_src_bit = one_family ? V2_SRC_BIT : 0u;
    Buf<int32_t> coll_rows_two, coll_cnt, coll_ins, l2_cnt;
    Buf<int32_t> &srccoll_rows = st->srccoll_rows, &srccoll_cnt = st->srccoll_cnt;
    if (!one_family) BT_CHECK(coll_rows_two.alloc(ctx->pool, B * P));
    BT_CHECK(srccoll_rows.alloc(ctx->pool, B * P));
    // (one family: it lives where the source rows of two families do -- list 4 reads it last)
    Buf<int32_t> &coll_rows = one_family ? srccoll_rows : coll_rows_two;
    // coll_cnt | coll_ins | l2_cnt | srccoll_cnt are one array, zeroed together (the last
    // quarter outlives this function: list 4 reads it)
    BT_CHECK(srccoll_cnt.alloc(ctx->pool, 4 * B));
    BT_HIP_CHECK(hipMemsetAsync(srccoll_cnt.get(), 0, (size_t) (4 * B) * 4, ctx->stream));
    int32_t *d_coll_cnt = srccoll_cnt.get() + B, *d_coll_ins = srccoll_cnt.get() + 2 * B,
            *d_l2_cnt = srccoll_cnt.get() + 3 * B;
    V2Rows<D> rows{};
    rows.child8 = st->child8.get();
    rows.parent = p.box_parent_ids;
    rows.flags = p.box_flags;
    rows.target_mask = p.target_boxes_mask;
    rows.coll_rows = coll_rows.get(); rows.coll_cnt = d_coll_cnt; rows.coll_ins = d_coll_ins;
    rows.srccoll_rows = one_family ? nullptr : srccoll_rows.get();
    rows.srccoll_cnt = srccoll_cnt.get();
    rows.id_mask = row_id_mask;
    rows.l2_cnt = d_l2_cnt;
    // what list 4 and the coarse part of list 1 read: the source colleagues of a box
    const int32_t *src_rows = coll_rows.get();
    const int32_t *src_cnt = srccoll_cnt.get();
    if (!one_family) src_rows = srccoll_rows.get();
    // Level lev: its boxes hand depth-first ranks and cells to their children, and the rows
    // of level lev + 1 are built from those of level lev -- one launch for both
    // (level_tables_kernel).  The last level has no children: nothing to do there.
    for (int lev = 0; lev < nlevels; ++lev) {
        if (lev > 0 && lev == nlevels - 1) break;
        DfsLevel dl{ls[lev], ls[lev + 1] - ls[lev], p.aligned_nboxes, p.box_child_ids, sizes,
                    p.box_levels, p.box_flags, st->dfs_rank.get(), st->box_of_rank.get(), cells};
        const int32_t dfs_blocks = (int32_t) nblk(dl.nb);
        int32_t np = 0, b0 = 0, nb = 0;
        const int32_t *pls = st->h_lev_starts.data() + 4 * (nlevels + 1);
        if (lev + 1 < nlevels) {
            b0 = ls[lev + 1]; nb = ls[lev + 2] - ls[lev + 1];
            if (p.active_level_ranges) {        // sharded traversal: this rank's boxes only
                b0 = p.active_level_ranges[2 * (lev + 1)];
                nb = p.active_level_ranges[2 * (lev + 1) + 1] - b0;
            }
            // a group of lanes per box of this level that has children
            np = pls[lev + 1] - pls[lev];
        }
        if (nb > 0 && np > 0 && one_family)
            level_tables_kernel<D, true><<<dfs_blocks + nblk((int64_t) np * V3Lanes<D>::N), 256, 0, ctx->stream>>>(
                dl, dfs_blocks, rows, st->parent_boxes.get() + pls[lev], np, b0, b0 + nb);
        else if (nb > 0 && np > 0)
            level_tables_kernel<D, false><<<dfs_blocks + nblk((int64_t) np * V3Lanes<D>::N), 256, 0, ctx->stream>>>(
                dl, dfs_blocks, rows, st->parent_boxes.get() + pls[lev], np, b0, b0 + nb);
        else
            dfs_rank_cells_kernel<D><<<dfs_blocks, 256, 0, ctx->stream>>>(dl);
    }
    if (with_blocks) {
        // source boxes in depth-first order (+ prefix counts over ranks): own-subtree blocks
        BT_CHECK(st->src_rank_prefix.alloc(ctx->pool, B + 1));
        SourceRankFlag<T, D> f{a.nodes, st->box_of_rank.get()};
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, f, B, st->src_rank_prefix.get(),
                                                          (int32_t *) nullptr, true)));
        BT_CHECK(st->src_by_rank.alloc(ctx->pool, B + 1));
        compact_sources_by_rank_kernel<T, D><<<nblk(B), 256, 0, ctx->stream>>>(
            f, (int32_t) B, st->src_rank_prefix.get(), st->src_by_rank.get());
    }
    a.dfs_rank = st->dfs_rank.get();
    FastTree ft{st->dfs_rank.get(), st->box_of_rank.get(), sizes,
                st->src_rank_prefix.get(), st->src_by_rank.get()};
    a.srccoll_rows = src_rows;                 // list 4 walks the same rows
    a.srccoll_cnt = src_cnt;
    a.srccoll_stride = P;
    a.srccoll_id_mask = row_id_mask;
    a.srccoll_src_bit = row_src_bit;
    BT_CHECK(tmark(ctx, st, "trav:colleague rows"));

    // ---- work items -----------------------------------------------------------------------
    constexpr int heavy_margin = 3;
    const int heavy_max_level = nlevels - heavy_margin;
    // columns: the target boxes of level tl make at most cap(tl) items
    L3Layout lay{};
    int64_t items_cap = 0, nflat = 0;
    {
        const int32_t *tls = st->h_lev_starts.data() + (nlevels + 1);     // target boxes per level
        for (int l = 0; l <= nlevels; ++l) {
            lay.ecap[l] = (int32_t) std::min<int64_t>(items_cap, INT_MAX);
            lay.base[l] = (int32_t) std::min<int64_t>(nflat, INT_MAX);
            nflat += items_cap;
            if (l < nlevels)
                items_cap += (int64_t) (tls[l + 1] - tls[l]) * (l <= heavy_max_level ? P + 1 : 1);
        }
        nflat -= items_cap;            // rows 0 .. nlevels-1 only
        lay.base[nlevels] = (int32_t) std::min<int64_t>(nflat, INT_MAX);
    }
    items_cap = std::max<int64_t>(64, div_up(items_cap, 64) * 64);
    if (nflat >= ((int64_t) 1 << 31) || items_cap >= ((int64_t) 1 << 31)) {
        set_error("list 3 bookkeeping exceeds int32 range");
        return BT_ERR_UNSUPPORTED;
    }
    Buf<int32_t> item_cnt, first_item, item_tbn, item_slot;
    Buf<int64_t> totals;            // device: nitems, coll, l2, l1, l3, close
    enum { T_NITEMS = 0, T_COLL, T_L2, T_L1, T_L3, T_CLOSE, T_L4, T_L4RAW, T_OVF, T_COUNT };
    if (int64_t *z = (int64_t *) bt::zero_alloc(ctx, T_COUNT * 8)) {
        totals.set_external(z, T_COUNT);
    } else {
        BT_CHECK(totals.alloc(ctx->pool, T_COUNT));
        BT_HIP_CHECK(hipMemsetAsync(totals.get(), 0, T_COUNT * 8, ctx->stream));
    }
    BT_CHECK(item_cnt.alloc(ctx->pool, ntb));
    BT_CHECK(first_item.alloc(ctx->pool, ntb + 1));
    BT_CHECK(item_tbn.alloc(ctx->pool, items_cap));
    BT_CHECK(item_slot.alloc(ctx->pool, items_cap));
    make_items_v2_kernel<false><<<nblk(ntb), 256, 0, ctx->stream>>>(
        (int32_t) ntb, st->target_boxes, cells, d_coll_cnt, heavy_max_level, item_cnt.get(),
        nullptr, nullptr);
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, ScanI32{item_cnt.get()}, ntb,
                                                      first_item.get(), totals.get() + T_NITEMS, true)));
    make_items_v2_kernel<true><<<nblk(ntb), 256, 0, ctx->stream>>>(
        (int32_t) ntb, st->target_boxes, cells, d_coll_cnt, heavy_max_level, first_item.get(),
        item_tbn.get(), item_slot.get());
    // the item count as int32 for the kernels: the low word of the little-endian total
    const int32_t *d_nitems = (const int32_t *) (totals.get() + T_NITEMS);

    // ---- the walk: lists 1 and 3 (+ close) into rows ------------------------------------------
    // (read per call: tests shrink the rows to send every item through the second walk)
    const int k1_env = [] { const char *e = getenv("BT_V2_K1"); return e ? atoi(e) : 0; }();
    const int k3_env = [] { const char *e = getenv("BT_V2_K3"); return e ? atoi(e) : 0; }();
    // (rows written in groups of four entries -- the walk without target extents, RowGroup --
    // hold whole groups)
    const int row_group = a.targets_have_extent ? 1 : 4;
    auto whole_groups = [&](int k) { return (k + row_group - 1) / row_group * row_group; };
    const int K1 = whole_groups(k1_env > 0 ? k1_env : (D == 3 ? 64 : D == 2 ? 24 : 8));
    // (with extents the lists 3 of the per-colleague items are long: at 10^8 + 10^7
    // particles 8 % of the items overflow 32 entries, 1 % overflow 64.  Without extents a
    // volume-filling cloud is what fills them: a leaf beside refined neighbours takes the
    // non-adjacent children of up to 26 of them -- at 1.25*10^8 uniform points 17.5 % of the
    // items overflow 32 entries and their second walk costs 1.7 ms; with 64 it is < 0.1 ms,
    // the first walk 0.3 ms longer, 96 and 128 change nothing more)
    const int K3 = whole_groups(k3_env > 0 ? k3_env : (D == 3 ? 64 : D == 2 ? 24 : 8));
    const int Kc = st->with_extent ? K3 : 0;
    Buf<int32_t> row1, row3, rowc, l1_item, l3_item, close_item, ovf_list;
    Buf<uint8_t> overflow;
    BT_CHECK(row1.alloc(ctx->pool, items_cap * K1));
    BT_CHECK(row3.alloc(ctx->pool, items_cap * K3));
    if (Kc) BT_CHECK(rowc.alloc(ctx->pool, items_cap * Kc));
    BT_CHECK(l1_item.alloc(ctx->pool, items_cap + 1));
    BT_CHECK(l3_item.alloc(ctx->pool, nflat + 1));
    if (st->with_extent) BT_CHECK(close_item.alloc(ctx->pool, items_cap + 1));
    (void) sizeof(nflat);
    BT_CHECK(overflow.alloc(ctx->pool, items_cap));
    BT_CHECK(ovf_list.alloc(ctx->pool, items_cap));
    // spill chunks for the items whose list 3 outgrows its row (see V2Walk): one chunk per 64
    // items at most -- beyond that the rows are simply too short for the tree
    const bool spill_env = [] { const char *e = getenv("BT_V2_SPILL"); return !e || atoi(e); }();
    Buf<int32_t> spill3, spill_count, spill_idx;
    const int32_t spill_per_shard = (int32_t) std::max<int64_t>(16, items_cap / 64 / SPILL_SHARDS);
    if (spill_env && !st->with_extent) {     // (extent trees: own-subtree blocks, close lists, registers)
        const int64_t nch = (int64_t) spill_per_shard * SPILL_SHARDS;
        BT_CHECK(spill3.alloc(ctx->pool, nch * SPILL_CHUNK));
        BT_CHECK(spill_idx.alloc(ctx->pool, 2 * items_cap));
        BT_CHECK(spill_count.alloc(ctx->pool, SPILL_SHARDS * 16));
        BT_HIP_CHECK(hipMemsetAsync(spill_count.get(), 0, SPILL_SHARDS * 16 * 4, ctx->stream));
    }
    Buf<int32_t> l1_cnt, l3_cnt, close_cnt;
    BT_CHECK(l1_cnt.alloc(ctx->pool, items_cap));
    BT_CHECK(l3_cnt.alloc(ctx->pool, nflat));
    if (st->with_extent) BT_CHECK(close_cnt.alloc(ctx->pool, items_cap));

    V2Walk w{};
    w.cells = cells;
    w.flags = p.box_flags;
    w.child8 = st->child8.get();
    w.coll_rows = coll_rows.get(); w.coll_cnt = d_coll_cnt;
    w.srccoll_rows = src_rows; w.srccoll_cnt = src_cnt;
    w.id_mask = row_id_mask; w.src_bit = row_src_bit;
    w.item_tbn = item_tbn.get(); w.item_slot = item_slot.get();
    w.d_nitems = d_nitems;
    w.items_cap = (int32_t) items_cap;
    w.lay = lay;
    w.nlevels = nlevels; w.walk_cap = walk_cap;
    w.with_blocks = with_blocks ? 1 : 0;
    w.row1 = row1.get(); w.row3 = row3.get(); w.rowc = Kc ? rowc.get() : nullptr;
    w.K1 = K1; w.K3 = K3; w.Kc = Kc;
    w.l1_cs = l1_cnt.get(); w.l3_cs = l3_cnt.get();
    w.close_cs = st->with_extent ? close_cnt.get() : nullptr;
    w.overflow = overflow.get();
    w.spill3 = spill3.get(); w.spill_count = spill_count.get();
    w.spill_idx = spill_idx.get(); w.spill_per_shard = spill_per_shard;
    w.ovf_count = (int32_t *) (totals.get() + T_OVF);
    w.ovf_list = ovf_list.get();
    static const bool trav_stats = [] { const char *e = getenv("BT_TRAV_STATS"); return e && atoi(e); }();
    // (an experiment, off by default: see walk13_v2_kernel)
    static const bool walk_two_pass = [] { const char *e = getenv("BT_WALK_TWO_PASS"); return e && atoi(e); }();
    Buf<int32_t> dbg_counts;
    if (trav_stats) {
        BT_CHECK(dbg_counts.alloc(ctx->pool, 16));
        BT_HIP_CHECK(hipMemsetAsync(dbg_counts.get(), 0, 64, ctx->stream));
        w.dbg_counts = dbg_counts.get();
    }
    // (with target extents the centre of the box being scanned has an LDS column as well)
    const size_t cen_lds = (size_t) D * sizeof(T) * WALK_THREADS;
    // (an experiment, off by default: 2^d lanes per item, see walk13_g8_kernel)
    // (1: rows as the one-lane walk lays them out; 2: item-major rows, which the row readers follow)
    static const int walk_g8 = [] { const char *e = getenv("BT_WALK_G8"); return e ? atoi(e) : 0; }();
    if (a.targets_have_extent) {
        if (walk_g8 == 2) walk13_g8_kernel<T, D, true, true><<<(unsigned) div_up(items_cap, WALK_THREADS >> D), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, w);
        else if (walk_g8) walk13_g8_kernel<T, D, true, false><<<(unsigned) div_up(items_cap, WALK_THREADS >> D), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, w);
        else if (walk_two_pass) walk13_v2_kernel<T, D, true, true, true><<<nblk(items_cap), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, w);
        else walk13_v2_kernel<T, D, true, true><<<nblk(items_cap), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, w);
    } else {
        if (walk_g8 == 2) walk13_g8_kernel<T, D, false, true><<<(unsigned) div_up(items_cap, WALK_THREADS >> D), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, w);
        else if (walk_g8) walk13_g8_kernel<T, D, false, false><<<(unsigned) div_up(items_cap, WALK_THREADS >> D), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, w);
        else if (walk_two_pass) walk13_v2_kernel<T, D, true, false, true><<<nblk(items_cap), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, w);
        else walk13_v2_kernel<T, D, true, false><<<nblk(items_cap), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, w);
    }
    BT_CHECK(tmark(ctx, st, "trav:walk (rows)"));

    // ---- starts of everything, totals in one transfer -------------------------------------------
    CsrList &coll = st->coll;
    coll.n = B;
    Buf<int32_t> l2_by_box;
    BT_CHECK(coll.starts.alloc(ctx->pool, B + 1));
    BT_CHECK(l2_by_box.alloc(ctx->pool, B + 1));
    {
        // one launch for the scans of all counts
        ScanI32 fs[5] = {{d_coll_cnt}, {d_l2_cnt}, {l1_cnt.get()}, {l3_cnt.get()}, {close_cnt.get()}};
        const int64_t ns[5] = {B, B, items_cap, nflat, items_cap};
        int32_t *outs[5] = {coll.starts.get(), l2_by_box.get(), l1_item.get(), l3_item.get(),
                            close_item.get()};
        int64_t *tots[5] = {totals.get() + T_COLL, totals.get() + T_L2, totals.get() + T_L1,
                            totals.get() + T_L3, totals.get() + T_CLOSE};
        BT_CHECK((device_exclusive_scan_batch<int64_t, int32_t, ScanI32, 5>(
            ctx, st->with_extent ? 5 : 4, fs, ns, outs, tots, true)));
    }
    // list 3 per (level, target box): starts, and the compressed (non-empty) numbering.
    // A source level can only have entries if some target box sits on a coarser level (the
    // entries themselves may be boxes of any level: a box stands for the sources below it):
    // the levels up to the first one with target boxes get no rows (at 10^8 sphere points
    // leaves start at level 6: 5 of 12 levels have rows, and these per-(level, box) arrays
    // are what the bookkeeping scans run over).
    {
        const int32_t *tls = st->h_lev_starts.data() + (nlevels + 1);    // target boxes per level
        int first_tgt = 0;
        while (first_tgt < nlevels && tls[first_tgt + 1] == tls[first_tgt]) ++first_tgt;
        // (with a target_boxes_mask the counts are those of the masked target boxes, the only
        // ones the walk makes items for)
        st->l3_lo = std::max(0, std::min(nlevels - 1, first_tgt + 1));
    }
    const int64_t nflat_box = (int64_t) (nlevels - st->l3_lo) * ntb;
    BT_CHECK(st->l3_starts.alloc(ctx->pool, nflat_box + 1));
    l3_box_starts_v2_kernel<<<nblk(nflat_box + 1), 256, 0, ctx->stream>>>(
        nflat_box, (int32_t) ntb, lay, nlevels, first_item.get(), l3_item.get(),
        st->l3_starts.get(), st->l3_lo);
    BT_CHECK(st->l3_cidx.alloc(ctx->pool, nflat_box + 1));
    BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, NonEmptyPred{st->l3_starts.get()}, nflat_box,
                                                      st->l3_cidx.get(), (int32_t *) nullptr, true)));
    Buf<int32_t> marks;
    BT_CHECK(marks.alloc(ctx->pool, 2 * (nlevels + 1)));
    l3_level_marks_kernel<<<1, 128, 0, ctx->stream>>>(nlevels, ntb, st->l3_starts.get(),
                                                     st->l3_cidx.get(), marks.get(), st->l3_lo);
    // list 4 (+ close): counts now, lists after the totals are known
    CsrList &c4 = st->l4;
    c4.n = st->nttp;
    CsrList raw4;
    raw4.n = st->nttp;
    Buf<int32_t> l4_cnt, raw4_cnt;
    BT_CHECK(l4_cnt.alloc(ctx->pool, c4.n));
    if (st->with_extent) BT_CHECK(raw4_cnt.alloc(ctx->pool, raw4.n));
    const bool l4_lattice = !st->with_extent && a.nway == 1;
    if (l4_lattice)
        list4_lattice_kernel<D, false><<<nblk(c4.n), 256, 0, ctx->stream>>>(
            (int32_t) c4.n, st->ttp_boxes.get(), cells, p.box_parent_ids, src_rows,
            src_cnt, P, row_id_mask, row_src_bit, l4_cnt.get(), nullptr);
    else
        list4_kernel<T, D, false><<<nblk(c4.n), 256, 0, ctx->stream>>>(
            a, (int32_t) c4.n, l4_cnt.get(), nullptr, st->with_extent ? raw4_cnt.get() : nullptr, nullptr);
    BT_CHECK(c4.starts.alloc(ctx->pool, c4.n + 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, ScanI32{l4_cnt.get()}, c4.n, c4.starts.get(),
                                                      totals.get() + T_L4, true)));
    if (st->with_extent) {
        BT_CHECK(raw4.starts.alloc(ctx->pool, raw4.n + 1));
        BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, ScanI32{raw4_cnt.get()}, raw4.n,
                                                          raw4.starts.get(), totals.get() + T_L4RAW, true)));
    }

    int64_t h_tot[T_COUNT];
    std::vector<int32_t> h_marks((size_t) 2 * (nlevels + 1));
    BT_CHECK(bt::d2h(ctx, h_tot, totals.get(), sizeof(h_tot)));
    BT_CHECK(bt::d2h(ctx, h_marks.data(), marks.get(), h_marks.size() * 4));
    BT_CHECK(bt::sync_stream(ctx));
    ctx->n_host_syncs++;
    for (int k = T_COLL; k <= T_L4RAW; ++k)
        if (h_tot[k] >= ((int64_t) 1 << 31)) {
            set_error("interaction list exceeds 2^31-1 entries (int32 CSR limit of the reference): "
                      "%lld", (long long) h_tot[k]);
            return BT_ERR_UNSUPPORTED;
        }
    const int64_t novf = h_tot[T_OVF] & 0xffffffffll;
    if (trav_stats) {
        int32_t hc[16];
        BT_HIP_CHECK(hipMemcpy(hc, dbg_counts.get(), 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[bt trav] boxes %lld target boxes %lld items %lld (cap %lld) K1 %d K3 %d Kc %d | "
                "overflow items %lld (list1 %d, list3 %d, close %d; per-colleague items %d) | entries: "
                "coll %lld l2 %lld l1 %lld l3 %lld close %lld l4 %lld\n", (long long) B, (long long) ntb,
                (long long) (h_tot[T_NITEMS] & 0xffffffffll), (long long) items_cap, K1, K3, Kc,
                (long long) novf, hc[0], hc[1], hc[2], hc[3], (long long) h_tot[T_COLL],
                (long long) h_tot[T_L2], (long long) h_tot[T_L1], (long long) h_tot[T_L3],
                (long long) h_tot[T_CLOSE], (long long) h_tot[T_L4]);
    }
    st->l3_level_base.assign((size_t) nlevels, 0);
    st->l3_level_count.assign((size_t) nlevels, 0);
    st->l3_nonempty.assign((size_t) nlevels, 0);
    st->l3_cidx_base.assign((size_t) nlevels, 0);
    for (int l = 0; l < nlevels; ++l) {
        st->l3_level_base[l] = h_marks[l];
        st->l3_level_count[l] = h_marks[l + 1] - h_marks[l];
        st->l3_cidx_base[l] = h_marks[nlevels + 1 + l];
        st->l3_nonempty[l] = h_marks[nlevels + 1 + l + 1] - h_marks[nlevels + 1 + l];
    }
    CsrList &c1 = st->l1;
    c1.n = ntb;
    CsrList &cs = st->close_smaller;
    cs.n = ntb;
    CsrList &cb = st->close_bigger;
    cb.n = ntb;
    coll.total = h_tot[T_COLL];
    st->l2.n = st->nttp;
    st->l2.total = h_tot[T_L2];
    c1.total = h_tot[T_L1];
    const int64_t total3 = h_tot[T_L3];
    cs.total = st->with_extent ? h_tot[T_CLOSE] : 0;
    c4.total = h_tot[T_L4];
    raw4.total = st->with_extent ? h_tot[T_L4RAW] : 0;
    cb.total = raw4.total;       // close lists are only made for target boxes (traversal.py:1003)

    // ---- final places: the caller's block if this is a one-block build ------------------------------
    fill_sizes(st, &st->sizes);
    const bt_trav_packed *pk = nullptr;
    if (st->packed_alloc) {
        BT_CHECK(make_arena(ctx, st));
        pk = st->packed;
    }
    BT_CHECK(place_list(ctx, st, coll.lists, coll.total, pk ? &pk->same_level_non_well_sep_boxes_lists : nullptr));
    compact_coll_rows_v2_kernel<8><<<nblk(B * 8), 256, 0, ctx->stream>>>(
        B, P, row_id_mask, coll_rows.get(), coll.starts.get(), coll.lists.get());
    a.coll_starts = coll.starts.get();
    a.coll_lists = coll.lists.get();
    {
        CsrList &c = st->l2;
        BT_CHECK(c.starts.alloc(ctx->pool, c.n + 1));
        BT_CHECK(place_list(ctx, st, c.lists, c.total, pk ? &pk->from_sep_siblings_lists : nullptr));
        gather_starts_dev_kernel<<<nblk(c.n + 1), 256, 0, ctx->stream>>>(
            (int32_t) c.n, st->ttp_boxes.get(), l2_by_box.get(), (int32_t) B, c.starts.get());
        rows.l2_starts = l2_by_box.get();
        rows.l2_lists = c.lists.get();
        // (0: every lane stores its own entries -- measured 0.2 ms slower on 1.25*10^8 uniform
        // points and on the 10^8 + 10^7 extent tree, the same on a sphere surface)
        rows.l2_stage = 1;
        if (B > 1 && c.total > 0 && st->nparents > 0)
            coll_rows_v3_kernel<D, true><<<nblk(st->nparents * V3Lanes<D>::N), 256, 0, ctx->stream>>>(
                rows, st->parent_boxes.get(), (int32_t) st->nparents, 1, (int32_t) B);
    }
    BT_CHECK(tmark(ctx, st, "trav:colleagues+list2"));

    BT_CHECK(place_list(ctx, st, c1.lists, c1.total, pk ? &pk->neighbor_source_boxes_lists : nullptr));
    BT_CHECK(place_list(ctx, st, st->l3_lists, total3, pk ? &pk->from_sep_smaller_lists[0] : nullptr));
    // (the rows' layout follows the walk kernel that wrote them: RowGroup, or item-major)
    const bool rows_im = walk_g8 == 2;
    const int32_t *sp1_idx = spill_idx.get() ? spill_idx.get() + items_cap : nullptr;
    if (rows_im)
        rows_to_csr_v2_kernel<1, true><<<nblk(items_cap), 256, 0, ctx->stream>>>(
            d_nitems, overflow.get(), row1.get(), K1, l1_item.get(), nullptr, 0, c1.lists.get(), sp1_idx, spill3.get());
    else if (a.targets_have_extent)
        rows_to_csr_v2_kernel<1><<<nblk(items_cap), 256, 0, ctx->stream>>>(
            d_nitems, overflow.get(), row1.get(), K1, l1_item.get(), nullptr, 0, c1.lists.get(), sp1_idx, spill3.get());
    else
        rows_to_csr_v2_kernel<4><<<nblk(items_cap), 256, 0, ctx->stream>>>(
            d_nitems, overflow.get(), row1.get(), K1, l1_item.get(), nullptr, 0, c1.lists.get(), sp1_idx, spill3.get());
    if (total3 > 0) {
        if (rows_im)
            l3_scatter_v2_kernel<1, true><<<nblk(items_cap), 256, lvl_lds, ctx->stream>>>(
                d_nitems, lay, nlevels, overflow.get(), row3.get(), K3,
                l3_item.get(), st->l3_lists.get(), spill_idx.get(), spill3.get());
        else if (a.targets_have_extent)
            l3_scatter_v2_kernel<1><<<nblk(items_cap), 256, lvl_lds, ctx->stream>>>(
                d_nitems, lay, nlevels, overflow.get(), row3.get(), K3,
                l3_item.get(), st->l3_lists.get(), spill_idx.get(), spill3.get());
        else
            l3_scatter_v2_kernel<4><<<nblk(items_cap), 256, lvl_lds, ctx->stream>>>(
                d_nitems, lay, nlevels, overflow.get(), row3.get(), K3,
                l3_item.get(), st->l3_lists.get(), spill_idx.get(), spill3.get());
    }
    if (st->with_extent) {
        BT_CHECK(place_list(ctx, st, cs.lists, cs.total, pk ? &pk->from_sep_close_smaller_lists : nullptr));
        if (cs.total > 0)
        {
            if (rows_im)
                rows_to_csr_v2_kernel<1, true><<<nblk(items_cap), 256, 0, ctx->stream>>>(
                    d_nitems, overflow.get(), rowc.get(), Kc, close_item.get(), nullptr, 0, cs.lists.get());
            else if (a.targets_have_extent)
                rows_to_csr_v2_kernel<1><<<nblk(items_cap), 256, 0, ctx->stream>>>(
                    d_nitems, overflow.get(), rowc.get(), Kc, close_item.get(), nullptr, 0, cs.lists.get());
            else
                rows_to_csr_v2_kernel<4><<<nblk(items_cap), 256, 0, ctx->stream>>>(
                    d_nitems, overflow.get(), rowc.get(), Kc, close_item.get(), nullptr, 0, cs.lists.get());
        }
    }
    if (novf > 0) {
        // items whose lists did not fit their rows: walk again, straight to the final places
        V2Walk wf = w;
        wf.l1_cs = l1_item.get(); wf.l3_cs = l3_item.get();
        wf.close_cs = st->with_extent ? close_item.get() : nullptr;
        wf.l1_lists = c1.lists.get(); wf.l3_lists = st->l3_lists.get();
        wf.close_lists = st->with_extent ? cs.lists.get() : nullptr;
        if (a.targets_have_extent) {
            if (walk_two_pass) walk13_v2_kernel<T, D, false, true, true><<<nblk(novf), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, wf);
            else walk13_v2_kernel<T, D, false, true><<<nblk(novf), 256, walk_lds + lvl_lds + cen_lds, ctx->stream>>>(a, ft, wf);
        } else {
            if (walk_two_pass) walk13_v2_kernel<T, D, false, false, true><<<nblk(novf), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, wf);
            else walk13_v2_kernel<T, D, false, false><<<nblk(novf), 256, walk_lds + lvl_lds, ctx->stream>>>(a, ft, wf);
        }
    }
    BT_CHECK(tmark(ctx, st, "trav:lists 1+3 (final)"));

    // per-box starts from the per-item starts (items of a box are consecutive)
    BT_CHECK(c1.starts.alloc(ctx->pool, ntb + 1));
    gather_i32_kernel<<<nblk(ntb + 1), 256, 0, ctx->stream>>>(
        (int32_t) (ntb + 1), first_item.get(), l1_item.get(), c1.starts.get());
    if (st->with_extent) {
        BT_CHECK(cs.starts.alloc(ctx->pool, ntb + 1));
        gather_i32_kernel<<<nblk(ntb + 1), 256, 0, ctx->stream>>>(
            (int32_t) (ntb + 1), first_item.get(), close_item.get(), cs.starts.get());
    }

    // list 4 (+ close, re-indexed to target boxes: _ListMerger, traversal.py:1259-1344)
    BT_CHECK(place_list(ctx, st, c4.lists, c4.total, pk ? &pk->from_sep_bigger_lists : nullptr));
    if (st->with_extent) BT_CHECK(raw4.lists.alloc(ctx->pool, raw4.total));
    if (l4_lattice)
        list4_lattice_kernel<D, true><<<nblk(c4.n), 256, 0, ctx->stream>>>(
            (int32_t) c4.n, st->ttp_boxes.get(), cells, p.box_parent_ids, src_rows,
            src_cnt, P, row_id_mask, row_src_bit, c4.starts.get(), c4.lists.get());
    else
        list4_kernel<T, D, true><<<nblk(c4.n), 256, 0, ctx->stream>>>(
            a, (int32_t) c4.n, c4.starts.get(), c4.lists.get(),
            st->with_extent ? raw4.starts.get() : nullptr, st->with_extent ? raw4.lists.get() : nullptr);
    if (st->with_extent) {
        Buf<int32_t> ttp_from_all;
        BT_CHECK(ttp_from_all.alloc(ctx->pool, B));
        BT_HIP_CHECK(hipMemsetAsync(ttp_from_all.get(), 0, (size_t) B * 4, ctx->stream));
        reverse_index_kernel<<<nblk(st->nttp), 256, 0, ctx->stream>>>(
            st->ttp_boxes.get(), (int32_t) st->nttp, ttp_from_all.get());
        BT_CHECK(cb.starts.alloc(ctx->pool, cb.n + 1));
        MergeCount mc{st->target_boxes, ttp_from_all.get(), raw4.starts.get()};
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, mc, cb.n, cb.starts.get(),
                                                          (int32_t *) nullptr, true)));
        BT_CHECK(place_list(ctx, st, cb.lists, cb.total, pk ? &pk->from_sep_close_bigger_lists : nullptr));
        merge_copy_kernel<<<nblk(cb.n), 256, 0, ctx->stream>>>(
            (int32_t) cb.n, mc, raw4.lists.get(), cb.starts.get(), cb.lists.get());
    }

    // list 1: order by depth-first rank, insert the own-subtree blocks
    {
        // [len | tier bytes | count | dst | src]: len and tier are cleared together, as one
        // 16-byte-aligned range (a ragged memset is two or three fill kernels)
        const int64_t clr_words = div_up(ntb + div_up(ntb, 4), 4) * 4;
        Buf<int32_t> jobbuf;
        BT_CHECK(jobbuf.alloc(ctx->pool, clr_words + 1 + 2 * ntb));
        static const bool l1_stats = [] { const char *e = getenv("BT_TRAV_STATS"); return e && atoi(e); }();
        Buf<int32_t> l1_dbg;
        if (l1_stats) {
            BT_CHECK(l1_dbg.alloc(ctx->pool, 16));
            BT_HIP_CHECK(hipMemsetAsync(l1_dbg.get(), 0, 64, ctx->stream));
        }
        int32_t *jb = jobbuf.get() + clr_words;
        BlockJobs jobs{jb, jb + 1, jb + 1 + ntb, jobbuf.get(), l1_dbg.get()};
        Buf<uint8_t> tier;
        Buf<int32_t> tier_present;
        tier.set_external((uint8_t *) (jobs.len + ntb), ntb);
        BT_HIP_CHECK(hipMemsetAsync(jobs.len, 0, (size_t) clr_words * 4, ctx->stream));
        if (int32_t *z = (int32_t *) bt::zero_alloc(ctx, 8)) {
            tier_present.set_external(z, 2);
        } else {
            BT_CHECK(tier_present.alloc(ctx->pool, 2));
            BT_HIP_CHECK(hipMemsetAsync(tier_present.get(), 0, 8, ctx->stream));
        }
        l1_finalize32_kernel<T, D><<<nblk(ntb * 16), 256, 0, ctx->stream>>>(
            a, ft, (int32_t) ntb, c1.starts.get(), c1.lists.get(), jobs, tier.get(),
            tier_present.get(), 1);
        // lists longer than 32 entries: counts stay on the device, fixed grids
        Buf<int32_t> pos[2], list[2];
        for (int which = 1; which <= 2; ++which) {
            TierIs pr{tier.get(), (uint8_t) which};
            BT_CHECK(pos[which - 1].alloc(ctx->pool, ntb + 1));
            BT_CHECK(list[which - 1].alloc(ctx->pool, ntb));
            BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, pr, ntb, pos[which - 1].get(),
                                                              (int32_t *) nullptr, true)));
            compact_tier_kernel<<<nblk(ntb), 256, 0, ctx->stream>>>((int32_t) ntb, pr,
                                                                   pos[which - 1].get(),
                                                                   list[which - 1].get());
            const unsigned grid = (unsigned) std::min<int64_t>(ctx->num_cus * 8, std::max<int64_t>(1, ntb));
            if (which == 1)
                l1_finalize_wave_kernel<T, D><<<grid, 256, 0, ctx->stream>>>(
                    a, ft, list[0].get(), pos[0].get() + ntb, c1.starts.get(), c1.lists.get(), jobs, 1);
            else
                l1_finalize_block_kernel<T, D><<<grid, 256, 0, ctx->stream>>>(
                    a, ft, list[1].get(), pos[1].get() + ntb, c1.starts.get(), c1.lists.get(), jobs, 1);
        }
        if (with_blocks)
            copy_rank_blocks_kernel<<<(unsigned) std::min<int64_t>(ctx->num_cus * 8, std::max<int64_t>(1, div_up(ntb, 4))), 256, 0,
                                      ctx->stream>>>(
                (int32_t) ntb, jobs.dst, jobs.src, jobs.len, st->src_by_rank.get(), c1.lists.get());
        if (l1_stats) {
            int32_t hd[16];
            BT_HIP_CHECK(hipMemcpy(hd, l1_dbg.get(), 64, hipMemcpyDeviceToHost));
            fprintf(stderr, "[bt trav] list-1 ordering: %d lists of <= 64 entries (%d with a head of <= 16; "
                    "head entries %d of %d), %d for the wave kernel (%d merged, %d sorted), %d for the "
                    "workgroup kernel, %d serial (longest %d)\n", hd[0], hd[1], hd[8], hd[9], hd[2],
                    hd[6], hd[7], hd[3], hd[4], hd[5]);
        }
        // the scratch buffers above return to the pool at scope exit: all work that uses
        // them is already queued on this stream, and so is whatever reuses them
    }
    BT_CHECK(tmark(ctx, st, "trav:list1 order + list4"));
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

template <class T, int D>
int trav_build_impl(bt_context *ctx, TravState *st, bt_trav_sizes *out)
{
    const bt_trav_params &p = st->p;
    const int64_t B = p.nboxes;
    const int nlevels = p.nlevels;
    const bool sat = p.sources_are_targets;
    st->with_extent = p.sources_have_extent || p.targets_have_extent;
    const int walk_cap = nlevels + 1;
    const size_t walk_lds = (size_t) walk_cap * WALK_THREADS * 4;
    const size_t lvl_lds = (size_t) nlevels * WALK_THREADS * 4;

    BT_CHECK(tmark(ctx, st, "trav:start"));
    // T1 + T2 + structure check, with ONE host synchronisation: the four flag scans
    // give the list sizes (pos[B]) and the level starts (pos[level_start_box_nrs[l]]:
    // the number of listed boxes before the level's first box, which is what
    // traversal.py:361-392 + 2093-2096 compute).
    BT_CHECK(st->d_level_start_box_nrs.alloc(ctx->pool, nlevels + 1));
    // (a fifth list, not part of the result: the boxes that have children, level by
    // level -- the colleague-row kernels run one group of lanes per such box)
    constexpr int NL = 5;
    // the mail block: [NL * (nlevels + 1) level starts | 4 flags | root centre (8 bytes an axis)]
    const size_t n_rows = (size_t) NL * (nlevels + 1);
    const size_t mail_words = n_rows + 4 + 2 * BT_MAX_DIMS;
    BT_CHECK(st->lev_starts.alloc(ctx->pool, (int64_t) mail_words));
    int32_t *d_bad = st->lev_starts.get() + n_rows;
    {
        // the level starts travel as kernel arguments (a host-to-device copy of 50 bytes is
        // a runtime blit with its bubbles either side); the same launch clears the flags
        HostWords hw{};
        if (nlevels + 1 > (int) (sizeof(hw.v) / sizeof(hw.v[0]))) return BT_ERR_INVALID;
        for (int i = 0; i <= nlevels; ++i) hw.v[i] = p.level_start_box_nrs[i];
        store_host_words_kernel<<<1, 128, 0, ctx->stream>>>(
            hw, nlevels + 1, st->d_level_start_box_nrs.get(), d_bad, 4);
    }
    const bool shared_tb = sat && !p.target_boxes_mask;     // target_boxes is source_boxes
    FlagPred preds[NL] = {
        {p.box_flags, p.source_boxes_mask, BT_BOX_IS_SOURCE_BOX},
        {p.box_flags, p.target_boxes_mask, BT_BOX_IS_TARGET_BOX, 1},
        {p.box_flags, p.source_parent_boxes_mask, BT_BOX_HAS_SOURCE_CHILD_BOXES},
        {p.box_flags, p.target_boxes_mask, BT_BOX_HAS_TARGET_CHILD_BOXES | BT_BOX_IS_TARGET_BOX},
        {p.box_flags, nullptr, BT_BOX_HAS_SOURCE_CHILD_BOXES | BT_BOX_HAS_TARGET_CHILD_BOXES}};
    Buf<int32_t> pos[NL];
    {
        // the positions of all lists in one launch
        FlagPred fs[NL];
        int64_t ns[NL];
        int32_t *outs[NL];
        int cnt = 0;
        for (int k = 0; k < NL; ++k) {
            if (k == 1 && shared_tb) continue;
            BT_CHECK(pos[k].alloc(ctx->pool, B + 1));
            fs[cnt] = preds[k]; ns[cnt] = B; outs[cnt] = pos[k].get();
            ++cnt;
        }
        BT_CHECK((device_exclusive_scan_batch<int32_t, int32_t, FlagPred, NL>(
            ctx, cnt, fs, ns, outs, (int32_t *const *) nullptr, true)));
    }
    // the packed box records of the walks, and (unless the generic kernels are forced) the
    // structure check, in one pass over the boxes
    BT_CHECK(st->nodes.alloc(ctx->pool, B * (int64_t) sizeof(Node<T, D>)));
    BT_CHECK(st->child_t.alloc(ctx->pool, B * (1 << D)));
    if (p.force_generic != 1) BT_CHECK(st->child8.alloc(ctx->pool, B));
    if (p.force_generic != 1)
        check_pack_kernel<T, D><<<nblk(B), 256, 0, ctx->stream>>>(
            (int32_t) B, p.aligned_nboxes, p.box_parent_ids, p.box_child_ids, p.box_levels,
            p.box_flags, (const T *) p.box_centers, (T) p.root_extent, (int *) d_bad,
            (Node<T, D> *) st->nodes.get(), st->child_t.get(), st->child8.get());
    else
        pack_nodes_kernel<T, D, true><<<nblk(B), 256, 0, ctx->stream>>>(
            (int32_t) B, p.aligned_nboxes, (const T *) p.box_centers, p.box_levels, p.box_flags,
            p.box_child_ids, (Node<T, D> *) st->nodes.get(), st->child_t.get());
    {
        BoxListMarks mk{};
        for (int k = 0; k < NL; ++k) mk.pos[k] = pos[(k == 1 && shared_tb) ? 0 : k].get();
        mk.rows = st->lev_starts.get();
        mk.centers = p.box_centers; mk.aligned = p.aligned_nboxes;
        mk.root_center = st->lev_starts.get() + n_rows + 4;
        mk.nlevels = nlevels; mk.dims = D; mk.csize = (int) sizeof(T);
        box_list_marks_kernel<<<1, 128, 0, ctx->stream>>>(mk, st->d_level_start_box_nrs.get());
    }
    std::vector<int32_t> h_mail(mail_words, 0);
    BT_CHECK(bt::d2h(ctx, h_mail.data(), st->lev_starts.get(), mail_words * 4));
    BT_CHECK(bt::sync_stream(ctx));
    ctx->n_host_syncs++;
    int32_t hb[4] = {1, 1, 1, 0};
    T root_center[D];
    st->h_lev_starts.assign(h_mail.begin(), h_mail.begin() + (long) n_rows);
    memcpy(hb, h_mail.data() + n_rows, 16);
    memcpy(root_center, h_mail.data() + n_rows + 4, sizeof(T) * D);
    {
        const int32_t *h = st->h_lev_starts.data();
        st->nsb = h[0 * (nlevels + 1) + nlevels];
        st->ntb = h[1 * (nlevels + 1) + nlevels];
        st->nspb = h[2 * (nlevels + 1) + nlevels];
        st->nttp = h[3 * (nlevels + 1) + nlevels];
        st->nparents = h[4 * (nlevels + 1) + nlevels];
        Buf<int32_t> *lists[NL] = {&st->source_boxes, &st->target_boxes_buf,
                                   &st->source_parent_boxes, &st->ttp_boxes, &st->parent_boxes};
        const int64_t ns[NL] = {st->nsb, st->ntb, st->nspb, st->nttp, st->nparents};
        BoxLists bl{};
        for (int k = 0; k < NL; ++k) {
            bl.pred[k] = preds[k];
            bl.pos[k] = pos[k].get();
            bl.out[k] = nullptr;
            if (k == 1 && shared_tb) continue;
            BT_CHECK(lists[k]->alloc(ctx->pool, ns[k]));
            bl.out[k] = lists[k]->get();
        }
        compact_lists_kernel<<<nblk(B), 256, 0, ctx->stream>>>(bl, (int32_t) B);
        st->target_boxes = shared_tb ? st->source_boxes.get() : st->target_boxes_buf.get();
    }

    TravArgs<T, D> a{};
    a.nodes = (const Node<T, D> *) st->nodes.get();
    a.child_t = st->child_t.get();
    a.parent = p.box_parent_ids;
    a.tgt_bbox_min = (const T *) p.box_target_bounding_box_min;
    a.tgt_bbox_max = (const T *) p.box_target_bounding_box_max;
    a.src_counts_cumul = p.box_source_counts_cumul;
    a.aligned = p.aligned_nboxes;
    a.nboxes = (int32_t) B;
    a.root_extent = (T) p.root_extent;
    a.stick_out_factor = (T) p.stick_out_factor;
    a.nway = p.well_sep_is_n_away;
    a.crit = p.from_sep_smaller_crit;
    a.min_nsources_cumul = p.from_sep_smaller_min_nsources_cumul;
    a.targets_have_extent = p.targets_have_extent;
    a.close_lists_exist = st->with_extent;
    a.target_mask = p.target_boxes_mask;
    a.target_boxes = st->target_boxes; a.ntarget_boxes = (int32_t) st->ntb;
    a.ttp_boxes = st->ttp_boxes.get(); a.nttp = (int32_t) st->nttp;

    BT_CHECK(tmark(ctx, st, "trav:boxlists"));

    // ---- which path? ---------------------------------------------------------------
    st->fast = false;
    st->lattice = false;
    if (p.force_generic != 1) {
        st->fast = hb[0] == 0;
        st->has_blocks = hb[2] != 0;
        // lattice kernels: exact lattice centres, and the deepest box many ulps wide
        // (bt_trav_v2.hpp header)
        double mag = std::fabs(p.root_extent);
        for (int ax = 0; ax < D; ++ax)
            mag = std::max(mag, std::fabs((double) root_center[ax]) + 0.5 * std::fabs(p.root_extent));
        const double eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
        const double rmin = p.root_extent / std::ldexp(1.0, nlevels);     // radius of the deepest level
        const bool levels_ok = rmin > 16.0 * (nlevels + 2) * eps * mag;
        st->lattice = st->fast && hb[1] == 0 && hb[3] == 0 && levels_ok && p.well_sep_is_n_away == 1
            && B < ((int64_t) 1 << 26) && nlevels <= 29 && p.force_generic != 2;
    }
    a.fast = st->fast ? 1 : 0;

    if (st->lattice) {
        BT_CHECK((fast_lists_v2<T, D>(ctx, st, a)));
    } else if (st->fast) {
        BT_CHECK((fast_lists<T, D>(ctx, st, a)));
    } else {
        // T3 colleagues
        {
            CsrList &c = st->coll;
            c.n = B;
            BT_CHECK(c.starts.alloc(ctx->pool, B + 1));
            list_kernel<T, D, GEN_COLL, false><<<nblk(B), 256, walk_lds, ctx->stream>>>(a, (int32_t) B, c.starts.get(), nullptr);
            BT_CHECK(counts_to_starts(ctx, c.starts, B, &c.total));
            BT_CHECK(c.lists.alloc(ctx->pool, c.total));
            list_kernel<T, D, GEN_COLL, true><<<nblk(B), 256, walk_lds, ctx->stream>>>(a, (int32_t) B, c.starts.get(), c.lists.get());
        }
        a.coll_starts = st->coll.starts.get();
        a.coll_lists = st->coll.lists.get();

        BT_CHECK(tmark(ctx, st, "trav:colleagues"));
        // T4 list 1
        {
            CsrList &c = st->l1;
            c.n = st->ntb;
            BT_CHECK(c.starts.alloc(ctx->pool, c.n + 1));
            list_kernel<T, D, GEN_L1, false><<<nblk(c.n), 256, walk_lds, ctx->stream>>>(a, (int32_t) c.n, c.starts.get(), nullptr);
            BT_CHECK(counts_to_starts(ctx, c.starts, c.n, &c.total));
            BT_CHECK(c.lists.alloc(ctx->pool, c.total));
            list_kernel<T, D, GEN_L1, true><<<nblk(c.n), 256, walk_lds, ctx->stream>>>(a, (int32_t) c.n, c.starts.get(), c.lists.get());
        }
        BT_CHECK(tmark(ctx, st, "trav:list1"));
        // T5 list 2
        {
            CsrList &c = st->l2;
            c.n = st->nttp;
            BT_CHECK(c.starts.alloc(ctx->pool, c.n + 1));
            list_kernel<T, D, GEN_L2, false><<<nblk(c.n), 256, 0, ctx->stream>>>(a, (int32_t) c.n, c.starts.get(), nullptr);
            BT_CHECK(counts_to_starts(ctx, c.starts, c.n, &c.total));
            BT_CHECK(c.lists.alloc(ctx->pool, c.total));
            list_kernel<T, D, GEN_L2, true><<<nblk(c.n), 256, 0, ctx->stream>>>(a, (int32_t) c.n, c.starts.get(), c.lists.get());
        }

    }
    BT_CHECK(tmark(ctx, st, "trav:list2"));
    // T6 list 3: one walk for all source levels (done together with list 1 on the
    // fast path)
    if (!st->fast) {
        const int64_t ntb = st->ntb;
        const int64_t nflat = (int64_t) nlevels * ntb;
        if (nflat >= ((int64_t) 1 << 31)) {
            set_error("list 3 bookkeeping exceeds int32 range");
            return BT_ERR_UNSUPPORTED;
        }
        BT_CHECK(st->l3_starts.alloc(ctx->pool, nflat + 1));
        CsrList &cs = st->close_smaller;
        cs.n = ntb;
        if (st->with_extent) BT_CHECK(cs.starts.alloc(ctx->pool, ntb + 1));
        list3_kernel<T, D, false><<<nblk(ntb), 256, walk_lds + lvl_lds, ctx->stream>>>(
            a, (int32_t) ntb, nlevels, walk_cap, st->l3_starts.get(), nullptr,
            st->with_extent ? cs.starts.get() : nullptr, nullptr);
        int64_t total = 0;
        BT_CHECK(counts_to_starts(ctx, st->l3_starts, nflat, &total));
        BT_CHECK(st->l3_lists.alloc(ctx->pool, total));
        if (st->with_extent) {
            BT_CHECK(counts_to_starts(ctx, cs.starts, ntb, &cs.total));
            BT_CHECK(cs.lists.alloc(ctx->pool, cs.total));
        }
        list3_kernel<T, D, true><<<nblk(ntb), 256, walk_lds + lvl_lds, ctx->stream>>>(
            a, (int32_t) ntb, nlevels, walk_cap, st->l3_starts.get(), st->l3_lists.get(),
            st->with_extent ? cs.starts.get() : nullptr,
            st->with_extent ? cs.lists.get() : nullptr);

    }
    if (!st->lattice) BT_CHECK(l3_postprocess(ctx, st));
    BT_CHECK(tmark(ctx, st, "trav:list3"));
    // T7 list 4 (+ close, re-indexed to target boxes)
    if (!st->lattice) {
        CsrList &c = st->l4;
        c.n = st->nttp;
        BT_CHECK(c.starts.alloc(ctx->pool, c.n + 1));
        CsrList raw;
        raw.n = st->nttp;
        if (st->with_extent) BT_CHECK(raw.starts.alloc(ctx->pool, raw.n + 1));
        list4_kernel<T, D, false><<<nblk(c.n), 256, 0, ctx->stream>>>(
            a, (int32_t) c.n, c.starts.get(), nullptr,
            st->with_extent ? raw.starts.get() : nullptr, nullptr);
        BT_CHECK(counts_to_starts(ctx, c.starts, c.n, &c.total));
        BT_CHECK(c.lists.alloc(ctx->pool, c.total));
        if (st->with_extent) {
            BT_CHECK(counts_to_starts(ctx, raw.starts, raw.n, &raw.total));
            BT_CHECK(raw.lists.alloc(ctx->pool, raw.total));
        }
        list4_kernel<T, D, true><<<nblk(c.n), 256, 0, ctx->stream>>>(
            a, (int32_t) c.n, c.starts.get(), c.lists.get(),
            st->with_extent ? raw.starts.get() : nullptr,
            st->with_extent ? raw.lists.get() : nullptr);
        if (st->with_extent) {
            Buf<int32_t> ttp_from_all;
            BT_CHECK(ttp_from_all.alloc(ctx->pool, B));
            BT_HIP_CHECK(hipMemsetAsync(ttp_from_all.get(), 0, (size_t) B * 4, ctx->stream));
            reverse_index_kernel<<<nblk(st->nttp), 256, 0, ctx->stream>>>(
                st->ttp_boxes.get(), (int32_t) st->nttp, ttp_from_all.get());
            CsrList &cb = st->close_bigger;
            cb.n = st->ntb;
            BT_CHECK(cb.starts.alloc(ctx->pool, cb.n + 1));
            MergeCount mc{st->target_boxes, ttp_from_all.get(), raw.starts.get()};
            BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, mc, cb.n, cb.starts.get(),
                                                              (int32_t *) nullptr, true)));
            int32_t t = 0;
            BT_CHECK(read_i32(ctx, cb.starts.get() + cb.n, &t));
            cb.total = t;
            BT_CHECK(cb.lists.alloc(ctx->pool, cb.total));
            merge_copy_kernel<<<nblk(cb.n), 256, 0, ctx->stream>>>(
                (int32_t) cb.n, mc, raw.lists.get(), cb.starts.get(), cb.lists.get());
            BT_CHECK(bt::sync_stream(ctx));
        }
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(tmark(ctx, st, "trav:list4"));
    // (no wait here: the export that follows is queued on the same stream)

    fill_sizes(st, out);
    st->sizes = *out;
    st->built = true;
    return BT_OK;
}


}  // namespace

extern "C" {

static int trav_build_entry(bt_context *ctx, const bt_trav_params *p, bt_trav_sizes *out,
                            bt_alloc_fn alloc, void *user, bt_trav_packed *packed)
{
    if (!ctx || !p || !out) { set_error("bt_traversal_build: NULL argument"); return BT_ERR_INVALID; }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (p->dims < 1 || p->dims > BT_MAX_DIMS || (p->coord_kind != BT_F32 && p->coord_kind != BT_F64)) {
        set_error("bt_traversal_build: bad dims/coord_kind");
        return BT_ERR_INVALID;
    }
    if (p->sources_have_extent) {
        set_error("trees with source extent are not supported for traversal generation");
        return BT_ERR_UNSUPPORTED;                       // traversal.py:2002-2006
    }
    if (p->nboxes >= ((int64_t) 1 << 28)) {
        set_error("bt_traversal_build: more than 2^28 boxes are not supported");
        return BT_ERR_UNSUPPORTED;
    }
    if (p->nlevels < 1 || p->nlevels > 32 || p->nboxes < 1 || !p->level_start_box_nrs) {
        set_error("bt_traversal_build: bad nlevels/nboxes");
        return BT_ERR_INVALID;
    }
    if (p->well_sep_is_n_away < 1) {
        set_error("well_sep_is_n_away must be >= 1");
        return BT_ERR_INVALID;
    }
    if (!p->box_centers || !p->box_levels || !p->box_child_ids || !p->box_flags || !p->box_parent_ids) {
        set_error("bt_traversal_build: NULL tree array");
        return BT_ERR_INVALID;
    }
    if (p->targets_have_extent && (!p->box_target_bounding_box_min || !p->box_target_bounding_box_max
                                   || !p->box_source_counts_cumul)) {
        set_error("bt_traversal_build: target-extent trees need the target bounding boxes "
                  "and box_source_counts_cumul");
        return BT_ERR_INVALID;
    }
    bt_free_trav_state(ctx);
    TravState *st = new TravState();
    ctx->trav = st;
    st->p = *p;
    st->nlevels = p->nlevels;
    st->packed_alloc = alloc; st->packed_user = user; st->packed = packed;
    BT_CHECK(bt::zero_begin(ctx));          // (resets the status word too)
    int s = BT_ERR_INVALID;
    const bool f64 = p->coord_kind == BT_F64;
    switch (p->dims) {
    case 1: s = f64 ? trav_build_impl<double, 1>(ctx, st, out) : trav_build_impl<float, 1>(ctx, st, out); break;
    case 2: s = f64 ? trav_build_impl<double, 2>(ctx, st, out) : trav_build_impl<float, 2>(ctx, st, out); break;
    case 3: s = f64 ? trav_build_impl<double, 3>(ctx, st, out) : trav_build_impl<float, 3>(ctx, st, out); break;
    }
    if (s != BT_OK) { bt::drop_pending_reads(ctx); bt_free_trav_state(ctx); }
    return s;
}

static int export_impl(bt_context *ctx, TravState *st, const bt_trav_arrays *o)
{
    const int nl = st->nlevels;
    // arrays that were made in scratch before the caller's block existed are moved by ONE
    // kernel (a dozen runtime copies cost a pipeline bubble each)
    SpanCopies sc{};
    int64_t sc_max = 0;
    auto flush = [&]() -> int {
        if (sc.count > 0) {
            const unsigned gx = (unsigned) std::max<int64_t>(
                1, std::min<int64_t>(ctx->num_cus * 2, div_up(sc_max, 256 * 4)));
            copy_spans_kernel<<<dim3(gx, (unsigned) sc.count), 256, 0, ctx->stream>>>(sc);
            BT_HIP_CHECK(hipGetLastError());
        }
        sc.count = 0; sc_max = 0;
        return BT_OK;
    };
    auto copy = [&](int32_t *dst, const int32_t *src, int64_t n) -> int {
        // (dst == src: the list was built in the caller's block, nothing to move)
        if (n <= 0 || !dst || dst == src) return BT_OK;
        if (sc.count == SpanCopies::MAX) BT_CHECK(flush());
        sc.src[sc.count] = src; sc.dst[sc.count] = dst; sc.n[sc.count] = n;
        sc.count++;
        sc_max = std::max(sc_max, n);
        return BT_OK;
    };
    BT_CHECK(copy(o->source_boxes, st->source_boxes.get(), st->nsb));
    if (!st->p.sources_are_targets || st->p.target_boxes_mask)
        BT_CHECK(copy(o->target_boxes, st->target_boxes, st->ntb));
    BT_CHECK(copy(o->source_parent_boxes, st->source_parent_boxes.get(), st->nspb));
    BT_CHECK(copy(o->target_or_target_parent_boxes, st->ttp_boxes.get(), st->nttp));
    BT_CHECK(copy(o->level_start_source_box_nrs, st->lev_starts.get() + 0 * (nl + 1), nl + 1));
    BT_CHECK(copy(o->level_start_target_box_nrs, st->lev_starts.get() + 1 * (nl + 1), nl + 1));
    BT_CHECK(copy(o->level_start_source_parent_box_nrs, st->lev_starts.get() + 2 * (nl + 1), nl + 1));
    BT_CHECK(copy(o->level_start_target_or_target_parent_box_nrs,
                  st->lev_starts.get() + 3 * (nl + 1), nl + 1));
    auto put = [&](const CsrList &c, int32_t *starts, int32_t *lists) -> int {
        BT_CHECK(copy(starts, c.starts.get(), c.n + 1));
        BT_CHECK(copy(lists, c.lists.get(), c.total));
        return BT_OK;
    };
    BT_CHECK(put(st->coll, o->same_level_non_well_sep_boxes_starts, o->same_level_non_well_sep_boxes_lists));
    BT_CHECK(put(st->l1, o->neighbor_source_boxes_starts, o->neighbor_source_boxes_lists));
    if (st->l2_pending.empty()) {
        BT_CHECK(put(st->l2, o->from_sep_siblings_starts, o->from_sep_siblings_lists));
    } else {
        BT_CHECK(copy(o->from_sep_siblings_starts, st->l2.starts.get(), st->l2.n + 1));
        for (const L2Pending &pend : st->l2_pending)
            compact_strided_rows_kernel<16><<<nblk((int64_t) pend.nb * 16), 256, 0, ctx->stream>>>(
                pend.nb, pend.stride, pend.rows.get(), pend.rel.get(), (int32_t) pend.base,
                o->from_sep_siblings_lists);
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK(put(st->l4, o->from_sep_bigger_starts, o->from_sep_bigger_lists));
    if (st->with_extent) {
        BT_CHECK(put(st->close_smaller, o->from_sep_close_smaller_starts, o->from_sep_close_smaller_lists));
        BT_CHECK(put(st->close_bigger, o->from_sep_close_bigger_starts, o->from_sep_close_bigger_lists));
    }
    const int64_t ntb = st->ntb;
    L3CompressAll ca{};
    for (int l = 0; l < nl; ++l) {
        if (!o->from_sep_smaller_starts[l]) {
            set_error("bt_traversal_export: NULL list-3 output for level %d", l);
            return BT_ERR_INVALID;
        }
        BT_CHECK(copy(o->from_sep_smaller_lists[l], st->l3_lists.get() + st->l3_level_base[l],
                      st->l3_level_count[l]));
        ca.starts[l] = o->from_sep_smaller_starts[l];
        ca.nonempty[l] = o->from_sep_smaller_nonempty_indices[l];
        ca.cidx[l] = o->from_sep_smaller_compressed_indices[l];
        ca.tboxes[l] = o->target_boxes_sep_smaller[l];
        ca.lev_base[l] = (int32_t) st->l3_level_base[l];
        ca.cidx_base[l] = (int32_t) st->l3_cidx_base[l];
        ca.lev_count[l] = (int32_t) st->l3_level_count[l];
    }
    BT_CHECK(flush());
    ca.lev0 = st->l3_lo;
    l3_compress_all_kernel<<<nblk((int64_t) nl * (ntb + 1)), 256, 0, ctx->stream>>>(
        (int32_t) ntb, nl, st->l3_starts.get(), st->l3_cidx.get(), st->target_boxes, ca);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(tmark(ctx, st, "trav:export"));
    ctx->n_host_syncs++;
    return finish_call(ctx);        // waits for the stream and reports device-side failures
                                    // (stream-ordered contexts: queues the status read)
}

int bt_traversal_build(bt_context *ctx, const bt_trav_params *p, bt_trav_sizes *out)
{
    bt::CallScope bt_call_scope_(ctx);
    return trav_build_entry(ctx, p, out, nullptr, nullptr, nullptr);
}

int bt_traversal_export(bt_context *ctx, const bt_trav_arrays *o)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !o) { set_error("bt_traversal_export: NULL argument"); return BT_ERR_INVALID; }
    TravState *st = ctx->trav;
    if (!st || !st->built) {
        set_error("bt_traversal_export: no traversal has been built on this context");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    return export_impl(ctx, st, o);
}

int bt_traversal_build_packed(bt_context *ctx, const bt_trav_params *p, bt_alloc_fn alloc,
                              void *user, bt_trav_packed *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!alloc || !out) { set_error("bt_traversal_build_packed: NULL argument"); return BT_ERR_INVALID; }
    memset(out, 0, sizeof(*out));
    bt_trav_sizes sizes;
    host_trace("trav:enter");
    BT_CHECK(trav_build_entry(ctx, p, &sizes, alloc, user, out));
    host_trace("trav:built");
    TravState *st = ctx->trav;
    if (!st->arena) BT_CHECK(make_arena(ctx, st));      // the general paths: sizes known only now
    bt_trav_arrays o{};
    o.source_boxes = span_ptr(st, out->source_boxes);
    o.target_boxes = span_ptr(st, out->target_boxes);
    o.source_parent_boxes = span_ptr(st, out->source_parent_boxes);
    o.target_or_target_parent_boxes = span_ptr(st, out->target_or_target_parent_boxes);
    o.same_level_non_well_sep_boxes_starts = span_ptr(st, out->same_level_non_well_sep_boxes_starts);
    o.same_level_non_well_sep_boxes_lists = span_ptr(st, out->same_level_non_well_sep_boxes_lists);
    o.neighbor_source_boxes_starts = span_ptr(st, out->neighbor_source_boxes_starts);
    o.neighbor_source_boxes_lists = span_ptr(st, out->neighbor_source_boxes_lists);
    o.from_sep_siblings_starts = span_ptr(st, out->from_sep_siblings_starts);
    o.from_sep_siblings_lists = span_ptr(st, out->from_sep_siblings_lists);
    o.from_sep_bigger_starts = span_ptr(st, out->from_sep_bigger_starts);
    o.from_sep_bigger_lists = span_ptr(st, out->from_sep_bigger_lists);
    o.from_sep_close_smaller_starts = span_ptr(st, out->from_sep_close_smaller_starts);
    o.from_sep_close_smaller_lists = span_ptr(st, out->from_sep_close_smaller_lists);
    o.from_sep_close_bigger_starts = span_ptr(st, out->from_sep_close_bigger_starts);
    o.from_sep_close_bigger_lists = span_ptr(st, out->from_sep_close_bigger_lists);
    for (int l = 0; l < st->nlevels; ++l) {
        o.from_sep_smaller_starts[l] = span_ptr(st, out->from_sep_smaller_starts[l]);
        o.from_sep_smaller_lists[l] = span_ptr(st, out->from_sep_smaller_lists[l]);
        o.from_sep_smaller_nonempty_indices[l] = span_ptr(st, out->from_sep_smaller_nonempty_indices[l]);
        o.from_sep_smaller_compressed_indices[l] = span_ptr(st, out->from_sep_smaller_compressed_indices[l]);
        o.target_boxes_sep_smaller[l] = span_ptr(st, out->target_boxes_sep_smaller[l]);
    }
    int s = export_impl(ctx, st, &o);
    host_trace("trav:leave");
    return s;
}


int bt_merge_csr_lists(bt_context *ctx, int nlists, const int32_t *const *starts,
                       const int32_t *const *lists, int64_t nrows, int32_t *out_starts,
                       int32_t *out_lists)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nlists < 1 || nlists > 4 || !starts || !lists || nrows < 0 || !out_starts) {
        set_error("bt_merge_csr_lists: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    MergeRows m{};
    m.nlists = nlists;
    for (int k = 0; k < nlists; ++k) { m.starts[k] = starts[k]; m.lists[k] = lists[k]; }
    MergeRowCount f{m};
    BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, f, nrows, out_starts,
                                                      (int32_t *) nullptr, true)));
    if (nrows > 0 && out_lists)
        merge_rows_kernel<<<nblk(nrows), 256, 0, ctx->stream>>>((int32_t) nrows, m, out_starts,
                                                                out_lists);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"
