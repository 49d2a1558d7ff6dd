// Lattice formulation of the parent-colleague list kernels (bt_trav_fast.hpp) for
// well_sep_is_n_away == 1.  Included by bt_trav.hip inside its anonymous namespace.
//
// A tree whose box centres are exactly "parent centre +/- root_extent / 2^(level+1)"
// (check_structure_kernel verifies it bit for bit) is a set of cells of a dyadic
// lattice.  On such a tree the reference's adjacency predicate
//     |c_t - c_s|_inf <= r_t + r_s + min(r_t, r_s)          (traversal.py:255-320)
// separates lattice distances r_t + r_s (touching or overlapping) from
// r_t + r_s + 2 min(r_t, r_s) (the next possible value) with a margin of min(r) on
// either side, while the centres carry rounding errors of at most a few ulp of the
// coordinate magnitude: as long as the deepest box is many ulps wide (the host
// checks this: v2_levels_ok) the float test and the integer test
//     boxes adjacent  <=>  per axis  -1 <= rel <= 2^k
// (rel = source cell minus first cell of the target box in units of the source's
// level, k = level difference) agree on every pair.  The kernels below therefore
// never load a box centre for an adjacency test.  What remains a float test is what
// depends on particle extents: the list-3/4 separation criteria with target
// extents (traversal.py:757-820, 933-972).
//
// Layout of the lattice information:
//  * colleague rows [nboxes][3^d - 1] hold "box | code << 26", code = 2 bits per
//    axis: (offset of the colleague's cell from the box's cell) + 1;
//  * child_t[box][2^d] holds "child | flags": bit 28 = the child is a source box,
//    bit 29 = it has source child boxes (tree walks read nothing else per step);
//  * ICell{c[3], lf}: integer cell coordinates at the box's own level.
// Boxes < 2^26 and levels <= 29 are required (the host falls back to the float
// kernels otherwise).

constexpr int V2_CODE_SHIFT = 26;
constexpr uint32_t V2_ID_MASK = (1u << V2_CODE_SHIFT) - 1u;
constexpr uint32_t V2_CODE_SELF = 0x15u;            // offsets (0,0,0): 01 01 01

struct alignas(16) ICell {
    uint32_t c[3];
    uint32_t lf;          // level | flags << 8
};

template <int D> struct V2Dims {
    static constexpr int C = 1 << D;
    static constexpr int P = (D == 1 ? 3 : D == 2 ? 9 : 27) - 1;
};

__device__ __forceinline__ int v2_off(uint32_t e, int ax)
{
    return (int) ((e >> (V2_CODE_SHIFT + 2 * ax)) & 3u) - 1;
}

template <int D>
__device__ __forceinline__ int v2_mbit(int m, int ax) { return (m >> (D - 1 - ax)) & 1; }

// ---- per-tree tables: slot in the parent, integer cells, depth-first ranks ----------

template <int D>
__global__ __launch_bounds__(256) void dfs_rank_cells_kernel(int32_t b0, int32_t nb, int64_t aligned,
        const int32_t *child, const int32_t *size, const uint8_t *levels, const uint8_t *flags,
        int32_t *rank, int32_t *box_of_rank, uint8_t *slot_of, ICell *cells)
{
    constexpr int C = 1 << D;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const int32_t p = b0 + i;          // parent whose children get their ranks
    ICell pc;
    if (p == 0) {
        rank[0] = 0; box_of_rank[0] = 0; slot_of[0] = 0;
        pc.c[0] = pc.c[1] = pc.c[2] = 0;
        pc.lf = (uint32_t) levels[0] | ((uint32_t) flags[0] << 8);
        cells[0] = pc;
    } else {
        pc = cells[p];
    }
    int32_t run = ((p == 0) ? 0 : rank[p]) + 1;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = child[(int64_t) m * aligned + p];
        if (c) {
            rank[c] = run;
            box_of_rank[run] = c;
            run += size[c];
            slot_of[c] = (uint8_t) m;
            ICell cc;
            cc.c[0] = cc.c[1] = cc.c[2] = 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) cc.c[ax] = 2u * pc.c[ax] + (uint32_t) v2_mbit<D>(m, ax);
            cc.lf = (uint32_t) levels[c] | ((uint32_t) flags[c] << 8);
            cells[c] = cc;
        }
    }
}

// ---- colleague rows (+ list-2 counts), one level, top-down ---------------------------

template <int D>
struct V2Rows {
    const int32_t *child_t;        // packed
    const int32_t *parent;
    const uint8_t *flags;
    const uint8_t *slot_of;
    const int8_t *target_mask;
    int32_t *coll_rows, *coll_cnt, *coll_ins;
    int32_t *srccoll_rows, *srccoll_cnt;
    int32_t *l2_cnt;               // [nboxes]
    const int32_t *l2_starts;      // fill pass: [nboxes + 1]
    int32_t *l2_lists;
};

// C lanes per box, lane m looks at child slot m of every candidate parent
// (colleagues of the box's parent, and the parent itself at its depth-first place).
// FILL = false: build the box's own row, count its list 2.
// FILL = true : write list 2 (all levels in one launch; rows are complete by then).
template <int D, bool FILL>
__global__ __launch_bounds__(256) void coll_rows_v2_kernel(V2Rows<D> t, int32_t b0, int32_t nb)
{
    constexpr int C = 1 << D;
    constexpr int P = V2Dims<D>::P;
    const int32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int32_t g = tid / C;
    const int m = tid % C;
    if (g >= nb) return;                 // whole groups drop out together
    const int32_t b = b0 + g;
    const int lane = threadIdx.x & 63;
    const int gshift = lane / C * C;
    const uint64_t gmask = (C == 64) ? ~0ull : ((1ull << C) - 1ull);
    const uint64_t lanes_below = (1ull << m) - 1ull;

    int32_t lcur = 0;
    if (FILL) {
        lcur = t.l2_starts[b];
        if (t.l2_starts[b + 1] == lcur) return;      // group-uniform
    }
    const int32_t p = t.parent[b];
    const int sb = t.slot_of[b];
    const uint8_t fl = t.flags[b];
    const bool ttp = (fl & (BT_BOX_HAS_TARGET_CHILD_BOXES | BT_BOX_IS_TARGET_BOX))
        && (!t.target_mask || t.target_mask[b]);    // list 2 only for wanted boxes
    const int32_t *prow = t.coll_rows + (int64_t) p * P;
    const int n = t.coll_cnt[p];
    const int ins = t.coll_ins[p];
    int rel0[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) rel0[ax] = v2_mbit<D>(m, ax) - v2_mbit<D>(sb, ax);

    int32_t *crow = t.coll_rows + (int64_t) b * P;
    int32_t *srow = t.srccoll_rows + (int64_t) b * P;
    int32_t ccur = 0, scur = 0, lcnt = 0;
    // the parent's row is read once, spread over the group's lanes (entry i sits in
    // register i / C of lane i % C): the loop below then depends on ONE load per
    // candidate (its child slot), not on a chain row entry -> child slot
    constexpr int NREG = (P + C - 1) / C;
    uint32_t preg[NREG];
#pragma unroll
    for (int j = 0; j < NREG; ++j) preg[j] = (m + C * j < n) ? (uint32_t) prow[m + C * j] : 0u;
    constexpr int UNR = 8;
    for (int i0 = 0; i0 <= n; i0 += UNR) {
        uint32_t es[UNR], chs[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u;
            const int ii = (i < ins) ? i : i - 1;        // index into the parent's row
            uint32_t sel = preg[0];
#pragma unroll
            for (int j = 1; j < NREG; ++j) sel = (ii / C == j) ? preg[j] : sel;
            const uint32_t e = (uint32_t) __shfl((int) sel, (ii < 0 ? 0 : ii) % C, C);
            es[u] = (i > n) ? 0u : (i == ins ? ((uint32_t) p | (V2_CODE_SELF << V2_CODE_SHIFT)) : e);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            chs[u] = (i0 + u <= n)
                ? (uint32_t) t.child_t[(int64_t) (es[u] & V2_ID_MASK) * C + m] : 0u;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (i0 + u > n) break;                      // uniform within the group
            const uint32_t ch = chs[u] & CH_ID_MASK;
            bool adjacent = true;
            uint32_t code = 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const int rel = 2 * v2_off(es[u], ax) + rel0[ax];
                adjacent = adjacent && rel >= -1 && rel <= 1;
                code |= (uint32_t) ((rel + 1) & 3) << (2 * ax);
            }
            const bool is_coll = ch != 0 && ch != (uint32_t) b && adjacent;   // traversal.py:429-442
            const bool is_l2 = ch != 0 && !adjacent && ttp;                  // traversal.py:588-597
            if (!FILL) {
                const bool is_src = is_coll && (chs[u] & CH_SRC);
                const uint64_t bc = (__ballot(is_coll) >> gshift) & gmask;
                const uint64_t bs = (__ballot(is_src) >> gshift) & gmask;
                const uint32_t entry = ch | (code << V2_CODE_SHIFT);
                // the candidates come in depth-first order; b itself is one of them
                if (ch == (uint32_t) b) t.coll_ins[b] = ccur + __popcll(bc & lanes_below);
                if (is_coll) crow[ccur + __popcll(bc & lanes_below)] = (int32_t) entry;
                if (is_src) srow[scur + __popcll(bs & lanes_below)] = (int32_t) entry;
                ccur += __popcll(bc);
                scur += __popcll(bs);
                lcnt += is_l2 ? 1 : 0;
            } else {
                const uint64_t bl = (__ballot(is_l2) >> gshift) & gmask;
                if (is_l2) t.l2_lists[lcur + __popcll(bl & lanes_below)] = (int32_t) ch;
                lcur += __popcll(bl);
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (int off = C / 2; off > 0; off >>= 1) lcnt += __shfl_xor(lcnt, off, C);
        if (m == 0) { t.coll_cnt[b] = ccur; t.srccoll_cnt[b] = scur; t.l2_cnt[b] = lcnt; }
    }
}

// colleague CSR from the rows (codes stripped); LANES lanes per row
template <int LANES>
__global__ __launch_bounds__(256) void compact_coll_rows_v2_kernel(int64_t nrows, int stride,
        const int32_t *rows, const int32_t *starts, int32_t *lists)
{
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LANES;
    const int lane = (int) (gid % LANES);
    if (r >= nrows) return;
    const int32_t s = starts[r], e = starts[r + 1];
    const int32_t *row = rows + r * stride;
    for (int32_t k = lane; k < e - s; k += LANES)
        lists[(int64_t) s + k] = (int32_t) ((uint32_t) row[k] & V2_ID_MASK);
}

// ---- work items (see make_items_kernel) -------------------------------------------------

template <bool FILL>
__global__ __launch_bounds__(256) void make_items_v2_kernel(int32_t ntb, const int32_t *target_boxes,
        const ICell *cells, const int32_t *coll_cnt, int heavy_max_level,
        int32_t *cnt_or_first, int32_t *item_tbn, int32_t *item_slot)
{
    const int32_t tbn = blockIdx.x * 256 + threadIdx.x;
    if (tbn >= ntb) return;
    const int32_t b = target_boxes[tbn];
    const int32_t ncoll = coll_cnt[b];
    const bool heavy = (int) (cells[b].lf & 0xffu) <= heavy_max_level && ncoll > 1;
    if (!FILL) {
        cnt_or_first[tbn] = heavy ? ncoll + 1 : 1;
    } else {
        const int32_t it = cnt_or_first[tbn];
        if (!heavy) {
            item_tbn[it] = tbn; item_slot[it] = SLOT_ALL;
        } else {
            item_tbn[it] = tbn; item_slot[it] = SLOT_SELF;
            for (int32_t k = 0; k < ncoll; ++k) { item_tbn[it + 1 + k] = tbn; item_slot[it + 1 + k] = k; }
        }
    }
}

// ---- lists 1 and 3 (+ close) in ONE walk per work item ------------------------------------
//
// ROWS = true : single pass.  Entries go to fixed-capacity scratch rows (tiles of 64
//   items, entry j of the 64 items of a tile contiguous: the lanes of a wave write one
//   256-byte stretch), exact counts are recorded, and an item whose lists do not fit
//   is put on the overflow list.
// ROWS = false: the items of the overflow list are walked again and write straight to
//   their final places.

// Per-level list-3 counts of the items, "staircase" layout: an item of a target box at
// level tl can only have entries at source levels > tl, and items are numbered in
// target-box (= level) order, so row l needs columns for the items of levels < l only:
// ecap[l] = an upper bound of their number, base[l] = sum of ecap[l'] for l' < l.
struct L3Layout {
    int32_t base[BT_MAX_LEVELS + 1];
    int32_t ecap[BT_MAX_LEVELS + 1];
};

struct V2Walk {
    const ICell *cells;
    const uint8_t *flags;
    const int32_t *child_t;            // packed
    const int32_t *coll_rows, *coll_cnt, *srccoll_rows, *srccoll_cnt;
    const int32_t *item_tbn, *item_slot;
    const int32_t *d_nitems;           // actual item count (device)
    int32_t items_cap;                 // columns of the item arrays (>= the item count)
    L3Layout lay;
    int nlevels, walk_cap;
    int with_blocks;                   // own-subtree blocks exist (extents)
    // rows
    int32_t *row1, *row3, *rowc;
    uint8_t *row3lev;
    int K1, K3, Kc;
    int32_t *l1_cs, *l3_cs, *close_cs; // counts (ROWS) / starts (!ROWS)
    uint8_t *overflow;                 // [items_cap]
    int32_t *ovf_count, *ovf_list;
    int32_t *dbg_counts;               // optional [4]
    // final places (!ROWS)
    int32_t *l1_lists, *l3_lists, *close_lists;
};

template <bool ROWS>
struct V2Emit {                        // list 1 / close list of one item
    int32_t *base;
    int stride, cap, n;
    __device__ __forceinline__ void operator()(int32_t v)
    {
        if (!ROWS || n < cap) base[(int64_t) n * stride] = v;
        ++n;
    }
};

template <class T, int D, bool ROWS>
__global__ __launch_bounds__(256) void walk13_v2_kernel(TravArgs<T, D> a, FastTree ft, V2Walk w)
{
    constexpr int C = 1 << D;
    constexpr int P = V2Dims<D>::P;
    int32_t item;
    if (ROWS) {
        item = blockIdx.x * 256 + threadIdx.x;
        if (item >= *w.d_nitems) {
            if (item < w.items_cap) {
                // the count arrays are scanned over all their columns
                for (int l = 0; l < w.nlevels; ++l)
                    if (item < w.lay.ecap[l]) w.l3_cs[w.lay.base[l] + item] = 0;
                w.l1_cs[item] = 0;
                if (w.close_cs) w.close_cs[item] = 0;
                w.overflow[item] = 0;
            }
            return;
        }
    } else {
        const int32_t idx = blockIdx.x * 256 + threadIdx.x;
        if (idx >= *w.ovf_count) return;
        item = w.ovf_list[idx];
    }
    const int32_t tbn = w.item_tbn[item];
    const int slot = w.item_slot[item];
    const int32_t b = a.target_boxes[tbn];
    const ICell cell = w.cells[b];
    const int tl = (int) (cell.lf & 0xffu);
    const uint8_t bflags = (uint8_t) (cell.lf >> 8);

    const int64_t tile = (int64_t) (item >> 6) * 64;
    const int tl64 = item & 63;
    V2Emit<ROWS> e1, ec;
    if (ROWS) {
        e1 = V2Emit<ROWS>{w.row1 + tile * w.K1 + tl64, 64, w.K1, 0};
        ec = V2Emit<ROWS>{w.rowc ? w.rowc + tile * w.Kc + tl64 : nullptr, 64, w.rowc ? w.Kc : 0, 0};
    } else {
        e1 = V2Emit<ROWS>{w.l1_lists + w.l1_cs[item], 1, INT_MAX, 0};
        ec = V2Emit<ROWS>{w.close_lists ? w.close_lists + w.close_cs[item] : nullptr, 1, INT_MAX, 0};
    }
    int32_t *lvl = s_walk_lds + w.walk_cap * WALK_THREADS + threadIdx.x;
    int n3 = 0;
    int32_t *row3 = ROWS ? w.row3 + tile * w.K3 + tl64 : nullptr;
    uint8_t *row3lev = ROWS ? w.row3lev + tile * w.K3 + tl64 : nullptr;
    if (ROWS) {
        for (int l = 0; l < w.nlevels; ++l) lvl[l * WALK_THREADS] = 0;
    } else {
        for (int l = 0; l < w.nlevels; ++l)
            lvl[l * WALK_THREADS] = item < w.lay.ecap[l] ? w.l3_cs[w.lay.base[l] + item] : 0;
    }
    auto emit3 = [&](int lev, int32_t box) {
        if (ROWS) {
            ++lvl[lev * WALK_THREADS];
            if (n3 < w.K3) { row3[(int64_t) n3 * 64] = box; row3lev[(int64_t) n3 * 64] = (uint8_t) lev; }
            ++n3;
        } else {
            w.l3_lists[lvl[lev * WALK_THREADS]++] = box;
        }
    };

    if (slot < 0) {
        if (w.flags[0] & BT_BOX_IS_SOURCE_BOX) e1(ft.dfs_rank[0]);       // traversal.py:489-495
        // b itself
        if (tl >= 1 && (bflags & BT_BOX_IS_SOURCE_BOX)) e1(ft.dfs_rank[b]);
        // coarser levels: the ancestors and their source-box colleagues.  A colleague
        // of the ancestor at offset o touches b iff b sits at the matching face of the
        // ancestor along every axis with o != 0.
        if (tl >= 2) {
            int32_t anc = a.parent[b];
            for (int k = tl - 1; k >= 1; --k, anc = a.parent[anc]) {
                if (w.flags[anc] & BT_BOX_IS_SOURCE_BOX) e1(ft.dfs_rank[anc]);
                const uint32_t mask = (1u << (tl - k)) - 1u;
                const int32_t *srow = w.srccoll_rows + (int64_t) anc * P;
                const int ns = w.srccoll_cnt[anc];
                for (int i = 0; i < ns; ++i) {
                    const uint32_t e = (uint32_t) srow[i];
                    bool adjacent = true;
#pragma unroll
                    for (int ax = 0; ax < D; ++ax) {
                        const int o = v2_off(e, ax);
                        const uint32_t r = cell.c[ax] & mask;
                        adjacent = adjacent && (o == 0 || (o < 0 ? r == 0u : r == mask));
                    }
                    if (adjacent) e1(ft.dfs_rank[e & V2_ID_MASK]);
                }
            }
        }
    }

    // the box's LAST item reserves the space of the own-subtree block at the end of the
    // box's list-1 segment
    const int32_t ncoll = w.coll_cnt[b];
    int32_t blk_len = 0;
    if (w.with_blocks) {
        const bool last_item = slot == SLOT_ALL || slot == ncoll - 1;
        if (last_item && (bflags & BT_BOX_HAS_SOURCE_CHILD_BOXES)) {
            const int32_t my_rank = ft.dfs_rank[b];
            blk_len = ft.src_prefix[my_rank + ft.subtree_size[b]] - ft.src_prefix[my_rank + 1];
        }
    }

    // float data of the separation criteria with target extents (traversal.py:757-820)
    T tc[D], ext_center[D], radii_vec[D];
    T stickout_rad = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) { tc[i] = 0; ext_center[i] = 0; radii_vec[i] = 0; }
    if (a.targets_have_extent) {
        load_center(a, b, tc);
        if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_STATIC_L2) {
            stickout_rad = (1 + a.stick_out_factor) * level_to_rad(a.root_extent, tl);
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {          // load_true_box_extent, :177-198
                const T mn = a.tgt_bbox_min[i * a.aligned + b];
                const T mx = a.tgt_bbox_max[i * a.aligned + b];
                ext_center[i] = ((T) 0.5) * (mn + mx);
                radii_vec[i] = ((T) 0.5) * (mx - mn);
            }
        }
    }

    // colleagues and everything below them
    int c0 = 0, c1 = ncoll;
    if (slot >= 0) { c0 = slot; c1 = (slot + 1 < ncoll) ? slot + 1 : ncoll; }
    else if (slot == SLOT_SELF) c1 = 0;
    const int32_t *crow = w.coll_rows + (int64_t) b * P;
    int32_t *stk = s_walk_lds + threadIdx.x;
    for (int ci = c0; ci < c1; ++ci) {
        const uint32_t ce = (uint32_t) crow[ci];
        const int32_t nws = (int32_t) (ce & V2_ID_MASK);
        const uint8_t cfl = w.flags[nws];
        // a colleague is adjacent (well_sep_is_n_away == 1)
        if (cfl & BT_BOX_IS_SOURCE_BOX) e1(ft.dfs_rank[nws]);
        if (!(cfl & BT_BOX_HAS_SOURCE_CHILD_BOXES)) continue;
        int prel[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) prel[ax] = v2_off(ce, ax);
        int size = 0, mnr = 0;
        int32_t parent = nws;
        bool go = true;
        while (go) {
            const uint32_t raw = (uint32_t) w.child_t[(int64_t) parent * C + mnr];
            const int32_t wb = (int32_t) (raw & CH_ID_MASK);
            bool descend = false;
            int rel[D];
            if (wb && (raw & (CH_SRC | CH_HSC))) {
                const int k = size + 1;                 // level of wb minus tl
                bool in_list_1 = true;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    rel[ax] = 2 * prel[ax] + v2_mbit<D>(mnr, ax);
                    in_list_1 = in_list_1 && rel[ax] >= -1 && rel[ax] <= (1 << k);
                }
                const int wl = tl + k;
                if (in_list_1) {
                    if (raw & CH_SRC) e1(ft.dfs_rank[wb]);
                    descend = (raw & CH_HSC) != 0;
                } else {
                    bool meets = true;
                    if (a.targets_have_extent) {
                        T wc[D];
                        load_center(a, wb, wc);
                        const T source_rad = level_to_rad(a.root_extent, wl);
                        if (a.crit == BT_CRIT_STATIC_LINF) {
                            T l_inf = 0;
#pragma unroll
                            for (int q = 0; q < D; ++q) {
                                T d = tc[q] - wc[q];
                                d = (d < 0) ? -d : d;
                                const T v = d - stickout_rad - source_rad;
                                l_inf = (v > l_inf) ? v : l_inf;
                            }
                            meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                        } else if (a.crit == BT_CRIT_PRECISE_LINF) {
                            T l_inf = 0;
#pragma unroll
                            for (int q = 0; q < D; ++q) {
                                T d = ext_center[q] - wc[q];
                                d = (d < 0) ? -d : d;
                                const T v = d - radii_vec[q] - source_rad;
                                l_inf = (v > l_inf) ? v : l_inf;
                            }
                            meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                        } else {
                            T l2sq = 0;
#pragma unroll
                            for (int q = 0; q < D; ++q) {
                                const T d = tc[q] - wc[q];
                                l2sq = l2sq + d * d;
                            }
                            const T rhs = sqrt(l2sq) - sqrt((T) D) * stickout_rad - source_rad;
                            meets = ((2 - 8 * Eps<T>::v) * source_rad <= rhs);
                        }
                    }
                    // (counts are >= 0: without a threshold nothing is forced and the
                    // random load of the count is saved)
                    const bool force_close = a.close_lists_exist && a.min_nsources_cumul > 0
                        && (a.src_counts_cumul[wb] < a.min_nsources_cumul);
                    if (meets && !force_close) {
                        emit3(wl, wb);
                    } else if (a.close_lists_exist) {
                        if (raw & CH_SRC) ec(wb);
                        descend = (raw & CH_HSC) != 0;
                    }
                }
            }
            if (descend) {
                stk[size * WALK_THREADS] = parent | (mnr << 28);
                ++size;
                parent = wb; mnr = 0;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) prel[ax] = rel[ax];
                continue;
            }
            while (true) {                              // walk_advance
                ++mnr;
                if (mnr < C) break;
                go = size > 0;
                if (!go) break;
                --size;
                const int32_t e = stk[size * WALK_THREADS];
                parent = e & 0x0fffffff;
                mnr = (int) ((uint32_t) e >> 28);
#pragma unroll
                for (int ax = 0; ax < D; ++ax) prel[ax] >>= 1;
            }
        }
    }

    if (ROWS) {
        for (int l = 0; l < w.nlevels; ++l)
            if (item < w.lay.ecap[l]) w.l3_cs[w.lay.base[l] + item] = lvl[l * WALK_THREADS];
        w.l1_cs[item] = e1.n + blk_len;
        if (w.close_cs) w.close_cs[item] = ec.n;
        const bool ovf = e1.n > w.K1 || n3 > w.K3 || (w.close_cs && ec.n > w.Kc);
        w.overflow[item] = ovf ? 1 : 0;
        if (ovf) w.ovf_list[atomicAdd(w.ovf_count, 1)] = item;
        if (ovf && w.dbg_counts) {          // BT_TRAV_STATS: why items overflow
            if (e1.n > w.K1) atomicAdd(w.dbg_counts + 0, 1);
            if (n3 > w.K3) atomicAdd(w.dbg_counts + 1, 1);
            if (w.close_cs && ec.n > w.Kc) atomicAdd(w.dbg_counts + 2, 1);
            if (slot >= 0) atomicAdd(w.dbg_counts + 3, 1);
        }
    }
}

// rows -> final places, one wave per tile of 64 items: a lane reads entry j of its own
// item (the wave reads 256 contiguous bytes) and writes it to the item's CSR segment
__global__ __launch_bounds__(256) void rows_to_csr_v2_kernel(const int32_t *d_nitems,
        const uint8_t *overflow, const int32_t *rows, int K, const int32_t *starts,
        const int32_t *blk_reserved /* unused */, int32_t *lists)
{
    const int32_t item = blockIdx.x * 256 + threadIdx.x;
    const int32_t nitems = *d_nitems;
    const bool active = item < nitems && !overflow[item];
    int32_t s = 0, n = 0;
    if (active) { s = starts[item]; n = starts[item + 1] - s; if (n > K) n = K; }
    int nmax = n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(nmax, off, 64);
        nmax = o > nmax ? o : nmax;
    }
    const int32_t *row = rows + (int64_t) (item >> 6) * 64 * K + (item & 63);
    for (int j = 0; j < nmax; ++j)
        if (j < n) lists[(int64_t) s + j] = row[(int64_t) j * 64];
}

// list 3: rows -> per-level lists (cursors start at the item's per-level starts)
__global__ __launch_bounds__(256) void l3_scatter_v2_kernel(const int32_t *d_nitems, L3Layout lay,
        int nlevels, const uint8_t *overflow, const int32_t *row3, const uint8_t *row3lev, int K3,
        const int32_t *l3_item_starts, int32_t *l3_lists)
{
    const int32_t item = blockIdx.x * 256 + threadIdx.x;
    if (item >= *d_nitems || overflow[item]) return;
    int32_t *cur = s_walk_lds + threadIdx.x;
    int n = 0;
    for (int l = 0; l < nlevels; ++l) {
        int32_t s = 0;
        if (item < lay.ecap[l]) {
            s = l3_item_starts[lay.base[l] + item];
            n += l3_item_starts[lay.base[l] + item + 1] - s;
        }
        cur[l * WALK_THREADS] = s;
    }
    const int32_t *row = row3 + (int64_t) (item >> 6) * 64 * K3 + (item & 63);
    const uint8_t *rl = row3lev + (int64_t) (item >> 6) * 64 * K3 + (item & 63);
    for (int j = 0; j < n; ++j) {
        const int lev = rl[(int64_t) j * 64];
        l3_lists[cur[lev * WALK_THREADS]++] = row[(int64_t) j * 64];
    }
}

// list-3 starts per (level, target box) from the per-(level, item) starts: a box whose
// first item lies beyond the level's columns has no entries there and starts where
// the level ends
__global__ __launch_bounds__(256) void l3_box_starts_v2_kernel(int64_t nflat_box, int32_t ntb,
        L3Layout lay, int nlevels, const int32_t *first_item, const int32_t *item_starts,
        int32_t *box_starts)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i > nflat_box) return;
    if (i == nflat_box) { box_starts[i] = item_starts[lay.base[nlevels]]; return; }
    const int lev = (int) (i / ntb);
    const int32_t tbn = (int32_t) (i % ntb);
    const int32_t item = first_item[tbn];
    const int32_t col = item < lay.ecap[lev] ? item : lay.ecap[lev];
    box_starts[i] = item_starts[lay.base[lev] + col];
}
