// Fast list generation for trees whose boxes are numbered level-major with
// every level in depth-first (Morton) order -- which is what the tree builder
// produces and what check_pack_kernel verifies for
// any input tree.  Included by bt_trav.hip inside its anonymous namespace.
//
// Instead of walking from the root for every box (traversal.py:398-550), the
// candidates of box b come from its PARENT's colleague list:
//   children of (colleagues(parent) U {parent})  =  colleagues(b)  U  list2(b)
// (traversal.py:556-601 already uses this for list 2).  The reference's float
// predicates are evaluated unchanged on the candidates, so the lists are
// identical whenever "child adjacent => parent adjacent" holds for the float
// test -- rounding errors are ~1e-16*root_extent against a tolerance of one box
// radius (traversal.py:257-276), and the parity tests compare both paths with
// the walk-based oracle.
//
// Output order.  The walks emit in depth-first order (within one level that is
// ascending box id for trees numbered in Morton order, not for level-restricted ones); across levels (list 1) entries are ordered by the
// depth-first preorder rank of the box, computed once per tree.

struct FastTree {
    const int32_t *dfs_rank;      // [nboxes] preorder rank
    const int32_t *box_of_rank;   // [nboxes]
    const int32_t *subtree_size;  // [nboxes] boxes in the subtree (incl. the box)
    const int32_t *src_prefix;    // [nboxes+1] #source boxes with rank < r
    const int32_t *src_by_rank;   // source boxes in depth-first order
};

// A target box WITH children (target extents) has every source box of its own
// subtree in list 1 -- a contiguous range of preorder ranks.  Those entries are
// copied by copy_rank_blocks_kernel instead of being walked by one thread (the
// root's walk alone was 80 ms at 1e8 sources).
struct BlockJobs {
    int32_t *count;               // number of jobs (atomic)
    int32_t *dst;                 // offset into the list-1 array
    int32_t *src;                 // offset into src_by_rank
    int32_t *len;
    int32_t *dbg;                 // BT_TRAV_STATS: counters of the ordering kernels (or null)
};

template <class T, int D>
struct SourceRankFlag {
    const Node<T, D> *nodes;
    const int32_t *box_of_rank;
    __device__ int32_t operator()(int64_t r) const
    {
        return ((nodes[box_of_rank[r]].lf >> 8) & BT_BOX_IS_SOURCE_BOX) ? 1 : 0;
    }
};

template <class T, int D>
__global__ __launch_bounds__(256) void compact_sources_by_rank_kernel(SourceRankFlag<T, D> f,
        int32_t nboxes, const int32_t *prefix, int32_t *src_by_rank)
{
    const int32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nboxes) return;
    if (f(r)) src_by_rank[prefix[r]] = f.box_of_rank[r];
}

// one workgroup per job at a time (the largest job, the root's, is a 12 MB copy at 1e8
// points); the job count stays on the device
// own-subtree blocks: slot j belongs to target box j (len 0: none).  The ordering kernels
// used to append jobs through one atomic counter: 4*10^5 appends to one address took about
// as long as ordering the lists.
__global__ __launch_bounds__(256) void copy_rank_blocks_kernel(int32_t nj, const int32_t *dst,
        const int32_t *src, const int32_t *len, const int32_t *src_by_rank, int32_t *lists)
{
    // a wave per slot: most slots are empty
    const int lane = threadIdx.x & 63;
    for (int32_t j = blockIdx.x * 4 + (threadIdx.x >> 6); j < nj; j += gridDim.x * 4) {
        const int32_t n = len[j];
        if (n <= 0) continue;
        const int32_t *in = src_by_rank + src[j];
        int32_t *out = lists + dst[j];
        for (int32_t i = lane; i < n; i += 64) out[i] = in[i];
    }
}

// ---- structure check --------------------------------------------------------------

template <int D>
__device__ __forceinline__ int find_slot(const int32_t *child, int64_t aligned, int32_t parent,
                                         int32_t b)
{
    constexpr int C = 1 << D;
#pragma unroll
    for (int m = 0; m < C; ++m)
        if (child[(int64_t) m * aligned + parent] == b) return m;
    return -1;
}

// bad[0]: numbering / flags not as the fast kernels need them.
// bad[2]: (information) some target box has source boxes below it.
// bad[1]: some box centre is not exactly "parent centre +/- root_extent / 2^(level+1)"
//         (the lattice kernels of bt_trav_v2.hpp then must not be used).
// The structure check and pack_nodes_kernel<.., true> (bt_geom.hpp) in one pass over the
// boxes: both read the child table, the centres, levels and flags of a box.
template <class T, int D>
__global__ __launch_bounds__(256) void check_pack_kernel(int32_t nboxes, int64_t aligned,
        const int32_t *parent, const int32_t *child, const uint8_t *levels, const uint8_t *flags,
        const T *centers, T root_extent, int *bad, Node<T, D> *nodes, int32_t *child_t,
        uint64_t *child8 /* Kids of bt_trav_v2.hpp; bad[3]: some box's children are not numbered
                            consecutively in slot order (the lattice kernels must not be used) */)
{
    constexpr int C = 1 << D;
    const int32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    int32_t ch[C];
#pragma unroll
    for (int m = 0; m < C; ++m) ch[m] = child[(int64_t) m * aligned + b];
    T cen[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) cen[ax] = centers[(int64_t) ax * aligned + b];
    const uint8_t lev = levels[b], fl = flags[b];

    // ---- the packed records ------------------------------------------------------------
    Node<T, D> n;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) n.c[ax] = cen[ax];
    n.lf = (uint32_t) lev | ((uint32_t) fl << 8);
    nodes[b] = n;
    bool ok = true, geom_ok = true, consecutive = true;
    int32_t row[C];
    uint32_t first = 0, nkids = 0, masks = 0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = ch[m];
        uint32_t e = (uint32_t) c;
        if (c > 0 && c < nboxes) {
            const uint8_t cf = flags[c];
            if (cf & BT_BOX_IS_SOURCE_BOX) { e |= CH_SRC; masks |= 0x100u << m; }
            if (cf & BT_BOX_HAS_SOURCE_CHILD_BOXES) { e |= CH_HSC; masks |= 0x10000u << m; }
        }
        row[m] = (int32_t) e;
        if (c != 0) {
            ok = ok && c > b && c < nboxes && parent[c] == b;
            if (nkids == 0) first = (uint32_t) c;
            else consecutive = consecutive && (uint32_t) c == first + nkids;
            ++nkids;
            masks |= 1u << m;
        }
    }
    child8[b] = (uint64_t) first | ((uint64_t) masks << 32);
    if (!consecutive) atomicExch(bad + 3, 1);
    // (the row in as few stores as its alignment allows: 2^d ints at a multiple of 2^d ints)
    if constexpr (C == 8) {
        int4 *r = reinterpret_cast<int4 *>(child_t + (int64_t) b * C);
        r[0] = make_int4(row[0], row[1], row[2], row[3]);
        r[1] = make_int4(row[4], row[5], row[6], row[7]);
    } else if constexpr (C == 4) {
        *reinterpret_cast<int4 *>(child_t + (int64_t) b * C) = make_int4(row[0], row[1], row[2], row[3]);
    } else {
        *reinterpret_cast<int2 *>(child_t + (int64_t) b * C) = make_int2(row[0], row[1]);
    }

    // ---- the structure check ---------------------------------------------------------------
    // level-major numbering; the order within a level is free (level-restricted trees append
    // force-split children at the end of their level: every order-sensitive step works on
    // depth-first ranks); anything with sources below must be reachable through the flags;
    // centres as tree_build_kernels.py:698-705 evaluates them
    if (b == 0) {
        ok = ok && lev == 0;
    } else {
        const int32_t p = parent[b];
        ok = ok && p >= 0 && p < b;
        if (ok) {
            ok = ok && lev == levels[p] + 1;
            const int slot = find_slot<D>(child, aligned, p, b);
            ok = ok && slot >= 0;
            ok = ok && levels[b - 1] <= lev;
            if (fl & (BT_BOX_IS_SOURCE_BOX | BT_BOX_HAS_SOURCE_CHILD_BOXES))
                ok = ok && (flags[p] & BT_BOX_HAS_SOURCE_CHILD_BOXES);
            if (ok && lev < 63) {
                const T radius = (root_extent * 1 / (T) (1ull << (1 + (int) lev)));
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const bool has_bit = (slot >> (D - 1 - ax)) & 1;
                    const T pc = centers[(int64_t) ax * aligned + p];
                    const T want = has_bit ? pc + radius : pc - radius;
                    geom_ok = geom_ok && want == cen[ax];
                }
            } else {
                geom_ok = false;
            }
        }
    }
    if (!ok) atomicExch(bad, 1);
    if (!geom_ok) atomicExch(bad + 1, 1);
    if ((fl & BT_BOX_IS_TARGET_BOX) && (fl & BT_BOX_HAS_SOURCE_CHILD_BOXES)
            && __hip_atomic_load(bad + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(bad + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- depth-first preorder rank ---------------------------------------------------

template <int D>
__global__ __launch_bounds__(256) void subtree_size_kernel(int32_t b0, int32_t nb, int64_t aligned,
        const int32_t *child, int32_t *size)
{
    constexpr int C = 1 << D;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const int32_t b = b0 + i;
    int32_t s = 1;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = child[(int64_t) m * aligned + b];
        if (c) s += size[c];
    }
    size[b] = s;
}

template <int D>
__global__ __launch_bounds__(256) void dfs_rank_kernel(int32_t b0, int32_t nb, int64_t aligned,
        const int32_t *child, const int32_t *size, int32_t *rank, int32_t *box_of_rank)
{
    constexpr int C = 1 << D;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const int32_t p = b0 + i;          // parent whose children get their ranks
    if (p == 0) { rank[0] = 0; box_of_rank[0] = 0; }
    int32_t run = ((p == 0) ? 0 : rank[p]) + 1;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = child[(int64_t) m * aligned + p];
        if (c) {
            rank[c] = run;
            box_of_rank[run] = c;
            run += size[c];
        }
    }
}

// ---- colleagues + list 2, one level (top-down) ---------------------------------------

struct CollL2Out {
    int32_t *coll_cnt_or_starts;   // count: [nb] counts ; fill: global starts [nboxes+1]
    int32_t *l2_cnt_or_starts;
    int32_t *coll_lists;
    int32_t *l2_lists;
};

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void coll_l2_kernel(TravArgs<T, D> a, int32_t b0, int32_t nb,
                                                      CollL2Out o)
{
    constexpr int C = 1 << D;
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    const int32_t g = t / C;
    const int m = t % C;
    if (g >= nb) return;                 // whole groups drop out together
    const int32_t b = b0 + g;
    const int lane = threadIdx.x & 63;
    const int gshift = lane / C * C;
    const uint64_t lanes_below = (1ull << m) - 1ull;

    T center[D];
    load_center(a, b, center);
    const int level = box_level(a, b);
    const int32_t p = a.parent[b];
    const bool ttp = (box_flags(a, b) & (BT_BOX_HAS_TARGET_CHILD_BOXES | BT_BOX_IS_TARGET_BOX))
        && (!a.target_mask || a.target_mask[b]);    // list 2 only for wanted boxes
    const int32_t ps = a.coll_starts[p];
    const int32_t n = a.coll_starts[p + 1] - ps;
    int ins = 0;                         // depth-first position of p among its colleagues
    const int32_t prank = a.dfs_rank[p];     // (ranks, not box numbers: the boxes of a level
                                             // need not be numbered in depth-first order)
    for (int i = 0; i < n; ++i) ins += (a.dfs_rank[a.coll_lists[ps + i]] < prank) ? 1 : 0;

    int32_t ccnt = 0, lcnt = 0;
    int32_t ccur = 0, lcur = 0;
    if (FILL) { ccur = o.coll_cnt_or_starts[b]; lcur = o.l2_cnt_or_starts[b]; }

    // batches of UNR parent-colleagues: all child-id loads of a batch are issued
    // before the node loads that depend on them, so a batch costs two memory round
    // trips instead of 2*UNR (the loop is latency-bound otherwise)
    constexpr int UNR = 4;
    for (int i0 = 0; i0 <= n; i0 += UNR) {
        int32_t cs[UNR], chs[UNR];
        uint32_t lfs[UNR];
        T ccs[UNR][D];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u;
            cs[u] = (i > n) ? 0 : (i < ins) ? a.coll_lists[ps + i]
                  : (i == ins ? p : a.coll_lists[ps + i - 1]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) chs[u] = (i0 + u <= n) ? child_of<D>(a, cs[u], m) : 0;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const Node<T, D> nd = a.nodes[chs[u]];      // box 0 when there is no child: harmless
#pragma unroll
            for (int ax = 0; ax < D; ++ax) ccs[u][ax] = nd.c[ax];
            lfs[u] = nd.lf;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (i0 + u > n) break;                      // uniform within the group
            const int32_t c = cs[u], ch = chs[u];
            bool is_coll = false, is_l2 = false;
            if (ch != 0 && ch != b) {
                const bool a_or_o = adj_nbhd<T, D>(a.root_extent, center, level, (T) a.nway,
                                                   ccs[u], (int) (lfs[u] & 0xffu));
                is_coll = a_or_o;                                   // traversal.py:429-442
                is_l2 = !a_or_o && c != p && ttp;                   // traversal.py:588-597
            }
            if (FILL) {
                const uint64_t bc = (__ballot(is_coll) >> gshift) & ((1ull << C) - 1);
                const uint64_t bl = (__ballot(is_l2) >> gshift) & ((1ull << C) - 1);
                if (is_coll) o.coll_lists[ccur + __popcll(bc & lanes_below)] = ch;
                if (is_l2) o.l2_lists[lcur + __popcll(bl & lanes_below)] = ch;
                ccur += __popcll(bc);
                lcur += __popcll(bl);
            } else {
                ccnt += is_coll;
                lcnt += is_l2;
            }
        }
    }
    if (!FILL) {
#pragma unroll
        for (int off = C / 2; off > 0; off >>= 1) {
            ccnt += __shfl_xor(ccnt, off, C);
            lcnt += __shfl_xor(lcnt, off, C);
        }
        if (m == 0) { o.coll_cnt_or_starts[g] = ccnt; o.l2_cnt_or_starts[g] = lcnt; }
    }
}

// Single-pass variant of coll_l2_kernel (well_sep_is_n_away == 1): no count pass.
// Colleagues go to fixed-stride rows (a box has at most 3^d - 1 of them) that the
// next level reads directly; list 2 goes to a per-level scratch row (at most
// 6^d - 3^d entries) and is compacted into the CSR afterwards.  Every adjacency
// test and every node load happens once instead of twice.
template <class T, int D>
__global__ __launch_bounds__(256) void coll_l2_rows_kernel(TravArgs<T, D> a, int32_t b0, int32_t nb,
        int32_t *coll_rows, int32_t *coll_cnt, int32_t *l2_rows, int32_t *l2_cnt,
        int32_t *srccoll_rows, int32_t *srccoll_cnt, int32_t *coll_ins)
{
    constexpr int C = 1 << D;
    constexpr int P = (D == 1 ? 3 : D == 2 ? 9 : 27) - 1;
    constexpr int S = (D == 1 ? 6 : D == 2 ? 36 : 216) - (P + 1);
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    const int32_t g = t / C;
    const int m = t % C;
    if (g >= nb) return;                 // whole groups drop out together
    const int32_t b = b0 + g;
    const int lane = threadIdx.x & 63;
    const int gshift = lane / C * C;
    const uint64_t lanes_below = (1ull << m) - 1ull;

    T center[D];
    load_center(a, b, center);
    const int level = box_level(a, b);
    const int32_t p = a.parent[b];
    const bool ttp = (box_flags(a, b) & (BT_BOX_HAS_TARGET_CHILD_BOXES | BT_BOX_IS_TARGET_BOX))
        && (!a.target_mask || a.target_mask[b]);    // list 2 only for wanted boxes
    const int32_t *prow = coll_rows + (int64_t) p * P;
    const int32_t n = coll_cnt[p];
    // depth-first position of p among its colleagues: recorded when p's own row was
    // made (the number of colleagues emitted before the walk reached p itself)
    const int ins = coll_ins[p];

    int32_t *crow = coll_rows + (int64_t) b * P;
    int32_t *lrow = l2_rows + (int64_t) g * S;
    int32_t *srow = srccoll_rows + (int64_t) b * P;   // colleagues that are source boxes
    int32_t ccur = 0, lcur = 0, scur = 0;
    constexpr int UNR = 4;
    for (int i0 = 0; i0 <= n; i0 += UNR) {
        int32_t cs[UNR], chs[UNR];
        uint32_t lfs[UNR];
        T ccs[UNR][D];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u;
            cs[u] = (i > n) ? 0 : (i < ins) ? prow[i] : (i == ins ? p : prow[i - 1]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) chs[u] = (i0 + u <= n) ? child_of<D>(a, cs[u], m) : 0;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const Node<T, D> nd = a.nodes[chs[u]];
#pragma unroll
            for (int ax = 0; ax < D; ++ax) ccs[u][ax] = nd.c[ax];
            lfs[u] = nd.lf;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (i0 + u > n) break;
            const int32_t c = cs[u], ch = chs[u];
            bool is_coll = false, is_l2 = false;
            if (ch != 0 && ch != b) {
                const bool a_or_o = adj_nbhd<T, D>(a.root_extent, center, level, (T) 1,
                                                   ccs[u], (int) (lfs[u] & 0xffu));
                is_coll = a_or_o;                                   // traversal.py:429-442
                is_l2 = !a_or_o && c != p && ttp;                   // traversal.py:588-597
            }
            const bool is_src = is_coll && ((lfs[u] >> 8) & BT_BOX_IS_SOURCE_BOX);
            const uint64_t bc = (__ballot(is_coll) >> gshift) & ((1ull << C) - 1);
            const uint64_t bl = (__ballot(is_l2) >> gshift) & ((1ull << C) - 1);
            const uint64_t bs = (__ballot(is_src) >> gshift) & ((1ull << C) - 1);
            // the candidates come in depth-first order; b itself is one of them
            if (ch == b) coll_ins[b] = ccur + __popcll(bc & lanes_below);
            if (is_coll) crow[ccur + __popcll(bc & lanes_below)] = ch;
            if (is_l2) lrow[lcur + __popcll(bl & lanes_below)] = ch;
            if (is_src) srow[scur + __popcll(bs & lanes_below)] = ch;
            ccur += __popcll(bc);
            lcur += __popcll(bl);
            scur += __popcll(bs);
        }
    }
    if (m == 0) { coll_cnt[b] = ccur; l2_cnt[g] = lcur; srccoll_cnt[b] = scur; }
}

// rows[r][0..count) -> lists[base + starts[r] ...); LANES lanes per row (measured at
// 5*10^6 boxes: colleague rows, <= 26 entries, are fastest with 8; list-2 rows, ~40
// entries on average, with 16)
template <int LANES>
__global__ __launch_bounds__(256) void compact_strided_rows_kernel(int64_t nrows, int stride,
        const int32_t *rows, const int32_t *starts, int32_t base, int32_t *lists)
{
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LANES;
    const int lane = (int) (gid % LANES);
    if (r >= nrows) return;
    const int32_t s = starts[r], e = starts[r + 1];
    const int32_t *row = rows + r * stride;
    for (int32_t k = lane; k < e - s; k += LANES) lists[(int64_t) base + s + k] = row[k];
}

// (a variant with one WAVE per parent box -- candidates loaded once for all children -- measured
// slower on c3, 6.0 against 4.9 ms, and was dropped with its switch in round 6)

__global__ __launch_bounds__(256) void add_base_kernel(int32_t n, const int32_t *rel, int32_t base,
                                                       int32_t *dst)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = rel[i] + base;
}

__global__ __launch_bounds__(256) void fill_i32_kernel(int32_t n, int32_t value, int32_t *dst)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = value;
}

__global__ __launch_bounds__(256) void gather_starts_kernel(int32_t n, const int32_t *boxes,
        const int32_t *starts_by_box, int32_t total, int32_t *out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = starts_by_box[boxes[i]];
    if (i == n) out[n] = total;
}

// ---- list 1 from the ancestors' colleague lists ----------------------------------------

__device__ __forceinline__ void sort_i32_inplace(int32_t *v, int n)
{
    if (n <= 48) {
        for (int i = 1; i < n; ++i) {
            const int32_t x = v[i];
            int j = i - 1;
            while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; }
            v[j + 1] = x;
        }
        return;
    }
    // heapsort (in place, no recursion) for the rare long lists
    for (int start = n / 2 - 1; start >= 0; --start) {
        int root = start;
        const int32_t x = v[root];
        while (true) {
            int ch = 2 * root + 1;
            if (ch >= n) break;
            if (ch + 1 < n && v[ch + 1] > v[ch]) ++ch;
            if (v[ch] <= x) break;
            v[root] = v[ch];
            root = ch;
        }
        v[root] = x;
    }
    for (int end = n - 1; end > 0; --end) {
        const int32_t x = v[end];
        v[end] = v[0];
        int root = 0;
        while (true) {
            int ch = 2 * root + 1;
            if (ch >= end) break;
            if (ch + 1 < end && v[ch + 1] > v[ch]) ++ch;
            if (v[ch] <= x) break;
            v[root] = v[ch];
            root = ch;
        }
        v[root] = x;
    }
}

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void list1_fast_kernel(TravArgs<T, D> a, FastTree ft, int32_t n,
        int32_t *counts_or_starts, int32_t *lists)
{
    constexpr int C = 1 << D;
    const int32_t tbn = blockIdx.x * 256 + threadIdx.x;
    if (tbn >= n) return;
    const int32_t b = a.target_boxes[tbn];
    T center[D];
    load_center(a, b, center);
    const int level = box_level(a, b);

    int32_t cnt = 0;
    int32_t *out = FILL ? lists + counts_or_starts[tbn] : nullptr;
    auto emit = [&](int32_t u) {
        if (FILL) out[cnt] = ft.dfs_rank[u];
        ++cnt;
    };

    if (box_flags(a, 0) & BT_BOX_IS_SOURCE_BOX) emit(0);              // traversal.py:489-495

    // finer boxes: descend like traversal.py:501-547, but only inside box u
    auto descend = [&](int32_t u) {
        Walk w(s_walk_lds + threadIdx.x);
        w.init(u);
        while (w.go) {
            const int32_t wb = child_of<D>(a, w.parent, w.mnr);
            if (wb) {
                T wc[D];
                load_center(a, wb, wc);
                if (adj<T, D>(a.root_extent, center, level, wc, box_level(a, wb))) {
                    const uint8_t wf = box_flags(a, wb);
                    if (wf & BT_BOX_IS_SOURCE_BOX) emit(wb);
                    if (wf & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
                        w.push(wb);
                        continue;
                    }
                }
            }
            w.template advance<C>();
        }
    };

    if (level == 0) {
        // the root as a target box (only with target extents): everything below it
        if (box_flags(a, 0) & BT_BOX_HAS_SOURCE_CHILD_BOXES) descend(0);
    }

    // ancestors (and b): their colleagues hold every adjacent box of that level
    int32_t anc = b;
    for (int k = level; k >= 1; --k, anc = a.parent[anc]) {
        const int32_t s0 = a.coll_starts[anc], s1 = a.coll_starts[anc + 1];
        for (int32_t i = s0 - 1; i < s1; ++i) {
            const int32_t u = (i < s0) ? anc : a.coll_lists[i];
            const uint8_t fl = box_flags(a, u);
            if (!(fl & (BT_BOX_IS_SOURCE_BOX | BT_BOX_HAS_SOURCE_CHILD_BOXES))) continue;
            if (u != anc) {
                T uc[D];
                load_center(a, u, uc);
                if (!adj<T, D>(a.root_extent, center, level, uc, k)) continue;
            }
            if (fl & BT_BOX_IS_SOURCE_BOX) emit(u);
            if (k == level && (fl & BT_BOX_HAS_SOURCE_CHILD_BOXES)) descend(u);
        }
    }

    if (!FILL) {
        counts_or_starts[tbn] = cnt;
    } else {
        sort_i32_inplace(out, cnt);                 // depth-first preorder
        for (int i = 0; i < cnt; ++i) out[i] = ft.box_of_rank[out[i]];
    }
}


// ---- source-box colleagues (filtered CSR) -------------------------------------------------

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void filter_source_colleagues_kernel(TravArgs<T, D> a,
        int32_t nboxes, int32_t *cnt_or_starts, int32_t *lists)
{
    const int32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    const int32_t s0 = a.coll_starts[b], s1 = a.coll_starts[b + 1];
    int32_t n = 0;
    int32_t *out = FILL ? lists + cnt_or_starts[b] : nullptr;
    for (int32_t i = s0; i < s1; ++i) {
        const int32_t u = a.coll_lists[i];
        if (box_flags(a, u) & BT_BOX_IS_SOURCE_BOX) {
            if (FILL) out[n] = u;
            ++n;
        }
    }
    if (!FILL) cnt_or_starts[b] = n;
}

// ---- lists 1 and 3 (+ close) in one walk per target box ----------------------------------

struct L1Emit {
    static constexpr bool active = true;
    const int32_t *dfs_rank;
    int32_t *out;       // null in the count pass
    int32_t n;
    __device__ __forceinline__ void operator()(int32_t u)
    {
        if (out) out[n] = dfs_rank[u];
        ++n;
    }
};

// Work items.  A target box far above the leaf level has long lists and one thread
// walking all of its colleagues' subtrees is a serial tail (2.7 ms at 1e8 points);
// such "heavy" boxes are split into one item per colleague (+ one for the box
// itself and its coarser neighbours).  Items of a box are consecutive, so the
// per-item output segments concatenate to the box's lists in colleague order.
constexpr int SLOT_ALL = -1, SLOT_SELF = -2;

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void make_items_kernel(TravArgs<T, D> a, int32_t ntb,
        int heavy_max_level, int32_t *cnt_or_first, int32_t *item_tbn, int32_t *item_slot)
{
    const int32_t tbn = blockIdx.x * 256 + threadIdx.x;
    if (tbn >= ntb) return;
    const int32_t b = a.target_boxes[tbn];
    const int32_t ncoll = a.coll_starts[b + 1] - a.coll_starts[b];
    const bool heavy = box_level(a, b) <= heavy_max_level && ncoll > 1;
    if (!FILL) {
        cnt_or_first[tbn] = heavy ? ncoll + 1 : 1;
    } else {
        int32_t it = cnt_or_first[tbn];
        if (!heavy) {
            item_tbn[it] = tbn; item_slot[it] = SLOT_ALL;
        } else {
            item_tbn[it] = tbn; item_slot[it] = SLOT_SELF;
            for (int32_t k = 0; k < ncoll; ++k) { item_tbn[it + 1 + k] = tbn; item_slot[it + 1 + k] = k; }
        }
    }
}

template <class T, int D, bool FILL>
__global__ __launch_bounds__(256) void list13_kernel(TravArgs<T, D> a, FastTree ft,
        const int32_t *lcoll_starts, const int32_t *lcoll_lists,
        const int32_t *item_tbn, const int32_t *item_slot, int32_t nitems, int nlevels,
        int walk_cap, int32_t *l1_cs, int32_t *l1_lists,
        int32_t *l3_cs, int32_t *l3_lists, int32_t *close_cs, int32_t *close_lists,
        int lcoll_stride)
{
    // lcoll_stride > 0: lcoll_lists are fixed-stride rows, lcoll_starts their counts
    const int32_t item = blockIdx.x * 256 + threadIdx.x;
    if (item >= nitems) return;
    const int32_t tbn = item_tbn[item];
    const int slot = item_slot[item];
    const int32_t b = a.target_boxes[tbn];
    T center[D];
    load_center(a, b, center);
    const int level = box_level(a, b);

    L1Emit e1{ft.dfs_rank, FILL ? l1_lists + l1_cs[item] : nullptr, 0};

    int32_t blk_len = 0;
    if (slot < 0) {
        if (box_flags(a, 0) & BT_BOX_IS_SOURCE_BOX) e1(0);       // traversal.py:489-495
        // b itself; the sources inside b (target boxes with children: extents only) are
        // a contiguous block of preorder ranks: space is reserved here, the entries are
        // written by l1_finalize32_kernel / copy_rank_blocks_kernel
        if (level >= 1 && (box_flags(a, b) & BT_BOX_IS_SOURCE_BOX)) e1(b);
        // coarser levels: source-box colleagues of the ancestors, and the ancestors
        if (level >= 2) {
            int32_t anc = a.parent[b];
            for (int k = level - 1; k >= 1; --k, anc = a.parent[anc]) {
                if (box_flags(a, anc) & BT_BOX_IS_SOURCE_BOX) e1(anc);
                const int64_t s0 = lcoll_stride ? (int64_t) anc * lcoll_stride : lcoll_starts[anc];
                const int64_t s1 = lcoll_stride ? s0 + lcoll_starts[anc] : lcoll_starts[anc + 1];
                for (int64_t i = s0; i < s1; ++i) {
                    const int32_t u = lcoll_lists[i];
                    T uc[D];
                    load_center(a, u, uc);
                    if (adj<T, D>(a.root_extent, center, level, uc, k)) e1(u);
                }
            }
        }
    }

    // the box's LAST item reserves the space of the own-subtree block at the end of
    // the box's list-1 segment
    {
        const int32_t ncoll = a.coll_starts[b + 1] - a.coll_starts[b];
        const bool last_item = slot == SLOT_ALL || slot == ncoll - 1;
        if (last_item && (box_flags(a, b) & BT_BOX_HAS_SOURCE_CHILD_BOXES)) {
            const int32_t my_rank = ft.dfs_rank[b];
            blk_len = ft.src_prefix[my_rank + ft.subtree_size[b]] - ft.src_prefix[my_rank + 1];
        }
    }

    // colleagues and everything below them: list 3 walk, which also yields the
    // list-1 boxes at the colleagues' level and finer
    const int32_t cfirst = slot >= 0 ? slot : 0;
    const int32_t ccount = slot >= 0 ? 1 : (slot == SLOT_SELF ? 0 : -1);
    int32_t *lvl = s_walk_lds + walk_cap * WALK_THREADS + threadIdx.x;
    if (!FILL) {
        L3CountMain em{lvl};
        for (int l = 0; l < nlevels; ++l) lvl[l * WALK_THREADS] = 0;
        CountEmit ec;
        gen_list3<T, D>(a, tbn, em, ec, e1, cfirst, ccount);
        for (int l = 0; l < nlevels; ++l)
            l3_cs[(int64_t) l * nitems + item] = lvl[l * WALK_THREADS];
        if (close_cs) close_cs[item] = ec.n;
        l1_cs[item] = e1.n + blk_len;
    } else {
        L3WriteMain em{l3_lists, lvl};
        for (int l = 0; l < nlevels; ++l)
            lvl[l * WALK_THREADS] = l3_cs[(int64_t) l * nitems + item];
        WriteEmit ec{close_lists ? close_lists + close_cs[item] : nullptr};
        CountEmit dummy;
        if (close_lists) gen_list3<T, D>(a, tbn, em, ec, e1, cfirst, ccount);
        else gen_list3<T, D>(a, tbn, em, dummy, e1, cfirst, ccount);
    }
}

// List 1 is ordered with 16 lanes per target box: lists of up to 64 entries
// (nearly all of them) are ordered by counting in registers; longer ones go to a wave
// or a workgroup.
constexpr int L1_BLOCK_MAX = 8192;
constexpr int L1_WAVE_MAX = 1024;

struct TierIs {
    const uint8_t *tier;
    uint8_t value;
    __device__ int32_t operator()(int64_t i) const { return tier[i] == value ? 1 : 0; }
};

__global__ __launch_bounds__(256) void compact_tier_kernel(int32_t n, TierIs pr, const int32_t *pos,
                                                           int32_t *out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && pr(i)) out[pos[i]] = i;
}

// all-reduce over an aligned group of 16 lanes (one DPP row): quad permutes, then the
// mirrors of half a row and of the row
__device__ __forceinline__ int group16_max(int v)
{
    int o;
    o = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false); v = o > v ? o : v;    // quad_perm [1,0,3,2]
    o = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false); v = o > v ? o : v;    // quad_perm [2,3,0,1]
    o = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false); v = o > v ? o : v;   // row_half_mirror
    o = __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false); v = o > v ? o : v;   // row_mirror
    return v;
}

__device__ __forceinline__ int group16_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return v;
}

template <class T, int D>
__global__ __launch_bounds__(256) void l1_finalize32_kernel(TravArgs<T, D> a, FastTree ft,
        int32_t ntb, const int32_t *l1_starts, int32_t *l1_lists, BlockJobs jobs,
        uint8_t *tier /* [ntb], zeroed: 1 = wave kernel, 2 = workgroup kernel */,
        int32_t *tier_present /* [2], zeroed */,
        int boxes /* the lists hold box numbers (ordered by their depth-first rank, one lookup
                     per entry) instead of ranks (ordered, then mapped back: a second lookup) */)
{
    // sixteen lanes per box, up to four entries per lane: four boxes' chains of dependent
    // loads (box -> rank -> row -> box_of_rank) are in flight per wave
    __shared__ int32_t s_l1[16 * 64];
    const int32_t gid = blockIdx.x * 256 + threadIdx.x;
    const int32_t tbn = gid >> 4;
    const int lane = gid & 15;
    if (tbn >= ntb) return;                          // whole 16-groups drop out together
    const int32_t b = a.target_boxes[tbn];
    int32_t *out = l1_lists + l1_starts[tbn];
    const int32_t n_all = l1_starts[tbn + 1] - l1_starts[tbn];
    // the entries are fetched before the length of the own block is known (three dependent
    // loads away: box -> rank -> source prefix): slots behind the entries hold junk and are
    // masked below
    int32_t x_early[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 16 * k;
        x_early[k] = (i < n_all && i < 64) ? out[i] : INT32_MAX;
    }
    int32_t blk_len = 0, blk_src = 0;
    const int32_t my_rank = ft.dfs_rank[b];
    if (box_flags(a, b) & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
        blk_src = ft.src_prefix[my_rank + 1];
        blk_len = ft.src_prefix[my_rank + ft.subtree_size[b]] - blk_src;
    }
    const int32_t n = n_all - blk_len;
    if (n <= 64) {
        // The list arrives as a short unordered head (the box itself, coarser boxes) followed
        // by a run that is already ascending: the colleagues come in depth-first order and
        // each is walked depth first.  The head is found by looking for the last descent;
        // every entry counts the head entries below it (c trips instead of n), a run entry
        // adds its place in the run, a head entry a binary search in the run.  A list that
        // is not of that form has c = n: plain ranking by counting.
        int32_t *sv = s_l1 + (threadIdx.x >> 4) * 64;
        int32_t x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 16 * k;
            x[k] = i < n ? (boxes ? ft.dfs_rank[x_early[k]] : x_early[k]) : INT32_MAX;
            sv[i] = x[k];
        }
        __builtin_amdgcn_wave_barrier();
        int last_desc = 0, below = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 16 * k;
            if (i >= 1 && i < n && sv[i - 1] > x[k]) last_desc = i;
            below += (i < n && x[k] <= my_rank) ? 1 : 0;       // entries before the own block
        }
        const int c_found = group16_max(last_desc);
        const int k = group16_sum(below);
        const int c = c_found <= 16 ? c_found : n;             // long heads: rank everything
        if (jobs.dbg && lane == 0) {
            atomicAdd(jobs.dbg + 0, 1); atomicAdd(jobs.dbg + 8, c); atomicAdd(jobs.dbg + 9, n);
            if (c_found <= 16) atomicAdd(jobs.dbg + 1, 1);
        }
        int cnt[4] = {0, 0, 0, 0};
        for (int t = 0; t < c; ++t) {
            const int32_t pv = sv[t];                          // one address: a broadcast read
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt[q] += (pv < x[q]) ? 1 : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 16 * q;
            if (i >= n) continue;
            int pos;
            if (i >= c) {
                pos = (i - c) + cnt[q];
            } else {
                int lo = c, hi = n;                            // run entries below x[q]
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sv[mid] < x[q]) lo = mid + 1; else hi = mid;
                }
                pos = cnt[q] + (lo - c);
            }
            out[pos < k ? pos : pos + blk_len] = boxes ? x_early[q] : ft.box_of_rank[x[q]];
        }
        if (blk_len > 0 && lane == 0) {
            const int32_t j = tbn;                  // one slot per target box: no shared counter
            jobs.dst[j] = l1_starts[tbn] + k;
            jobs.src[j] = blk_src;
            jobs.len[j] = blk_len;
        }
        return;
    }
    if (lane != 0) return;
    // longer rows are left to a wave (LDS ranking) or a workgroup (LDS sort); they are
    // collected by a scan over `tier`, not by atomic appends (a million appends to one
    // counter serialise)
    if (jobs.dbg) {
        atomicAdd(jobs.dbg + (n <= L1_WAVE_MAX ? 2 : n <= L1_BLOCK_MAX ? 3 : 4), 1);
        if (n > L1_BLOCK_MAX) atomicMax(jobs.dbg + 5, n);
    }
    if (n <= L1_WAVE_MAX && tier) { tier[tbn] = 1; tier_present[0] = 1; return; }
    if (n <= L1_BLOCK_MAX && tier) { tier[tbn] = 2; tier_present[1] = 1; return; }
    if (boxes)
        for (int32_t i = 0; i < n; ++i) out[i] = ft.dfs_rank[out[i]];
    sort_i32_inplace(out, n);
    int32_t k = n;
    if (blk_len > 0) {
        k = 0;
        while (k < n && out[k] <= my_rank) ++k;
        for (int32_t i = n - 1; i >= k; --i) out[i + blk_len] = ft.box_of_rank[out[i]];
        const int32_t j = tbn;                  // one slot per target box: no shared counter
        jobs.dst[j] = l1_starts[tbn] + k;
        jobs.src[j] = blk_src;
        jobs.len[j] = blk_len;
    }
    for (int32_t i = 0; i < k; ++i) out[i] = ft.box_of_rank[out[i]];
}

// list 1 of listed target boxes with 32 < entries <= L1_WAVE_MAX: one wave per box
// stages the ranks in LDS and every lane finds the final position of its entries by
// counting the smaller ones (ranks are distinct)
template <class T, int D>
__global__ __launch_bounds__(256) void l1_finalize_wave_kernel(TravArgs<T, D> a, FastTree ft,
        const int32_t *mid_list, const int32_t *d_nmid, const int32_t *l1_starts, int32_t *l1_lists,
        BlockJobs jobs, int boxes /* see l1_finalize32_kernel */)
{
    __shared__ int32_t s_all[4][L1_WAVE_MAX];
    __shared__ int32_t s_box_all[4][L1_WAVE_MAX];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int32_t *s_v = s_all[w];
    int32_t *s_box = s_box_all[w];
    const int32_t nmid = *d_nmid;
  for (int32_t idx = blockIdx.x * 4 + w; idx < nmid; idx += gridDim.x * 4) {
    const int32_t tbn = mid_list[idx];
    const int32_t b = a.target_boxes[tbn];
    int32_t *out = l1_lists + l1_starts[tbn];
    const int32_t n_all = l1_starts[tbn + 1] - l1_starts[tbn];
    // (fetched before the length of the own block is known, as in l1_finalize32_kernel)
    for (int i = lane; i < n_all && i < L1_WAVE_MAX; i += 64) {
        const int32_t v = out[i];
        s_box[i] = v;
        s_v[i] = v;
    }
    int32_t blk_len = 0, blk_src = 0;
    const int32_t my_rank = ft.dfs_rank[b];
    if (box_flags(a, b) & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
        blk_src = ft.src_prefix[my_rank + 1];
        blk_len = ft.src_prefix[my_rank + ft.subtree_size[b]] - blk_src;
    }
    const int32_t n = n_all - blk_len;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (boxes) {
        for (int i = lane; i < n; i += 64) s_v[i] = ft.dfs_rank[s_box[i]];
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    // the unordered head of the list (see l1_finalize32_kernel): up to the last descent
    int c = 0;
    for (int base = ((n - 1) >> 6) << 6; base >= 0; base -= 64) {
        const int i = base + lane;
        const uint64_t bal = __ballot(i >= 1 && i < n && s_v[i - 1] > s_v[i]);
        if (bal) { c = base + 63 - __clzll(bal); break; }
    }
    if (jobs.dbg && lane == 0) atomicAdd(jobs.dbg + (c <= 64 ? 6 : 7), 1);
    if (c <= 64) {
        // merge: an entry's place = head entries below it + (run entries below it: its own
        // index in the run, or a binary search for a head entry)
        int32_t k0 = 0;                              // entries before the own-subtree block
        for (int base = 0; base < n; base += 64)
            k0 += __popcll(__ballot(base + lane < n && s_v[base + lane] <= my_rank));
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            if (i >= n) continue;
            const int32_t xv = s_v[i];
            int cnt = 0;
            for (int t = 0; t < c; ++t) cnt += (s_v[t] < xv) ? 1 : 0;
            int pos;
            if (i >= c) {
                pos = (i - c) + cnt;
            } else {
                int lo = c, hi = n;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_v[mid] < xv) lo = mid + 1; else hi = mid;
                }
                pos = cnt + (lo - c);
            }
            out[pos < k0 ? pos : pos + blk_len] = boxes ? s_box[i] : ft.box_of_rank[xv];
        }
        if (blk_len > 0 && lane == 0) {
            const int32_t j = tbn;                  // one slot per target box: no shared counter
            jobs.dst[j] = l1_starts[tbn] + k0;
            jobs.src[j] = blk_src;
            jobs.len[j] = blk_len;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        continue;
    }
    // a long unordered head: bitonic sort of the (distinct) ranks in LDS by the wave,
    // O(n log^2 n / 64) steps (ranking every entry against all others took 3.4 ms on the
    // 10^5 mid-size lists of the 10^8 + 10^7 extent workload)
    int m = 64;
    while (m < n) m <<= 1;
    for (int i = n + lane; i < m; i += 64) s_v[i] = INT32_MAX;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < m; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const int32_t x = s_v[i], y = s_v[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { s_v[i] = y; s_v[l] = x; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
    int32_t k0 = 0;                                  // entries before the own-subtree block
    for (int base = 0; base < n; base += 64)
        k0 += __popcll(__ballot(base + lane < n && s_v[base + lane] <= my_rank));
    for (int i = lane; i < n; i += 64)
        out[i < k0 ? i : i + blk_len] = ft.box_of_rank[s_v[i]];
    if (blk_len > 0 && lane == 0) {
        const int32_t j = tbn;                  // one slot per target box: no shared counter
        jobs.dst[j] = l1_starts[tbn] + k0;
        jobs.src[j] = blk_src;
        jobs.len[j] = blk_len;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
}

// list 1 of one listed target box (32 < entries <= L1_BLOCK_MAX): bitonic sort of the
// ranks in LDS by one workgroup, then the same placement as above
template <class T, int D>
__global__ __launch_bounds__(256) void l1_finalize_block_kernel(TravArgs<T, D> a, FastTree ft,
        const int32_t *big_list, const int32_t *d_nbig, const int32_t *l1_starts,
        int32_t *l1_lists, BlockJobs jobs, int boxes /* see l1_finalize32_kernel */)
{
    __shared__ int32_t s_v[L1_BLOCK_MAX];
    const int32_t nbig = *d_nbig;
  for (int32_t jb = blockIdx.x; jb < nbig; jb += gridDim.x) {
    __syncthreads();
    const int32_t tbn = big_list[jb];
    const int32_t b = a.target_boxes[tbn];
    int32_t *out = l1_lists + l1_starts[tbn];
    const int32_t n_all = l1_starts[tbn + 1] - l1_starts[tbn];
    int32_t blk_len = 0, blk_src = 0;
    const int32_t my_rank = ft.dfs_rank[b];
    if (box_flags(a, b) & BT_BOX_HAS_SOURCE_CHILD_BOXES) {
        blk_src = ft.src_prefix[my_rank + 1];
        blk_len = ft.src_prefix[my_rank + ft.subtree_size[b]] - blk_src;
    }
    const int32_t n = n_all - blk_len;
    int m = 64;
    while (m < n) m <<= 1;
    for (int i = threadIdx.x; i < m; i += 256)
        s_v[i] = (i < n) ? (boxes ? ft.dfs_rank[out[i]] : out[i]) : INT32_MAX;
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < m; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const int32_t x = s_v[i], y = s_v[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { s_v[i] = y; s_v[l] = x; }
                }
            }
            __syncthreads();
        }
    }
    // entries before the own-subtree block: ranks <= my_rank (binary search, sorted)
    int32_t lo = 0, hi = n;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (s_v[mid] <= my_rank) lo = mid + 1; else hi = mid;
    }
    const int32_t k0 = lo;
    for (int i = threadIdx.x; i < n; i += 256)
        out[i < k0 ? i : i + blk_len] = ft.box_of_rank[s_v[i]];
    if (blk_len > 0 && threadIdx.x == 0) {
        const int32_t j = tbn;                  // one slot per target box: no shared counter
        jobs.dst[j] = l1_starts[tbn] + k0;
        jobs.src[j] = blk_src;
        jobs.len[j] = blk_len;
    }
  }
}

// l3 bookkeeping per (level, target box) from the per-(level, item) starts
__global__ __launch_bounds__(256) void l3_box_starts_kernel(int64_t nflat_box, int32_t ntb,
        int32_t nitems, int nlevels, const int32_t *first_item, const int32_t *item_starts,
        int32_t *box_starts)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i > nflat_box) return;
    if (i == nflat_box) { box_starts[i] = item_starts[(int64_t) nlevels * nitems]; return; }
    const int64_t lev = i / ntb, tbn = i % ntb;
    box_starts[i] = item_starts[lev * nitems + first_item[tbn]];
}
