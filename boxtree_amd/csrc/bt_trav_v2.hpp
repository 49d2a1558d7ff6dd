// Lattice formulation of the parent-colleague list kernels (bt_trav_fast.hpp) for
// well_sep_is_n_away == 1.  Included by bt_trav.hip inside its anonymous namespace.
//
// A tree whose box centres are exactly "parent centre +/- root_extent / 2^(level+1)"
// (check_pack_kernel verifies it bit for bit) is a set of cells of a dyadic
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
//    axis: (offset of the colleague's cell from the box's cell) + 1.  Lists 1 and 4 want the
//    colleagues of a box's ancestors that are source boxes -- few of them: most colleagues
//    of a box above the leaves have children.  3D trees of fewer than 2^25 boxes mark them
//    with bit 25 of the entry (V2_SRC_BIT) and keep one 32-bit mask per box of the row's
//    entries that carry it; other trees keep a second family of rows with the source boxes;
//  * child_t[box][2^d] holds "child | flags": bit 28 = the child is a source box,
//    bit 29 = it has source child boxes (tree walks read nothing else per step);
//  * ICell{c[3], lf}: integer cell coordinates at the box's own level.
// Boxes < 2^26 and levels <= 29 are required (the host falls back to the float
// kernels otherwise).

constexpr int V2_CODE_SHIFT = 26;
constexpr uint32_t V2_ID_MASK = (1u << V2_CODE_SHIFT) - 1u;
constexpr uint32_t V2_CODE_SELF = 0x15u;            // offsets (0,0,0): 01 01 01
constexpr uint32_t V2_SRC_BIT = 1u << (V2_CODE_SHIFT - 1);    // one row family: the entry is a source box
constexpr int64_t V2_ONE_FAMILY_MAX_BOXES = (int64_t) 1 << (V2_CODE_SHIFT - 1);

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

// the 2^d packed child words of a box with one or two vector loads
template <int C>
__device__ __forceinline__ void v2_load_children(const int32_t *child_t, int32_t box, uint32_t (&cw)[C])
{
    const int32_t *src = child_t + (int64_t) box * C;
    if constexpr (C == 8) {
        const int4 lo = *reinterpret_cast<const int4 *>(src);
        const int4 hi = *reinterpret_cast<const int4 *>(src + 4);
        cw[0] = lo.x; cw[1] = lo.y; cw[2] = lo.z; cw[3] = lo.w;
        cw[4] = hi.x; cw[5] = hi.y; cw[6] = hi.z; cw[7] = hi.w;
    } else if constexpr (C == 4) {
        const int4 lo = *reinterpret_cast<const int4 *>(src);
        cw[0] = lo.x; cw[1] = lo.y; cw[2] = lo.z; cw[3] = lo.w;
    } else {
        const int2 lo = *reinterpret_cast<const int2 *>(src);
        cw[0] = lo.x; cw[1] = lo.y;
    }
}

// The children of a box in 8 bytes.  Every tree this path accepts numbers the children of a
// box consecutively, in slot order (check_pack_kernel verifies it), so a box's row of the
// child table is its first child plus three masks over the slots: present | is a source box
// << 8 | has source child boxes << 16.  A quarter of the bytes of the 2^d packed words -- the
// walks read a record per box they visit, at random --, two registers instead of 2^d, and
// the masks are what the kernels want anyway.
struct Kids {
    uint32_t first, masks;
    __device__ __forceinline__ uint32_t present() const { return masks & 0xffu; }
    __device__ __forceinline__ uint32_t source() const { return (masks >> 8) & 0xffu; }
    __device__ __forceinline__ uint32_t has_src_children() const { return (masks >> 16) & 0xffu; }
    // number of the child in slot m (which must be present)
    __device__ __forceinline__ int32_t id(int m) const
    {
        return (int32_t) (first + (uint32_t) __popc(masks & 0xffu & ((1u << m) - 1u)));
    }
};

__device__ __forceinline__ Kids v2_load_kids(const uint64_t *child8, int32_t box)
{
    const uint64_t w = child8[box];
    return Kids{(uint32_t) w, (uint32_t) (w >> 32)};
}

// ---- per-tree tables: integer cells, depth-first ranks -------------------------------

struct DfsLevel {
    int32_t b0, nb;                    // the parents: boxes of one level
    int64_t aligned;
    const int32_t *child, *size;
    const uint8_t *levels, *flags;
    int32_t *rank, *box_of_rank;
    ICell *cells;
};

template <int D>
__device__ __forceinline__ void dfs_rank_cells_block(const DfsLevel &a, int32_t blk)
{
    constexpr int C = 1 << D;
    const int32_t b0 = a.b0;
    const int64_t aligned = a.aligned;
    const int32_t *child = a.child, *size = a.size;
    const uint8_t *levels = a.levels, *flags = a.flags;
    int32_t *rank = a.rank, *box_of_rank = a.box_of_rank;
    ICell *cells = a.cells;
    const int32_t i = blk * 256 + threadIdx.x;
    if (i >= a.nb) return;
    const int32_t p = b0 + i;          // parent whose children get their ranks
    ICell pc;
    if (p == 0) {
        rank[0] = 0; box_of_rank[0] = 0;
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
            ICell cc;
            cc.c[0] = cc.c[1] = cc.c[2] = 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) cc.c[ax] = 2u * pc.c[ax] + (uint32_t) v2_mbit<D>(m, ax);
            cc.lf = (uint32_t) levels[c] | ((uint32_t) flags[c] << 8);
            cells[c] = cc;
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void dfs_rank_cells_kernel(DfsLevel a)
{
    dfs_rank_cells_block<D>(a, (int32_t) blockIdx.x);
}

// ---- colleague rows (+ list-2 counts), one level, top-down ---------------------------

template <int D>
struct V2Rows {
    const uint64_t *child8;        // Kids
    const int32_t *parent;
    const uint8_t *flags;
    const int8_t *target_mask;
    int32_t *coll_rows, *coll_cnt, *coll_ins;
    int32_t *srccoll_rows, *srccoll_cnt;   // second family (one family: no rows, and
                                           // srccoll_cnt holds the masks of the source entries)
    uint32_t id_mask;              // the box number of a row entry
    int32_t *l2_cnt;               // [nboxes]
    const int32_t *l2_starts;      // fill pass: [nboxes + 1]
    int32_t *l2_lists;
    int l2_stage;                  // fill pass: a child's List 2 is put together in LDS (3D)
};

// One group of lanes per PARENT and one lane per candidate (a colleague of the parent, or
// the parent itself, at its depth-first place).  FILL = false: build the children's rows,
// count their lists 2.  FILL = true: write list 2 (all levels in one launch; rows are
// complete by then).  The children of one parent share
// their candidates, so the group loads each candidate's 2^d child words once (one or two
// vector loads per lane) and then serves the parent's children one after the other
// without another load.  Whether child slot m of a candidate at offset d (per axis -1, 0,
// +1) is adjacent to the child in slot sb does not depend on the data: per axis
// rel = 2 d + m_ax - sb_ax must lie in [-1, 1], i.e. d = 0 always, d = -1 only for
// (m_ax, sb_ax) = (1, 0), d = +1 only for (0, 1) -- so "present", "source" and
// "adjacent" are bit masks over the child slots, the lanes agree on write positions with
// one packed prefix sum per child, and each lane writes its (at most 2^d) entries.
// An earlier form (a group of 2^d lanes per BOX, a lane per child slot, one 4-byte load per
// candidate and lane, two ballots per candidate) took 10-14 % longer at 10^8 sphere points;
// both forms are bound by vector-ALU issue (SQ_INSTS_VALU x 4 cycles per wave64
// instruction / 1024 SIMDs accounts for the whole duration), not by memory: see LAB_NOTES.md
// section 4.
// four list entries stored at once at any 4-byte boundary (global_store_dwordx4)
struct __attribute__((packed, aligned(4))) PackedI4 { int32_t x, y, z, w; };

template <int D> struct V3Lanes { static constexpr int N = D == 3 ? 32 : D == 2 ? 16 : 4; };

// inclusive prefix sum over aligned groups of LANES (4, 16 or 32) lanes with DPP moves
// (row_shr within a row of 16 lanes, row_bcast:15 from one row into the next): five
// dependent adds, where shuffles through ds_bpermute cost an LDS round trip each
template <int LANES>
__device__ __forceinline__ uint32_t v3_group_scan(uint32_t v)
{
    static_assert(LANES == 16 || LANES == 32, "groups of 4 lanes: see the specialisation");
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false);
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false);
    if (LANES == 32)
        v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false);
    return v;
}

template <>
__device__ __forceinline__ uint32_t v3_group_scan<4>(uint32_t v)
{
    const int j = threadIdx.x & 3;
    uint32_t up = (uint32_t) __shfl_up((int) v, 1, 4);
    if (j >= 1) v += up;
    up = (uint32_t) __shfl_up((int) v, 2, 4);
    if (j >= 2) v += up;
    return v;
}

// (M(m) - S(sb)) << V2_CODE_SHIFT, modulo 2^32 (see coll_rows_v3_kernel)
template <int D>
__device__ __forceinline__ constexpr uint32_t v3_code_delta(int m, int sb)
{
    uint32_t r = 0;
    for (int ax = 0; ax < D; ++ax) {
        r += (uint32_t) ((m >> (D - 1 - ax)) & 1) << (2 * ax);
        r -= (uint32_t) ((sb >> (D - 1 - ax)) & 1) << (2 * ax);
    }
    return r << V2_CODE_SHIFT;
}

template <int D>
__device__ __forceinline__ constexpr uint32_t v3_set_mask(int ax)
{
    // child slots m whose bit for axis `ax` is set (v2_mbit)
    uint32_t r = 0;
    for (int m = 0; m < (1 << D); ++m) if ((m >> (D - 1 - ax)) & 1) r |= 1u << m;
    return r;
}

// parents[0 .. np): boxes that have children; rows / lists are made for their children in
// [b_lo, b_hi)
template <int D, bool FILL, bool ONE /* one row family: V2_SRC_BIT in the entries */>
__device__ __forceinline__ void coll_rows_v3_block(const V2Rows<D> &t, const int32_t *parents, int32_t np,
        int32_t b_lo, int32_t b_hi, int32_t blk)
{
    constexpr int C = 1 << D;
    constexpr int P = V2Dims<D>::P;
    constexpr int LANES = V3Lanes<D>::N;
    constexpr uint32_t FULL = (1u << C) - 1u;
    const int32_t tid = blk * 256 + threadIdx.x;
    const int32_t g = tid / LANES;
    const int j = tid % LANES;
    if (g >= np) return;                 // whole groups drop out together
    const int32_t p = parents[g];

    // the parent's own children (every lane reads the same word)
    const Kids pk = v2_load_kids(t.child8, p);

    // Everything the children's turns need is loaded here, before the first store: loads
    // and stores complete in order on this hardware, so a load issued after a turn's stores
    // would wait for them (a full round trip per child).
    const int n = t.coll_cnt[p];
    const int ins = t.coll_ins[p];
    const uint32_t e_at = (uint32_t) t.coll_rows[(int64_t) p * P + (j < P ? j : P - 1)];
    const uint32_t e_before = (uint32_t) t.coll_rows[(int64_t) p * P + (j >= 1 && j <= P ? j - 1 : 0)];
    bool want[C];               // FILL: list 2 not empty; else: the box wants a list 2
    int32_t lstart[C];
#pragma unroll
    for (int sb = 0; sb < C; ++sb) {
        const int32_t b = ((pk.masks >> sb) & 1u) ? pk.id(sb) : 0;
        const bool mine = b != 0 && b >= b_lo && b < b_hi;
        want[sb] = false; lstart[sb] = 0;
        if (FILL) {
            if (mine) {
                lstart[sb] = t.l2_starts[b];
                want[sb] = t.l2_starts[b + 1] != lstart[sb];
            }
        } else if (mine) {
            const uint8_t fl = t.flags[b];
            want[sb] = (fl & (BT_BOX_HAS_TARGET_CHILD_BOXES | BT_BOX_IS_TARGET_BOX))
                && (!t.target_mask || t.target_mask[b]);        // list 2 only for wanted boxes
        }
    }
    // candidate j: the parent's colleagues in depth-first order, the parent itself at
    // its own place `ins` (n + 1 candidates)
    const bool valid = j <= n;
    const bool self = valid && j == ins;
    uint32_t e = 0;
    if (valid)
        e = self ? ((uint32_t) p | (V2_CODE_SELF << V2_CODE_SHIFT)) : (j < ins ? e_at : e_before);
    const uint32_t q = e & t.id_mask;

    // the candidate's children (the parent's own for the parent itself)
    Kids ck{0u, 0u};
    if (valid && !self) ck = v2_load_kids(t.child8, (int32_t) q);
    else if (self) ck = pk;
    const uint32_t present = ck.present(), source = ck.source();
    // per axis: the child slots of this candidate adjacent to a child whose own bit on
    // that axis is 0 (adj0) or 1 (adj1)
    uint32_t adj0[D], adj1[D];
    // The code of child slot m of this candidate in the row of the parent's child sb is, per
    // axis, rel + 1 = (2 d + 1) + m_ax - sb_ax, a value in [0, 2] for adjacent pairs: the
    // fields never borrow from each other, so the whole code is one integer sum
    // off + M(m) - S(sb) with compile-time M and S.
    uint32_t off = 0;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const int d = v2_off(e, ax);
        const uint32_t set = v3_set_mask<D>(ax);
        adj0[ax] = d == 0 ? FULL : d < 0 ? set : 0u;
        adj1[ax] = d == 0 ? FULL : d > 0 ? (FULL & ~set) : 0u;
        off += (uint32_t) (2 * d + 1) << (2 * ax);
    }
    off <<= V2_CODE_SHIFT;
    // (numbers of absent children are never used: every use is under `present`)
    uint32_t chid[C], plain[C];
#pragma unroll
    for (int m = 0; m < C; ++m) {
        plain[m] = (uint32_t) ck.id(m);
        chid[m] = plain[m] + off + (ONE ? ((source >> m) & 1u) * V2_SRC_BIT : 0u);
    }

#pragma unroll
    for (int sb = 0; sb < C; ++sb) {
        const int32_t b = ((pk.masks >> sb) & 1u) ? pk.id(sb) : 0;
        if (b == 0 || b < b_lo || b >= b_hi) continue;          // group-uniform
        if (FILL && !want[sb]) continue;                        // group-uniform
        const int32_t lbase = lstart[sb];
        const bool ttp = want[sb];
        uint32_t adjacent = FULL;
#pragma unroll
        for (int ax = 0; ax < D; ++ax) adjacent &= v2_mbit<D>(sb, ax) ? adj1[ax] : adj0[ax];
        uint32_t cm = present & adjacent;                       // traversal.py:429-442
        if (self) cm &= ~(1u << sb);                            // the box itself
        const uint32_t lm = ttp ? (present & ~adjacent & FULL) : 0u;   // traversal.py:588-597
        const uint32_t sm = ONE ? 0u : (cm & source);

        // positions: an inclusive scan of (colleagues | source colleagues << 10 | list 2 << 20)
        const uint32_t packed = FILL ? (uint32_t) __popc(lm)
                                     : ((uint32_t) __popc(cm) | ((uint32_t) __popc(sm) << 10)
                                        | ((uint32_t) __popc(lm) << 20));
        const uint32_t incl = v3_group_scan<LANES>(packed);
        const uint32_t excl = incl - packed;

        // Each lane writes its own entries (at most 2^d stores, a few lanes each).  Putting
        // a row together in LDS first and writing it with one store was measured slower:
        // what this kernel spends its time on is the chain of dependent steps per child --
        // masks, scan, addresses --, and stores are not part of it.
        if (!FILL) {
            int32_t *crow = t.coll_rows + (int64_t) b * P;
            int32_t *srow = ONE ? nullptr : t.srccoll_rows + (int64_t) b * P;
            int pc = (int) (excl & 0x3ffu), ps = (int) ((excl >> 10) & 0x3ffu);
            if (self) t.coll_ins[b] = pc + __popc(cm & ((1u << sb) - 1u));
            if constexpr (LANES == 32) {
                // The group puts the two rows together in LDS and writes each with ONE store:
                // what the direct form costs is its store instructions -- up to sixteen per
                // child, a few lanes each, every one a transaction of its own on the way to
                // the L2 --, not the bytes (run without the source rows' stores: -0.17 ms at
                // 10^8 sphere points, -0.3 ms at 1.25*10^8 uniform ones).
                __shared__ int32_t s_stage[256 / 32][2][32];
                int32_t *lc = &s_stage[threadIdx.x / 32][0][0], *ls = lc + 32;
#pragma unroll
                for (int m = 0; m < C; ++m) {
                    if ((cm >> m) & 1u) {
                        const int32_t entry = (int32_t) (chid[m] + v3_code_delta<D>(m, sb));
                        lc[pc++] = entry;
                        if (!ONE && ((sm >> m) & 1u)) ls[ps++] = entry;
                    }
                }
                if (j == LANES - 1) lc[31] = (int32_t) incl;       // (P = 27: slot 31 is free)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const uint32_t tot = (uint32_t) lc[31];
                const int nc = (int) (tot & 0x3ffu), ns = (int) ((tot >> 10) & 0x3ffu);
                const int32_t vc = lc[j], vs = ONE ? 0 : ls[j];
                if (j < nc) crow[j] = vc;
                if (!ONE && j < ns) srow[j] = vs;
                if (ONE) {
                    // which entries of the row are source boxes: the group's half of a ballot
                    const uint64_t bal = __ballot(j < nc && ((uint32_t) vc & V2_SRC_BIT));
                    if (j == LANES - 1)
                        t.srccoll_cnt[b] = (int32_t) (uint32_t) (bal >> (32 * ((threadIdx.x >> 5) & 1)));
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
                for (int m = 0; m < C; ++m) {
                    if ((cm >> m) & 1u) {
                        const int32_t entry = (int32_t) (chid[m] + v3_code_delta<D>(m, sb));
                        crow[pc++] = entry;
                        if (!ONE && ((sm >> m) & 1u)) srow[ps++] = entry;
                    }
                }
            }
            if (j == LANES - 1) {               // the last lane's inclusive sums are the totals
                t.coll_cnt[b] = (int32_t) (incl & 0x3ffu);
                if (!ONE) t.srccoll_cnt[b] = (int32_t) ((incl >> 10) & 0x3ffu);
                t.l2_cnt[b] = (int32_t) (incl >> 20);
            }
        } else if (LANES == 32 && t.l2_stage) {
            // the group's List 2 of this child through LDS: every lane drops its entries at
            // their places, the group writes the row out in 16-byte pieces
            __shared__ int32_t s_l2[256 / 32][192];
            int32_t *lr = s_l2[threadIdx.x / 32];
            int pos = (int) excl;
#pragma unroll
            for (int m = 0; m < C; ++m)
                if ((lm >> m) & 1u) lr[pos++] = (int32_t) plain[m];
            if (j == LANES - 1) lr[191] = (int32_t) incl;          // (at most 189 entries)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int tot = lr[191];
            int32_t *dst = t.l2_lists + lbase;
            for (int k = j * 4; k < tot; k += 128) {
                if (k + 4 <= tot) {
                    *reinterpret_cast<PackedI4 *>(dst + k) = PackedI4{lr[k], lr[k + 1], lr[k + 2], lr[k + 3]};
                } else {
                    for (int q = k; q < tot; ++q) dst[q] = lr[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            int32_t pl = lbase + (int32_t) excl;
            // A candidate that touches the box nowhere gives all its children (19 of the 27
            // candidates of a box in a uniform octree, 152 of its 189 entries): two 16-byte
            // stores instead of eight predicated 4-byte ones.
            if (C == 8 && lm == FULL) {
                PackedI4 *dst = reinterpret_cast<PackedI4 *>(t.l2_lists + pl);
                dst[0] = PackedI4{(int32_t) plain[0], (int32_t) plain[1], (int32_t) plain[2], (int32_t) plain[3]};
                dst[1] = PackedI4{(int32_t) plain[C > 4 ? 4 : 0], (int32_t) plain[C > 5 ? 5 : 0],
                                  (int32_t) plain[C > 6 ? 6 : 0], (int32_t) plain[C > 7 ? 7 : 0]};
                // (the same for half a candidate -- the four children beyond a face across the
                // first axis -- was measured: -0.04 ms at 1.25*10^8 uniform points, +0.05 ms at
                // 10^8 sphere-surface points, where few halves are full; not kept)
            } else {
#pragma unroll
                for (int m = 0; m < C; ++m)
                    if ((lm >> m) & 1u) t.l2_lists[pl++] = (int32_t) plain[m];
            }
        }
    }
}

template <int D, bool FILL>
__global__ __launch_bounds__(256) void coll_rows_v3_kernel(V2Rows<D> t, const int32_t *parents, int32_t np,
        int32_t b_lo, int32_t b_hi)
{
    static_assert(FILL, "the rows are built by level_tables_kernel");
    coll_rows_v3_block<D, FILL, false>(t, parents, np, b_lo, b_hi, (int32_t) blockIdx.x);
}

// One launch per level for the two top-down tables: the first dfs_blocks workgroups give the
// children of level `lev` their depth-first ranks and cells, the others build the colleague
// rows of level `lev + 1` from those of level `lev`.  Neither reads what the other writes.
template <int D, bool ONE>
__global__ __launch_bounds__(256) void level_tables_kernel(DfsLevel d, int32_t dfs_blocks, V2Rows<D> t,
        const int32_t *parents, int32_t np, int32_t b_lo, int32_t b_hi)
{
    if ((int32_t) blockIdx.x < dfs_blocks) dfs_rank_cells_block<D>(d, (int32_t) blockIdx.x);
    else coll_rows_v3_block<D, false, ONE>(t, parents, np, b_lo, b_hi, (int32_t) blockIdx.x - dfs_blocks);
}

// List 4 without extents (traversal.py:931-1146, no close lists), thread per target (or
// target-parent) box: the source colleagues of the box's ancestors that touch the box's
// parent but not the box.  A row entry carries the colleague's offset from the ancestor's
// cell, so both tests are integer arithmetic on the box's own cell -- no centre is loaded
// (the float form loads three coordinates per candidate).  With k = level(box) - level(
// ancestor): the box relative to the colleague is (low k bits of its cell) - offset * 2^k
// per axis, and boxes touch iff that is in [-1, 2^k] on every axis.
template <int D, bool FILL>
__global__ __launch_bounds__(256) void list4_lattice_kernel(int32_t n, const int32_t *ttp_boxes,
        const ICell *cells, const int32_t *parent, const int32_t *srccoll_rows,
        const int32_t *srccoll_cnt, int stride, uint32_t id_mask,
        uint32_t src_bit /* one row family: entries without it are skipped; else 0 */,
        int32_t *counts_or_starts, int32_t *lists)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t tgt = ttp_boxes[i];
    const ICell tc = cells[tgt];
    const int tl = (int) (tc.lf & 0xffu);
    int32_t cnt = 0;
    int32_t *out = FILL ? lists + counts_or_starts[i] : nullptr;
    PackedI4 buf{0, 0, 0, 0};
    int32_t cur = tgt;
    for (int k = 1; k < tl; ++k) {                      // ancestors on levels tl-1 .. 1
        cur = parent[cur];
        int64_t lt[D], lp[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            lt[ax] = (int64_t) (tc.c[ax] & ((1u << k) - 1u));
            lp[ax] = (int64_t) ((tc.c[ax] >> 1) & ((1u << (k - 1)) - 1u));
        }
        const int32_t *row = srccoll_rows + (int64_t) cur * stride;
        // (one family: srccoll_cnt is the mask of the row's source entries)
        uint32_t smask = src_bit ? (uint32_t) srccoll_cnt[cur] : 0u;
        const int nrow = src_bit ? __popc(smask) : srccoll_cnt[cur];
        for (int jj = 0; jj < nrow; ++jj) {
            const int j = src_bit ? __builtin_ctz(smask) : jj;
            smask &= smask - 1u;
            const uint32_t e = (uint32_t) row[j];
            bool adj_box = true, adj_parent = true;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const int64_t off = v2_off(e, ax);
                const int64_t rt = lt[ax] - off * ((int64_t) 1 << k);
                const int64_t rp = lp[ax] - off * ((int64_t) 1 << (k - 1));
                adj_box = adj_box && rt >= -1 && rt <= ((int64_t) 1 << k);
                adj_parent = adj_parent && rp >= -1 && rp <= ((int64_t) 1 << (k - 1));
            }
            if (!adj_box && adj_parent) {
                if (FILL) {
                    // entries leave four at a time (one 16-byte store per lane instead of four
                    // lane-strided 4-byte ones)
                    const int32_t id = (int32_t) (e & id_mask);
                    const int q = cnt & 3;
                    if (q == 0) buf.x = id; else if (q == 1) buf.y = id; else if (q == 2) buf.z = id;
                    else { buf.w = id; *reinterpret_cast<PackedI4 *>(out + cnt - 3) = buf; }
                }
                ++cnt;
            }
        }
    }
    if (FILL) {
        const int q = cnt & 3, c0 = cnt - q;
        if (q >= 1) out[c0] = buf.x;
        if (q >= 2) out[c0 + 1] = buf.y;
        if (q >= 3) out[c0 + 2] = buf.z;
    }
    if (!FILL) counts_or_starts[i] = cnt;
}

// colleague CSR from the rows (codes stripped); LANES lanes per row
template <int LANES>
__global__ __launch_bounds__(256) void compact_coll_rows_v2_kernel(int64_t nrows, int stride,
        uint32_t id_mask, const int32_t *rows, const int32_t *starts, int32_t *lists)
{
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LANES;
    const int lane = (int) (gid % LANES);
    if (r >= nrows) return;
    const int32_t s = starts[r], e = starts[r + 1];
    const int32_t *row = rows + r * stride;
    for (int32_t k = lane; k < e - s; k += LANES)
        lists[(int64_t) s + k] = (int32_t) ((uint32_t) row[k] & id_mask);
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
    const uint64_t *child8;            // Kids
    const int32_t *coll_rows, *coll_cnt, *srccoll_rows, *srccoll_cnt;
    uint32_t id_mask, src_bit;         // of the row entries (src_bit: one row family, else 0)
    const int32_t *item_tbn, *item_slot;
    const int32_t *d_nitems;           // actual item count (device)
    int32_t items_cap;                 // columns of the item arrays (>= the item count)
    L3Layout lay;
    int nlevels, walk_cap;
    int with_blocks;                   // own-subtree blocks exist (extents)
    // rows
    int32_t *row1, *row3 /* box | level << V2_CODE_SHIFT */, *rowc;
    int K1, K3, Kc;
    int32_t *l1_cs, *l3_cs, *close_cs; // counts (ROWS) / starts (!ROWS)
    uint8_t *overflow;                 // [items_cap]
    int32_t *ovf_count, *ovf_list;
    // list-3 entries beyond an item's row: one chunk of SPILL_CHUNK entries per such item,
    // handed out by SPILL_SHARDS counters (one word for all would serialise); an item whose
    // chunk cannot be had, or does not suffice, is walked again (overflow list)
    int32_t *spill3;                   // [SPILL_SHARDS * spill_per_shard][SPILL_CHUNK], or null
    int32_t *spill_count;              // [SPILL_SHARDS * 16] (a cache line apart), zeroed
    int32_t *spill_idx;                // [2][items_cap] chunk of the item's list 3 / list 1, or -1
    int32_t spill_per_shard;
    int32_t *dbg_counts;               // optional [4]
    // final places (!ROWS)
    int32_t *l1_lists, *l3_lists, *close_lists;
};

constexpr int SPILL_CHUNK = 1024;
constexpr int SPILL_SHARDS = 64;

// Scratch rows, G = entries per group: entry j of item t of a tile of 64 items sits at
// (j / G) * 64 * G + t * G + j % G.  G = 1 is "entry j of the 64 items contiguous"; with
// G = 4 a lane keeps three entries in registers and writes four with one 16-byte store --
// the walk of a volume-filling cloud spends a third of its time on row stores, and what they
// cost is their number, not their bytes (run with every other store left out: 2.6 -> 1.9 ms at
// 1.25*10^8 points).  The extent-tree walk has no registers to spare: G = 1 there.
template <bool TEXT> struct RowGroup { static constexpr int G = TEXT ? 1 : 4; };

template <bool ROWS, int G>
struct V2Emit {                        // list 1 / close list of one item
    int32_t *base;                     // ROWS: the item's slot 0 of group 0; else: final place
    int cap, n;
    int32_t p0, p1, p2;                // G = 4: entries of the group being filled
    __device__ __forceinline__ void operator()(int32_t v)
    {
        if (!ROWS) {
            base[n] = v;
        } else if (n < cap) {
            if (G == 1) {
                base[(int64_t) n * 64] = v;
            } else {
                const int q = n & 3;
                if (q == 0) p0 = v; else if (q == 1) p1 = v; else if (q == 2) p2 = v;
                else *reinterpret_cast<PackedI4 *>(base + (int64_t) (n >> 2) * 256) = PackedI4{p0, p1, p2, v};
            }
        }
        ++n;
    }
    __device__ __forceinline__ void flush()            // the entries of a group left open
    {
        if (!ROWS || G == 1) return;
        const int m = n < cap ? n : cap, q = m & 3;
        int32_t *g = base + (int64_t) (m >> 2) * 256;
        if (q >= 1) g[0] = p0;
        if (q >= 2) g[1] = p1;
        if (q >= 3) g[2] = p2;
    }
};

// TEXT: targets may have extents (the separation criteria in floating point: box centres
// and the target's bounding box live in registers).  Without them the kernel needs a
// quarter fewer registers, which is a wave more per SIMD for a kernel bound by the latency
// of dependent loads.
// TWO (BT_WALK_TWO_PASS=1, an experiment: LAB_NOTES.md section 10): two passes over the colleagues.
// The first is the same for every lane of the wave: the colleague itself goes to List 1 if it is a
// source box, and the colleagues with source boxes below them are marked.  The second walks below
// the marked ones only, trip t being the lane's t-th such colleague: the lanes' walks start together
// instead of wherever each lane's row happens to hold a colleague with children, and the trips that
// find nothing to walk are gone.  Entries of List 3 and of the close list come from the walks alone
// and keep their order; List 1 is put into depth-first order afterwards whatever order it is
// written in.
template <class T, int D, bool ROWS, bool TEXT, bool TWO = false>
__global__ __launch_bounds__(256) void walk13_v2_kernel(TravArgs<T, D> a, FastTree ft, V2Walk w)
{
    const bool targets_have_extent = TEXT && a.targets_have_extent;
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
    constexpr int G = RowGroup<TEXT>::G;
    V2Emit<ROWS, G> e1, ec;
    if (ROWS) {
        e1 = V2Emit<ROWS, G>{w.row1 + tile * w.K1 + tl64 * G, w.K1, 0, 0, 0, 0};
        ec = V2Emit<ROWS, G>{w.rowc ? w.rowc + tile * w.Kc + tl64 * G : nullptr, w.rowc ? w.Kc : 0, 0, 0, 0, 0};
    } else {
        e1 = V2Emit<ROWS, G>{w.l1_lists + w.l1_cs[item], INT_MAX, 0, 0, 0, 0};
        ec = V2Emit<ROWS, G>{w.close_lists ? w.close_lists + w.close_cs[item] : nullptr, INT_MAX, 0, 0, 0, 0};
    }
    // List 1 is ordered by depth-first rank at the end.  The walk writes box numbers and the
    // ordering kernels look the ranks up (l1_finalize*_kernel: one lookup per entry, none to
    // map back); a lookup here would stall the walk once per entry, the store having to wait
    // for the load.
    int32_t sp1 = -1;                  // list 1 beyond its row: as for list 3 (store3, below)
    auto emit1 = [&](int32_t box) {
        if (ROWS && !TEXT && e1.n >= e1.cap && w.spill3 && sp1 != -2) {
            if (sp1 == -1) {
                const int shard = blockIdx.x & (SPILL_SHARDS - 1);
                const int32_t got = atomicAdd(w.spill_count + shard * 16, 1);
                sp1 = got < w.spill_per_shard ? shard * w.spill_per_shard + got : -2;
            }
            const int j = e1.n - e1.cap;
            if (sp1 >= 0 && j < SPILL_CHUNK) w.spill3[(int64_t) sp1 * SPILL_CHUNK + j] = box;
            else sp1 = -2;
        }
        e1(box);
    };
    int32_t *lvl = s_walk_lds + w.walk_cap * WALK_THREADS + threadIdx.x;
    int n3 = 0;
    int32_t *row3 = ROWS ? w.row3 + tile * w.K3 + tl64 * G : nullptr;
    int32_t q0 = 0, q1 = 0, q2 = 0;    // G = 4: list-3 entries of the group being filled
    if (ROWS) {
        for (int l = 0; l < w.nlevels; ++l) lvl[l * WALK_THREADS] = 0;
    } else {
        for (int l = 0; l < w.nlevels; ++l)
            lvl[l * WALK_THREADS] = item < w.lay.ecap[l] ? w.l3_cs[w.lay.base[l] + item] : 0;
    }
    int32_t sp = -1;                   // spill chunk of this item: -1 none yet, -2 not to be had
    auto store3 = [&](int lev, int32_t box) {          // ROWS: entry n3 of the item's list 3
        // (box numbers have V2_CODE_SHIFT bits on this path: the level rides above them,
        // one store per entry instead of two)
        const int32_t packed = box | (lev << V2_CODE_SHIFT);
        if (n3 < w.K3) {
            if (G == 1) {
                row3[(int64_t) n3 * 64] = packed;
            } else {
                const int q = n3 & 3;
                if (q == 0) q0 = packed; else if (q == 1) q1 = packed; else if (q == 2) q2 = packed;
                else *reinterpret_cast<PackedI4 *>(row3 + (int64_t) (n3 >> 2) * 256) = PackedI4{q0, q1, q2, packed};
            }
        } else if (!TEXT && w.spill3 && sp != -2) {     // (with extents: 2 registers over what six waves allow)
            if (sp == -1) {
                const int shard = blockIdx.x & (SPILL_SHARDS - 1);
                const int32_t got = atomicAdd(w.spill_count + shard * 16, 1);
                sp = got < w.spill_per_shard ? shard * w.spill_per_shard + got : -2;
            }
            const int j = n3 - w.K3;
            if (sp >= 0 && j < SPILL_CHUNK) w.spill3[(int64_t) sp * SPILL_CHUNK + j] = packed;
            else sp = -2;
        }
        ++n3;
    };
    auto emit3 = [&](int lev, int32_t box) {
        if (ROWS) {
            ++lvl[lev * WALK_THREADS];
            store3(lev, box);
        } else {
            w.l3_lists[lvl[lev * WALK_THREADS]++] = box;
        }
    };

    if (slot < 0) {
        if (w.flags[0] & BT_BOX_IS_SOURCE_BOX) emit1(0);       // traversal.py:489-495
        // b itself
        if (tl >= 1 && (bflags & BT_BOX_IS_SOURCE_BOX)) emit1(b);
        // coarser levels: the ancestors and their source-box colleagues.  A colleague
        // of the ancestor at offset o touches b iff b sits at the matching face of the
        // ancestor along every axis with o != 0.
        if (tl >= 2) {
            int32_t anc = a.parent[b];
            for (int k = tl - 1; k >= 1; --k, anc = a.parent[anc]) {
                if (w.flags[anc] & BT_BOX_IS_SOURCE_BOX) emit1(anc);
                const uint32_t mask = (1u << (tl - k)) - 1u;
                const int32_t *srow = w.srccoll_rows + (int64_t) anc * P;
                uint32_t smask = w.src_bit ? (uint32_t) w.srccoll_cnt[anc] : 0u;
                const int ns = w.src_bit ? __popc(smask) : w.srccoll_cnt[anc];
                for (int ii = 0; ii < ns; ++ii) {
                    const int i = w.src_bit ? __builtin_ctz(smask) : ii;
                    smask &= smask - 1u;
                    const uint32_t e = (uint32_t) srow[i];
                    bool adjacent = true;
#pragma unroll
                    for (int ax = 0; ax < D; ++ax) {
                        const int o = v2_off(e, ax);
                        const uint32_t r = cell.c[ax] & mask;
                        adjacent = adjacent && (o == 0 || (o < 0 ? r == 0u : r == mask));
                    }
                    if (adjacent) emit1(e & w.id_mask);
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
    // One centre and one radius per axis serve all criteria (registers decide how many
    // waves fit): the box centre and the stick-out radius (static criteria), or centre and
    // half widths of the targets' bounding box (precise criterion).
    T cen[D], rad[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { cen[i] = 0; rad[i] = 0; }
    if (targets_have_extent) {
        if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_STATIC_L2) {
            load_center(a, b, cen);
            const T stickout_rad = (1 + a.stick_out_factor) * level_to_rad(a.root_extent, tl);
#pragma unroll
            for (int i = 0; i < D; ++i) rad[i] = stickout_rad;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {          // load_true_box_extent, :177-198
                const T mn = a.tgt_bbox_min[i * a.aligned + b];
                const T mx = a.tgt_bbox_max[i * a.aligned + b];
                cen[i] = ((T) 0.5) * (mn + mx);
                rad[i] = ((T) 0.5) * (mx - mn);
            }
        }
    }

    // colleagues and everything below them
    int c0 = 0, c1 = ncoll;
    if (slot >= 0) { c0 = slot; c1 = (slot + 1 < ncoll) ? slot + 1 : ncoll; }
    else if (slot == SLOT_SELF) c1 = 0;
    const int32_t *crow = w.coll_rows + (int64_t) b * P;
    int32_t *stk = s_walk_lds + threadIdx.x;
    uint32_t below = 0;
    if constexpr (TWO) {
        for (int ci = c0; ci < c1; ++ci) {
            const uint32_t ce = (uint32_t) crow[ci];
            const int32_t nws = (int32_t) (ce & w.id_mask);
            const uint8_t cfl = w.flags[nws];
            if (cfl & BT_BOX_IS_SOURCE_BOX) emit1(nws);
            if (cfl & BT_BOX_HAS_SOURCE_CHILD_BOXES) below |= 1u << ci;
        }
    }
    for (int cnext = c0; TWO ? below != 0u : cnext < c1; ++cnext) {
        int ci = cnext;
        if constexpr (TWO) {
            ci = __builtin_ctz(below);
            below &= below - 1u;
        }
        const uint32_t ce = (uint32_t) crow[ci];
        const int32_t nws = (int32_t) (ce & w.id_mask);
        if constexpr (!TWO) {
            const uint8_t cfl = w.flags[nws];
            // a colleague is adjacent (well_sep_is_n_away == 1)
            if (cfl & BT_BOX_IS_SOURCE_BOX) emit1(nws);
            if (!(cfl & BT_BOX_HAS_SOURCE_CHILD_BOXES)) continue;
        }
        int prel[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) prel[ax] = v2_off(ce, ax);
        int size = 0, mnr = 0;
        int32_t parent = nws;
        bool go = true;
        // The children of the box being scanned sit in two registers (Kids: one 8-byte load
        // when the scan enters or returns to a box) and a step picks its child out of them:
        // a load per step made every step wait for a load of its own.
        Kids kd = v2_load_kids(w.child8, parent);
        if (!TEXT && !a.close_lists_exist) {
            // The bottom of the tree, where most of the walk happens: a colleague whose
            // children are all leaves.  Nothing is descended into, so its children are
            // handled in straight-line code -- the child words and the Morton bits are
            // compile-time selections, no stack, one update of the level counter -- instead
            // of one trip of the general loop each (a volume-filling cloud at 1.25*10^8
            // points: walk 3.0 -> see LAB_NOTES.md section 5).
            if (!kd.has_src_children()) {
                int n3_here = 0;
#pragma unroll
                for (int m = 0; m < C; ++m) {
                    if (!((kd.source() >> m) & 1u)) continue;
                    const int32_t wb = kd.id(m);
                    bool in_list_1 = true;
#pragma unroll
                    for (int ax = 0; ax < D; ++ax) {
                        const int r = 2 * prel[ax] + v2_mbit<D>(m, ax);
                        in_list_1 = in_list_1 && r >= -1 && r <= 2;
                    }
                    if (in_list_1) {
                        emit1(wb);
                    } else if (ROWS) {
                        store3(tl + 1, wb);
                        ++n3_here;
                    } else {
                        emit3(tl + 1, wb);
                    }
                }
                if (ROWS) lvl[(tl + 1) * WALK_THREADS] += n3_here;
                continue;
            }
        }
        // With target extents the separation criteria need the centre of every candidate.
        // On this path the tree is a lattice, i.e. every stored centre IS "parent centre
        // +/- level_to_rad(child level)" bit for bit (check_pack_kernel), so a child's
        // centre is computed from the centre of the box being scanned; that centre is
        // loaded when the scan enters a colleague or returns to a box -- once per box
        // instead of once per child (a random 24-byte load each: 10^8 of them at c4).
        // (kept in an LDS column: three more coordinates in registers cost a wave of occupancy)
        T *pcen = reinterpret_cast<T *>(s_walk_lds + (w.walk_cap + w.nlevels) * WALK_THREADS)
            + threadIdx.x;
        if (targets_have_extent) {
            T pc0[D];
            load_center(a, parent, pc0);
#pragma unroll
            for (int q = 0; q < D; ++q) pcen[q * WALK_THREADS] = pc0[q];
        }
        while (go) {
            const bool wb_src = (kd.source() >> mnr) & 1u, wb_hsc = (kd.has_src_children() >> mnr) & 1u;
            const int32_t wb = kd.id(mnr);              // (used only if the child is there)
            bool descend = false;
            int rel[D];
            if (wb_src || wb_hsc) {                     // (set for present children only)
                const int k = size + 1;                 // level of wb minus tl
                bool in_list_1 = true;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    rel[ax] = 2 * prel[ax] + v2_mbit<D>(mnr, ax);
                    in_list_1 = in_list_1 && rel[ax] >= -1 && rel[ax] <= (1 << k);
                }
                const int wl = tl + k;
                // centre of wb (target extents only): pcen +/- the radius of its level, per
                // axis where it is needed -- not kept across the emits
                const T child_rad = targets_have_extent ? level_to_rad(a.root_extent, wl) : (T) 0;
                auto wc = [&](int q) -> T {
                    const T pq = pcen[q * WALK_THREADS];
                    return v2_mbit<D>(mnr, q) ? pq + child_rad : pq - child_rad;
                };
                if (in_list_1) {
                    if (wb_src) emit1(wb);
                    descend = wb_hsc;
                } else {
                    bool meets = true;
                    if (targets_have_extent) {
                        const T source_rad = level_to_rad(a.root_extent, wl);
                        if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_PRECISE_LINF) {
                            T l_inf = 0;
#pragma unroll
                            for (int q = 0; q < D; ++q) {
                                T d = cen[q] - wc(q);
                                d = (d < 0) ? -d : d;
                                const T v = d - rad[q] - source_rad;
                                l_inf = (v > l_inf) ? v : l_inf;
                            }
                            meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                        } else {
                            T l2sq = 0;
#pragma unroll
                            for (int q = 0; q < D; ++q) {
                                const T d = cen[q] - wc(q);
                                l2sq = l2sq + d * d;
                            }
                            const T rhs = sqrt(l2sq) - sqrt((T) D) * rad[0] - source_rad;
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
                        if (wb_src) ec(wb);
                        descend = wb_hsc;
                    }
                }
            }
            if (descend) {
                stk[size * WALK_THREADS] = parent | (mnr << 28);
                ++size;
                if (targets_have_extent) {
                    const T child_rad = level_to_rad(a.root_extent, tl + size);
#pragma unroll
                    for (int q = 0; q < D; ++q)
                        pcen[q * WALK_THREADS] = v2_mbit<D>(mnr, q) ? pcen[q * WALK_THREADS] + child_rad
                                                                   : pcen[q * WALK_THREADS] - child_rad;
                }
                parent = wb; mnr = 0;
                kd = v2_load_kids(w.child8, parent);
#pragma unroll
                for (int ax = 0; ax < D; ++ax) prel[ax] = rel[ax];
                continue;
            }
            bool popped = false;
            while (true) {                              // walk_advance
                ++mnr;
                if (mnr < C) break;
                go = size > 0;
                if (!go) break;
                --size;
                const int32_t e = stk[size * WALK_THREADS];
                parent = e & 0x0fffffff;
                mnr = (int) ((uint32_t) e >> 28);
                popped = true;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) prel[ax] >>= 1;
            }
            if (popped && go) {
                kd = v2_load_kids(w.child8, parent);
                if (targets_have_extent) {
                    T pc0[D];
                    load_center(a, parent, pc0);
#pragma unroll
                    for (int q = 0; q < D; ++q) pcen[q * WALK_THREADS] = pc0[q];
                }
            }
        }
    }

    if (ROWS) {
        e1.flush();
        ec.flush();
        if (G == 4) {
            const int m = n3 < w.K3 ? n3 : w.K3, q = m & 3;
            int32_t *g = row3 + (int64_t) (m >> 2) * 256;
            if (q >= 1) g[0] = q0;
            if (q >= 2) g[1] = q1;
            if (q >= 3) g[2] = q2;
        }
        for (int l = 0; l < w.nlevels; ++l)
            if (item < w.lay.ecap[l]) w.l3_cs[w.lay.base[l] + item] = lvl[l * WALK_THREADS];
        w.l1_cs[item] = e1.n + blk_len;
        if (w.close_cs) w.close_cs[item] = ec.n;
        const bool ovf = (e1.n > w.K1 && sp1 < 0) || (n3 > w.K3 && sp < 0) || (w.close_cs && ec.n > w.Kc);
        w.overflow[item] = ovf ? 1 : 0;
        if (w.spill_idx) {
            w.spill_idx[item] = (!ovf && n3 > w.K3) ? sp : -1;
            w.spill_idx[w.items_cap + item] = (!ovf && e1.n > w.K1) ? sp1 : -1;
        }
        // one append per wave (the lanes are together again here): 10^5 appends to one
        // counter, one by one, cost as much as a tenth of the walk
        const uint64_t obal = __ballot(ovf);
        if (obal) {
            const int lane = threadIdx.x & 63;
            const int leader = __ffsll((long long) obal) - 1;
            int32_t base = 0;
            if (lane == leader) base = atomicAdd(w.ovf_count, (int32_t) __popcll(obal));
            base = __shfl(base, leader, 64);
            if (ovf) w.ovf_list[base + __popcll(obal & ((1ull << lane) - 1ull))] = item;
        }
        if (ovf && w.dbg_counts) {          // BT_TRAV_STATS: why items overflow
            if (e1.n > w.K1 && sp1 < 0) atomicAdd(w.dbg_counts + 0, 1);
            if (n3 > w.K3 && sp < 0) atomicAdd(w.dbg_counts + 1, 1);
            if (w.close_cs && ec.n > w.Kc) atomicAdd(w.dbg_counts + 2, 1);
            if (slot >= 0) atomicAdd(w.dbg_counts + 3, 1);
        }
    }
}

// ---- the first walk with 2^d lanes per work item -------------------------------------------------
//
// (BT_WALK_G8=1 | 2, an experiment awaiting its measurement on a GPU -- its lists equal the oracle's
// under the CPU emulation of tests/emu, LAB_NOTES.md sections 10-12.)  The kernel
// above gives a lane a work item and lets it test ONE child of the box it scans per trip; the 64
// walks of a wave diverge, and what the walk of the 10^8 + 10^7 extent tree is bound by is the
// issue of vector instructions at 16 of 64 lanes active.  Here a GROUP of C = 2^d lanes owns an
// item: lane m of the group tests child slot m of the box being scanned, so a scanned box costs the
// group one trip -- its Kids word and its centre are loaded once for the group -- and a wave makes
// 64 / C walks whose control flow is uniform within each group.
//
// What has to come out the same (the rows feed rows_to_csr_v2_kernel / l3_scatter_v2_kernel and the
// second walk of the overflowed items unchanged; an item whose lists outgrow its rows is walked
// again -- no spill chunks here):
//  * List 3 is kept per source level, and within a level in the order the entries are emitted.  The
//    one-lane walk emits the entries of level l + 1 when it tests the children of a box of level l,
//    boxes being scanned in depth-first preorder and children in slot order.  Testing a box's
//    children all at once keeps exactly that: boxes are still scanned in preorder, a box's children
//    are appended in slot order (a ballot prefix over the group), and what the subtree of a child
//    emits lies at deeper levels.
//  * The close list is ONE list per item, in emission order: child m's entry, then everything the
//    subtree below m emits, then child m + 1's.  So the children that go to the close list or are
//    descended into are handled one after the other, in slot order, after the tests: the masks of
//    both kinds wait on the group's stack.
//  * List 1 is put into depth-first order afterwards whatever order it is written in.
// Per-item state (counts, stack, level counters) is uniform over the group: every lane of the
// group holds the same values, only the child tests and the row stores differ by lane.
template <class T, int D, bool TEXT, bool IM /* item-major rows: a group's entries are neighbours */>
__global__ __launch_bounds__(256) void walk13_g8_kernel(TravArgs<T, D> a, FastTree ft, V2Walk w)
{
    constexpr int C = 1 << D;
    constexpr int G = RowGroup<TEXT>::G;               // entries per group of the rows (V2Emit)
    constexpr int P = V2Dims<D>::P;
    constexpr int IPB = WALK_THREADS / C;              // items per workgroup
    const int sub = threadIdx.x % C;                   // the child slot this lane tests
    const int git = threadIdx.x / C;                   // the group's number in the workgroup
    const int lane = threadIdx.x & 63;
    const int gshift = lane & ~(C - 1);                // the group's first lane in its wave
    const uint32_t gmask = (C == 64) ? ~0u : ((1u << C) - 1u);
    const int32_t item = blockIdx.x * IPB + git;
    const int32_t nitems = *w.d_nitems;
    if (item >= nitems) {
        if (item < w.items_cap) {
            // the count arrays are scanned over all their columns
            for (int l = sub; l < w.nlevels; l += C)
                if (item < w.lay.ecap[l]) w.l3_cs[w.lay.base[l] + item] = 0;
            if (sub == 0) {
                w.l1_cs[item] = 0;
                if (w.close_cs) w.close_cs[item] = 0;
                w.overflow[item] = 0;
                if (w.spill_idx) { w.spill_idx[item] = -1; w.spill_idx[w.items_cap + item] = -1; }
            }
        }
        return;                                        // whole groups drop out together
    }
    const int32_t tbn = w.item_tbn[item];
    const int slot = w.item_slot[item];
    const int32_t b = a.target_boxes[tbn];
    const ICell cell = w.cells[b];
    const int tl = (int) (cell.lf & 0xffu);
    const uint8_t bflags = (uint8_t) (cell.lf >> 8);

    // rows: tiles of 64 items, entry j of item t of a tile at (j / G) * 64 * G + t * G + j % G (V2Emit)
    const int64_t tile = (int64_t) (item >> 6) * 64;
    const int tl64 = item & 63;
    int32_t *row1 = IM ? w.row1 + (int64_t) item * w.K1 : w.row1 + tile * w.K1 + tl64 * G;
    int32_t *row3 = IM ? w.row3 + (int64_t) item * w.K3 : w.row3 + tile * w.K3 + tl64 * G;
    int32_t *rowc = !w.rowc ? nullptr : IM ? w.rowc + (int64_t) item * w.Kc : w.rowc + tile * w.Kc + tl64 * G;
    auto slot_of = [](int j) -> int64_t { return IM ? (int64_t) j : (int64_t) (j / G) * (64 * G) + (j % G); };
    int n1 = 0, n3 = 0, nc = 0;                        // group-uniform
    // LDS: per item a stack of {box | masks} pairs and the per-level List-3 counters
    int32_t *stk = s_walk_lds + git;                   // entry i: stk[(2 i) * IPB], stk[(2 i + 1) * IPB]
    int32_t *lvl = s_walk_lds + 2 * w.walk_cap * IPB + git;
    for (int l = sub; l < w.nlevels; l += C) lvl[l * IPB] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // appends of the group: the lanes with `take` store their value at the list's end, in lane order
    auto group_bits = [&](bool take) -> uint32_t {
        return (uint32_t) (__ballot(take) >> gshift) & gmask;
    };
    auto append = [&](int32_t *row, int cap, int &n, bool take, int32_t v) {
        const uint32_t m = group_bits(take);
        if (take) {
            const int j = n + __popc(m & ((1u << sub) - 1u));
            if (j < cap) row[slot_of(j)] = v;
        }
        n += __popc(m);
    };
    auto append_one = [&](int32_t *row, int cap, int &n, int32_t v) {       // group-uniform value
        if (sub == 0 && n < cap) row[slot_of(n)] = v;
        ++n;
    };

    if (slot < 0) {
        if (w.flags[0] & BT_BOX_IS_SOURCE_BOX) append_one(row1, w.K1, n1, 0);      // traversal.py:489-495
        if (tl >= 1 && (bflags & BT_BOX_IS_SOURCE_BOX)) append_one(row1, w.K1, n1, b);
        // coarser levels: the ancestors and their source-box colleagues (see walk13_v2_kernel); the
        // group takes C row entries per trip
        if (tl >= 2) {
            int32_t anc = a.parent[b];
            for (int k = tl - 1; k >= 1; --k, anc = a.parent[anc]) {
                if (w.flags[anc] & BT_BOX_IS_SOURCE_BOX) append_one(row1, w.K1, n1, anc);
                const uint32_t mask = (1u << (tl - k)) - 1u;
                const int32_t *srow = w.srccoll_rows + (int64_t) anc * P;
                const uint32_t smask0 = w.src_bit ? (uint32_t) w.srccoll_cnt[anc] : 0u;
                const int ns = w.src_bit ? __popc(smask0) : w.srccoll_cnt[anc];
                for (int i0 = 0; i0 < ns; i0 += C) {
                    const int ii = i0 + sub;
                    bool adjacent = ii < ns;
                    uint32_t e = 0;
                    if (adjacent) {
                        int i = ii;
                        if (w.src_bit) {               // the ii-th set bit of the mask
                            uint32_t sm = smask0;
                            for (int q = 0; q < ii; ++q) sm &= sm - 1u;
                            i = __builtin_ctz(sm);
                        }
                        e = (uint32_t) srow[i];
#pragma unroll
                        for (int ax = 0; ax < D; ++ax) {
                            const int o = v2_off(e, ax);
                            const uint32_t r = cell.c[ax] & mask;
                            adjacent = adjacent && (o == 0 || (o < 0 ? r == 0u : r == mask));
                        }
                    }
                    append(row1, w.K1, n1, adjacent, (int32_t) (e & w.id_mask));
                }
            }
        }
    }

    // the box's LAST item reserves the space of the own-subtree block at the end of the box's
    // list-1 segment
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
    const bool targets_have_extent = TEXT && a.targets_have_extent;
    T cen[D], rad[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { cen[i] = 0; rad[i] = 0; }
    if (targets_have_extent) {
        if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_STATIC_L2) {
            load_center(a, b, cen);
            const T stickout_rad = (1 + a.stick_out_factor) * level_to_rad(a.root_extent, tl);
#pragma unroll
            for (int i = 0; i < D; ++i) rad[i] = stickout_rad;
        } else {
#pragma unroll
            for (int i = 0; i < D; ++i) {          // load_true_box_extent, :177-198
                const T mn = a.tgt_bbox_min[i * a.aligned + b];
                const T mx = a.tgt_bbox_max[i * a.aligned + b];
                cen[i] = ((T) 0.5) * (mn + mx);
                rad[i] = ((T) 0.5) * (mx - mn);
            }
        }
    }

    // colleagues and everything below them
    int c0 = 0, c1 = ncoll;
    if (slot >= 0) { c0 = slot; c1 = (slot + 1 < ncoll) ? slot + 1 : ncoll; }
    else if (slot == SLOT_SELF) c1 = 0;
    const int32_t *crow = w.coll_rows + (int64_t) b * P;
    for (int ci = c0; ci < c1; ++ci) {
        const uint32_t ce = (uint32_t) crow[ci];
        const int32_t nws = (int32_t) (ce & w.id_mask);
        const uint8_t cfl = w.flags[nws];
        // a colleague is adjacent (well_sep_is_n_away == 1)
        if (cfl & BT_BOX_IS_SOURCE_BOX) append_one(row1, w.K1, n1, nws);
        if (!(cfl & BT_BOX_HAS_SOURCE_CHILD_BOXES)) continue;
        int prel[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) prel[ax] = v2_off(ce, ax);
        int size = 0;                              // stack entries below the box being scanned
        int32_t parent = nws;
        bool fresh = true;                         // `parent` has not been scanned yet
        uint32_t closem = 0, descm = 0;            // children of `parent` still to take, slot order
        Kids kd{0u, 0u};
        while (true) {
            if (fresh) {
                // ---- scan `parent`: lane `sub` tests child slot `sub` --------------------------------
                kd = v2_load_kids(w.child8, parent);
                // (the centre of the box being scanned is needed here only: the tree is a lattice on
                // this path, a child's centre is the parent's +/- the radius of its level, bit for bit)
                T pcen[D];
#pragma unroll
                for (int q = 0; q < D; ++q) pcen[q] = 0;
                if (targets_have_extent) load_center(a, parent, pcen);
                const bool wb_src = (kd.source() >> sub) & 1u, wb_hsc = (kd.has_src_children() >> sub) & 1u;
                const int32_t wb = kd.id(sub);          // (used only if the child is there)
                const int k = size + 1;                 // level of the children minus tl
                const int wl = tl + k;
                bool to1 = false, to3 = false, toc = false, desc = false;
                if (wb_src || wb_hsc) {                 // (set for present children only)
                    bool in_list_1 = true;
#pragma unroll
                    for (int ax = 0; ax < D; ++ax) {
                        const int r = 2 * prel[ax] + v2_mbit<D>(sub, ax);
                        in_list_1 = in_list_1 && r >= -1 && r <= (1 << k);
                    }
                    if (in_list_1) {
                        to1 = wb_src;
                        desc = wb_hsc;
                    } else {
                        bool meets = true;
                        if (targets_have_extent) {
                            const T source_rad = level_to_rad(a.root_extent, wl);
                            const T child_rad = source_rad;
                            if (a.crit == BT_CRIT_STATIC_LINF || a.crit == BT_CRIT_PRECISE_LINF) {
                                T l_inf = 0;
#pragma unroll
                                for (int q = 0; q < D; ++q) {
                                    const T wcq = v2_mbit<D>(sub, q) ? pcen[q] + child_rad : pcen[q] - child_rad;
                                    T d = cen[q] - wcq;
                                    d = (d < 0) ? -d : d;
                                    const T v = d - rad[q] - source_rad;
                                    l_inf = (v > l_inf) ? v : l_inf;
                                }
                                meets = l_inf >= (2 - 8 * Eps<T>::v) * source_rad;
                            } else {
                                T l2sq = 0;
#pragma unroll
                                for (int q = 0; q < D; ++q) {
                                    const T wcq = v2_mbit<D>(sub, q) ? pcen[q] + child_rad : pcen[q] - child_rad;
                                    const T d = cen[q] - wcq;
                                    l2sq = l2sq + d * d;
                                }
                                const T rhs = sqrt(l2sq) - sqrt((T) D) * rad[0] - source_rad;
                                meets = ((2 - 8 * Eps<T>::v) * source_rad <= rhs);
                            }
                        }
                        const bool force_close = a.close_lists_exist && a.min_nsources_cumul > 0
                            && (a.src_counts_cumul[wb] < a.min_nsources_cumul);
                        if (meets && !force_close) {
                            to3 = true;
                        } else if (a.close_lists_exist) {
                            toc = wb_src;
                            desc = wb_hsc;
                        }
                    }
                }
                append(row1, w.K1, n1, to1, wb);
                {
                    const uint32_t m3 = group_bits(to3);
                    if (to3) {
                        const int j = n3 + __popc(m3 & ((1u << sub) - 1u));
                        if (j < w.K3) row3[slot_of(j)] = wb | (wl << V2_CODE_SHIFT);
                    }
                    const int c3 = __popc(m3);
                    n3 += c3;
                    if (sub == 0 && c3) lvl[wl * IPB] += c3;
                }
                closem = group_bits(toc);
                descm = group_bits(desc);
                fresh = false;
            }
            // ---- the children of `parent` that go to the close list or are walked into, in slot order
            const uint32_t pend = closem | descm;
            if (pend) {
                const int m = __builtin_ctz(pend);
                const int32_t wb = kd.id(m);
                if ((closem >> m) & 1u) append_one(rowc, w.Kc, nc, wb);
                closem &= ~(1u << m);
                if ((descm >> m) & 1u) {
                    descm &= ~(1u << m);
                    if (sub == 0) {
                        stk[(2 * size) * IPB] = parent;
                        stk[(2 * size + 1) * IPB] = (int32_t) (closem | (descm << 8));
                    }
                    ++size;
#pragma unroll
                    for (int ax = 0; ax < D; ++ax) prel[ax] = 2 * prel[ax] + v2_mbit<D>(m, ax);
                    parent = wb;
                    fresh = true;
                }
                continue;
            }
            // ---- `parent` is finished: back to the box it was entered from ------------------------
            if (size == 0) break;
            --size;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            parent = stk[(2 * size) * IPB];
            const uint32_t ms = (uint32_t) stk[(2 * size + 1) * IPB];
            closem = ms & 0xffu; descm = (ms >> 8) & 0xffu;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) prel[ax] >>= 1;
            if (closem | descm) kd = v2_load_kids(w.child8, parent);
        }
    }

    // ---- the item's counts ---------------------------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int l = sub; l < w.nlevels; l += C)
        if (item < w.lay.ecap[l]) w.l3_cs[w.lay.base[l] + item] = lvl[l * IPB];
    const bool ovf = n1 > w.K1 || n3 > w.K3 || (w.close_cs && nc > w.Kc);
    if (sub == 0) {
        w.l1_cs[item] = n1 + blk_len;
        if (w.close_cs) w.close_cs[item] = nc;
        w.overflow[item] = ovf ? 1 : 0;
        if (w.spill_idx) { w.spill_idx[item] = -1; w.spill_idx[w.items_cap + item] = -1; }
    }
    // one append per wave (the lanes are together again here)
    const uint64_t obal = __ballot(ovf && sub == 0);
    if (obal) {
        const int leader = __ffsll((long long) obal) - 1;
        int32_t base = 0;
        if (lane == leader) base = atomicAdd(w.ovf_count, (int32_t) __popcll(obal));
        base = __shfl(base, leader, 64);
        if (ovf && sub == 0) w.ovf_list[base + __popcll(obal & ((1ull << lane) - 1ull))] = item;
    }
}

// rows -> final places, one wave per tile of 64 items: a lane reads entry j of its own
// item (the wave reads 256 contiguous bytes) and writes it to the item's CSR segment
// IM: the rows are item-major (entry j of item t at t * K + j: what walk13_g8_kernel<..., true> writes)
template <int G /* entries per group of the rows, see V2Emit */, bool IM = false>
__global__ __launch_bounds__(256) void rows_to_csr_v2_kernel(const int32_t *d_nitems,
        const uint8_t *overflow, const int32_t *rows, int K, const int32_t *starts,
        const int32_t *translate /* entries are indices into this table, or null */,
        int32_t ntranslate, int32_t *lists,
        const int32_t *spill_idx = nullptr /* [items]: chunk with the entries beyond K, or -1 */,
        const int32_t *spill = nullptr)
{
    const int32_t item = blockIdx.x * 256 + threadIdx.x;
    const int32_t nitems = *d_nitems;
    const bool active = item < nitems && !overflow[item];
    int32_t s = 0, n = 0, n_all = 0;
    if (active) { s = starts[item]; n = n_all = starts[item + 1] - s; if (n > K) n = K; }
    int nmax = n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(nmax, off, 64);
        nmax = o > nmax ? o : nmax;
    }
    const int32_t *row = IM ? rows + (int64_t) item * K : rows + (int64_t) (item >> 6) * 64 * K + (item & 63) * G;
    // eight loads in flight per lane: one load and one store per trip made every trip wait
    // for its load (the stores may alias the rows as far as the compiler knows)
    constexpr int UNR = 8;
    for (int j0 = 0; j0 < nmax; j0 += UNR) {
        int32_t v[UNR];
        if (IM) {
#pragma unroll
            for (int u = 0; u < UNR; u += 4) {
                PackedI4 g{0, 0, 0, 0};
                if (j0 + u + 4 <= K) { if (j0 + u < n) g = *reinterpret_cast<const PackedI4 *>(row + j0 + u); }
                else {                                  // (a row that is not a whole number of groups)
                    if (j0 + u < n) g.x = row[j0 + u];
                    if (j0 + u + 1 < n) g.y = row[j0 + u + 1];
                    if (j0 + u + 2 < n) g.z = row[j0 + u + 2];
                }
                v[u] = g.x; v[u + 1] = g.y; v[u + 2] = g.z; v[u + 3] = g.w;
            }
        } else if (G == 1) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = (j0 + u < n) ? row[(int64_t) (j0 + u) * 64] : 0;
        } else {
#pragma unroll
            for (int u = 0; u < UNR; u += 4) {
                PackedI4 g{0, 0, 0, 0};
                if (j0 + u < n) g = *reinterpret_cast<const PackedI4 *>(row + (int64_t) ((j0 + u) >> 2) * 256);
                v[u] = g.x; v[u + 1] = g.y; v[u + 2] = g.z; v[u + 3] = g.w;
            }
        }
        if (translate) {
#pragma unroll
            // (a list's segment may end with space reserved for a block that is copied in
            // later: the row slots behind the item's entries hold no box numbers)
            for (int u = 0; u < UNR; ++u)
                v[u] = (j0 + u < n && (uint32_t) v[u] < (uint32_t) ntranslate) ? translate[v[u]] : 0;
        }
        // a lane's entries are contiguous in ITS segment: full groups of four leave as one
        // 16-byte store (4-byte aligned -- the hardware takes that), a quarter of the write
        // requests of the lane-strided single stores this kernel is bound by
        int32_t *dst = lists + (int64_t) s + j0;
#pragma unroll
        for (int u = 0; u < UNR; u += 4) {
            if (j0 + u + 4 <= n) {
                *reinterpret_cast<PackedI4 *>(dst + u) = PackedI4{v[u], v[u + 1], v[u + 2], v[u + 3]};
            } else {
#pragma unroll
                for (int q = u; q < u + 4; ++q)
                    if (j0 + q < n) dst[q] = v[q];
            }
        }
    }
    if (spill_idx && n_all > K) {
        // (few items, long lists: a lane copies its own chunk front to back)
        const int32_t sp = spill_idx[item];
        if (sp >= 0)
            for (int j = 0; j < n_all - K; ++j)
                lists[(int64_t) s + K + j] = spill[(int64_t) sp * SPILL_CHUNK + j];
    }
}

// list 3: rows -> per-level lists (cursors start at the item's per-level starts)
template <int G /* entries per group of the rows, see V2Emit */, bool IM = false /* item-major rows */>
__global__ __launch_bounds__(256) void l3_scatter_v2_kernel(const int32_t *d_nitems, L3Layout lay,
        int nlevels, const uint8_t *overflow, const int32_t *row3, int K3,
        const int32_t *l3_item_starts, int32_t *l3_lists, const int32_t *spill_idx,
        const int32_t *spill3)
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
    const int32_t *row = IM ? row3 + (int64_t) item * K3 : row3 + (int64_t) (item >> 6) * 64 * K3 + (item & 63) * G;
    constexpr int UNR = 8;                 // loads in flight per lane (see rows_to_csr_v2_kernel)
    const int n_all = n;
    n = n < K3 ? n : K3;                   // the rest is in the item's spill chunk
    for (int j0 = 0; j0 < n; j0 += UNR) {
        int32_t v[UNR];
        int lev[UNR];
        int32_t raw[UNR];
        if (IM) {
#pragma unroll
            for (int u = 0; u < UNR; u += 4) {
                PackedI4 g{0, 0, 0, 0};
                if (j0 + u + 4 <= K3) { if (j0 + u < n) g = *reinterpret_cast<const PackedI4 *>(row + j0 + u); }
                else {
                    if (j0 + u < n) g.x = row[j0 + u];
                    if (j0 + u + 1 < n) g.y = row[j0 + u + 1];
                    if (j0 + u + 2 < n) g.z = row[j0 + u + 2];
                }
                raw[u] = g.x; raw[u + 1] = g.y; raw[u + 2] = g.z; raw[u + 3] = g.w;
            }
        } else if (G == 1) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) raw[u] = (j0 + u < n) ? row[(int64_t) (j0 + u) * 64] : 0;
        } else {
#pragma unroll
            for (int u = 0; u < UNR; u += 4) {
                PackedI4 g{0, 0, 0, 0};
                if (j0 + u < n) g = *reinterpret_cast<const PackedI4 *>(row + (int64_t) ((j0 + u) >> 2) * 256);
                raw[u] = g.x; raw[u + 1] = g.y; raw[u + 2] = g.z; raw[u + 3] = g.w;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            v[u] = raw[u] & (int32_t) V2_ID_MASK;
            lev[u] = raw[u] >> V2_CODE_SHIFT;
        }
        // the walk emits the children of a box one after the other: four entries of one level
        // leave as one 16-byte store
#pragma unroll
        for (int u = 0; u < UNR; u += 4) {
            if (j0 + u + 4 <= n && lev[u] == lev[u + 1] && lev[u] == lev[u + 2] && lev[u] == lev[u + 3]) {
                int32_t *cp = cur + lev[u] * WALK_THREADS;
                const int32_t c = *cp;
                *reinterpret_cast<PackedI4 *>(l3_lists + c) = PackedI4{v[u], v[u + 1], v[u + 2], v[u + 3]};
                *cp = c + 4;
            } else {
#pragma unroll
                for (int q = u; q < u + 4; ++q)
                    if (j0 + q < n) l3_lists[cur[lev[q] * WALK_THREADS]++] = v[q];
            }
        }
    }
    if (n_all > K3 && spill_idx) {
        // (few items, long lists: a lane reads its own chunk front to back)
        const int64_t base = (int64_t) spill_idx[item] * SPILL_CHUNK;
        for (int j = 0; j < n_all - K3; ++j) {
            const int32_t e = spill3[base + j];
            l3_lists[cur[(e >> V2_CODE_SHIFT) * WALK_THREADS]++] = e & (int32_t) V2_ID_MASK;
        }
    }
}

// list-3 starts per (level, target box) from the per-(level, item) starts: a box whose
// first item lies beyond the level's columns has no entries there and starts where
// the level ends
__global__ __launch_bounds__(256) void l3_box_starts_v2_kernel(int64_t nflat_box, int32_t ntb,
        L3Layout lay, int nlevels, const int32_t *first_item, const int32_t *item_starts,
        int32_t *box_starts, int lev0 /* the first level with a row */)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i > nflat_box) return;
    if (i == nflat_box) { box_starts[i] = item_starts[lay.base[nlevels]]; return; }
    const int lev = lev0 + (int) (i / ntb);
    const int32_t tbn = (int32_t) (i % ntb);
    const int32_t item = first_item[tbn];
    const int32_t col = item < lay.ecap[lev] ? item : lay.ecap[lev];
    box_starts[i] = item_starts[lay.base[lev] + col];
}
