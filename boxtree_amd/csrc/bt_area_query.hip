// Peer lists, area queries, leaves-to-balls lookup and space invader queries
// (boxtree/area_query.py) for gfx950.
//
//   peer lists      top-down, one launch per level: the peers of a box are the
//                   peers of its parent, each either kept, replaced by its
//                   adjacent children, or (not touching the box) replaced by the
//                   first ancestor that does when no child of that ancestor does.  Same lists, same order as the reference's
//                   walk from the root (area_query.py:393-475), which re-descends
//                   from box 0 for every box.
//   area query      one thread per ball: guiding box (:172-292), then a pruned
//                   walk below every peer of the guiding box (:295-366); count
//                   pass, scan, fill pass.  Walk stacks live in LDS.
//   leaves-to-balls stable 32-bit radix sort of (leaf, ball) pairs (:847-924)
//   space invader   same walk, float32 atomic max per leaf (:613-651)
#include "bt_common.hpp"
#include "bt_prims.hpp"
#include "bt_geom.hpp"
#include "bt_sort.hpp"

#include <algorithm>
#include <vector>

using namespace bt;

struct AqState {
    Buf<int32_t> starts, lists;
    int64_t nrows = 0, nentries = 0;
};

void bt_free_aq_state(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    delete ctx->aq;
    ctx->aq = nullptr;
}

namespace {

constexpr uint32_t HAS_CHILDREN = 12u << 8;     // HAS_SOURCE_OR_TARGET_CHILD_BOXES in Node::lf

template <class T, int D>
struct AqTree {
    const Node<T, D> *nodes;
    const int32_t *child_t;     // [nboxes][C]
    const int32_t *parent;
    int32_t nboxes;
    T root_extent;
    T bbox_min[D];
};

constexpr int pow3(int d) { return d == 1 ? 3 : d == 2 ? 9 : 27; }

// ---- peer lists ------------------------------------------------------------

// One thread per box of one level.  rows[box][P] holds the peers of the levels
// above (written by earlier launches), counts[box] their number.
template <class T, int D>
__global__ __launch_bounds__(256) void peer_level_kernel(AqTree<T, D> t, int32_t box_begin,
        int32_t box_end, int level, int32_t *rows, int32_t *counts, DeviceStatus *status)
{
    constexpr int C = 1 << D;
    constexpr int P = pow3(D);
    const int32_t b = box_begin + blockIdx.x * 256 + threadIdx.x;
    if (b >= box_end) return;
    int32_t *out = rows + (int64_t) b * P;
    if (b == 0) {                       // area_query.py:413-417
        out[0] = 0; counts[0] = 1;
        return;
    }
    const Node<T, D> me = t.nodes[b];
    const int32_t par = t.parent[b];
    const int32_t *prow = rows + (int64_t) par * P;
    const int np = counts[par];
    int n = 0;
    bool overflow = false;
    int32_t last_anc = -1;
    auto emit = [&](int32_t x) { if (n < P) out[n++] = x; else overflow = true; };
    for (int i = 0; i < np; ++i) {
        const int32_t p = prow[i];
        const Node<T, D> pn = t.nodes[p];
        const int pl = (int) (pn.lf & 0xffu);
        if (!adj<T, D>(t.root_extent, me.c, level, pn.c, pl)) {
            // The walk for b stops above p, at the first ancestor a of p that
            // touches b: a is a peer of b when none of its children touches b
            // (children of a may have been pruned away next to b).  All peers of
            // the parent below a are consecutive, so a is looked at once.
            int32_t a = t.parent[p];
            Node<T, D> an = t.nodes[a];
            while (a != 0 && !adj<T, D>(t.root_extent, me.c, level, an.c, (int) (an.lf & 0xffu))) {
                a = t.parent[a];
                an = t.nodes[a];
            }
            if (a == last_anc) continue;
            last_anc = a;
            bool any = false;
#pragma unroll
            for (int m = 0; m < C; ++m) {
                const int32_t c = t.child_t[(int64_t) a * C + m];
                if (!c) continue;
                const Node<T, D> cn = t.nodes[c];
                any |= adj<T, D>(t.root_extent, me.c, level, cn.c, (int) (an.lf & 0xffu) + 1);
            }
            if (!any) emit(a);
            continue;
        }
        if (!(pn.lf & HAS_CHILDREN) || pl + 1 < level) {
            // a leaf, or a bigger box none of whose children touches the parent
            emit(p);
            continue;
        }
        int nadj = 0;
#pragma unroll
        for (int m = 0; m < C; ++m) {
            const int32_t c = t.child_t[(int64_t) p * C + m];
            if (!c) continue;
            const Node<T, D> cn = t.nodes[c];
            if (adj<T, D>(t.root_extent, me.c, level, cn.c, pl + 1)) { emit(c); ++nadj; }
        }
        if (!nadj) emit(p);             // must_be_peer, :443-460
    }
    counts[b] = n;
    if (overflow) atomicExch(&status->internal, 1);
}

struct RowCount {
    const int32_t *counts;
    __device__ __forceinline__ int32_t operator()(int64_t i) const { return counts[i]; }
};

template <int P>
__global__ __launch_bounds__(256) void compact_rows_kernel(int32_t nrows, const int32_t *rows,
        const int32_t *starts, int32_t *lists)
{
    // P lanes per row
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / 32;
    const int lane = (int) (gid % 32);
    if (r >= nrows) return;
    const int32_t s = starts[r], e = starts[r + 1];
    if (lane < e - s) lists[s + lane] = rows[r * P + lane];
}

// ---- ball walks ------------------------------------------------------------------

// traversal.py:200-214
template <class T, int D>
__device__ __forceinline__ T linf_dist(const T *bc, const Node<T, D> &n)
{
    T md = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T d = bc[i] - n.c[i];
        d = (d < 0) ? -d : d;
        md = (d > md) ? d : md;
    }
    return md;
}

template <class T, int D>
__device__ __forceinline__ bool ball_overlap(T root_extent, const Node<T, D> &n, T r, const T *bc)
{
    const T size_sum = level_to_rad(root_extent, (int) (n.lf & 0xffu)) + r;
    return linf_dist<T, D>(bc, n) <= size_sum;
}

// Conservative test for inner boxes: a leaf below `n` that meets the ball lies
// within rad(n) of n's centre, up to rounding of the centres far below rad(n); a
// whole extra rad(n) of slack keeps every such leaf.  The reference descends
// unconditionally -- skipping subtrees that cannot contain a hit changes nothing.
template <class T, int D>
__device__ __forceinline__ bool ball_may_reach(T root_extent, const Node<T, D> &n, T r, const T *bc)
{
    const T size_sum = 2 * level_to_rad(root_extent, (int) (n.lf & 0xffu)) + r;
    return linf_dist<T, D>(bc, n) <= size_sum;
}

// area_query.py:179-291
template <class T, int D>
__device__ __forceinline__ int32_t guiding_box(const AqTree<T, D> &t, const T *bc, T r)
{
    constexpr int C = 1 << D;
    T qc[D], bmax[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        bmax[d] = t.bbox_min[d] + (T) (t.root_extent / (1 + 1e-4));
        T c = bc[d];
        c = (c > t.bbox_min[d]) ? c : t.bbox_min[d];
        qc[d] = (bmax[d] < c) ? bmax[d] : c;
    }
    T qr = 0;
#pragma unroll
    for (int m = 0; m < C; ++m) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T off = ((1 << (D - 1 - d)) & m) ? +r : -r;
            T corner = bc[d] + off;
            corner = (corner > t.bbox_min[d]) ? corner : t.bbox_min[d];
            corner = (bmax[d] < corner) ? bmax[d] : corner;
            T dist = corner - qc[d];
            dist = (dist < 0) ? -dist : dist;
            qr = (dist > qr) ? dist : qr;
        }
    }
    int32_t box = 0;
    if (level_to_rad(t.root_extent, 0) / 2 >= qr) {
        for (unsigned lev = 0;; ++lev) {
            const uint32_t lf = t.nodes[box].lf;
            if (!(lf & HAS_CHILDREN)
                    || (level_to_rad(t.root_extent, (int) lev) / 2 < qr
                        && qr <= level_to_rad(t.root_extent, (int) lev)))
                break;
            int morton = 0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T off_scaled = (qc[d] - t.bbox_min[d]) / t.root_extent;
                const unsigned bits = (unsigned) (off_scaled * (T) (1U << (1 + lev)));
                morton |= (int) (bits & 1U) << (D - 1 - d);
            }
            const int32_t next = t.child_t[(int64_t) box * C + morton];
            if (next) box = next;
            else break;
        }
    }
    return box;
}

// area_query.py:295-366; found(leaf, node) per overlapping leaf, in walk order
template <class T, int D, class F>
__device__ __forceinline__ void ball_walk(const AqTree<T, D> &t, const int32_t *pl_starts,
        const int32_t *pl_lists, const T *bc, T r, int32_t *stack_column, F &found)
{
    constexpr int C = 1 << D;
    const int32_t g = guiding_box<T, D>(t, bc, r);
    Walk w(stack_column);
    for (int32_t i = pl_starts[g], e = pl_starts[g + 1]; i < e; ++i) {
        const int32_t peer = pl_lists[i];
        const Node<T, D> pn = t.nodes[peer];
        if (!(pn.lf & HAS_CHILDREN)) {
            if (ball_overlap<T, D>(t.root_extent, pn, r, bc)) found(peer, pn);
            continue;
        }
        if (!ball_may_reach<T, D>(t.root_extent, pn, r, bc)) continue;
        w.init(peer);
        while (w.go) {
            const int32_t wb = t.child_t[(int64_t) w.parent * C + w.mnr];
            if (wb) {
                const Node<T, D> wn = t.nodes[wb];
                if (!(wn.lf & HAS_CHILDREN)) {
                    if (ball_overlap<T, D>(t.root_extent, wn, r, bc)) found(wb, wn);
                } else if (ball_may_reach<T, D>(t.root_extent, wn, r, bc)) {
                    w.push(wb);
                    continue;
                }
            }
            w.template advance<C>();
        }
    }
}

template <class T, int D>
struct Balls {
    const T *c[D];
    const T *radii;
    int64_t n;
};

struct CountFound {
    int32_t n = 0;
    template <class N> __device__ __forceinline__ void operator()(int32_t, const N &) { ++n; }
};
struct WriteFound {
    int32_t *p;
    template <class N> __device__ __forceinline__ void operator()(int32_t b, const N &) { *p++ = b; }
};
template <class T, int D>
struct InvaderFound {
    int32_t *out;           // float32 bit patterns (non-negative: integer order == float order)
    const T *bc;
    __device__ __forceinline__ void operator()(int32_t leaf, const Node<T, D> &n)
    {
        const float f = (float) linf_dist<T, D>(bc, n);       // area_query.py:629-650
        const int32_t bits = __float_as_int(f);
        if (__hip_atomic_load(out + leaf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bits)
            atomicMax(out + leaf, bits);
    }
};

// MODE 0: count, 1: fill, 2: space invader
template <class T, int D, int MODE>
__global__ __launch_bounds__(WALK_THREADS) void ball_kernel(AqTree<T, D> t, Balls<T, D> balls,
        const int32_t *pl_starts, const int32_t *pl_lists, int32_t *counts_or_starts,
        int32_t *lists)
{
    const int64_t i = (int64_t) blockIdx.x * WALK_THREADS + threadIdx.x;
    if (i >= balls.n) return;
    T bc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) bc[d] = balls.c[d][i];
    const T r = balls.radii[i];
    int32_t *col = s_walk_lds + threadIdx.x;
    if (MODE == 0) {
        CountFound f;
        ball_walk<T, D>(t, pl_starts, pl_lists, bc, r, col, f);
        counts_or_starts[i] = f.n;
    } else if (MODE == 1) {
        WriteFound f{lists + counts_or_starts[i]};
        ball_walk<T, D>(t, pl_starts, pl_lists, bc, r, col, f);
    } else {
        InvaderFound<T, D> f{lists, bc};
        ball_walk<T, D>(t, pl_starts, pl_lists, bc, r, col, f);
    }
}

// ---- leaves-to-balls -------------------------------------------------------------

// entry j of the area query belongs to ball upper_bound(starts, j) - 1
__global__ __launch_bounds__(256) void expand_starts_kernel(const int32_t *starts, int64_t nballs,
        int64_t nentries, uint32_t *ball_of_entry)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= nentries) return;
    int64_t lo = 0, hi = nballs;        // last ball with starts[ball] <= j
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) starts[mid] <= j) lo = mid; else hi = mid;
    }
    ball_of_entry[j] = (uint32_t) lo;
}

__global__ __launch_bounds__(256) void key_starts_kernel(const uint32_t *sorted_keys, int64_t n,
        int64_t nkeys, int32_t *starts)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k > nkeys) return;
    int64_t lo = 0, hi = n;             // first entry with key >= k
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) sorted_keys[mid] < k) lo = mid + 1; else hi = mid;
    }
    starts[k] = (int32_t) lo;
}

// ---- host ------------------------------------------------------------------------

template <class T, int D>
struct Packed {
    Buf<Node<T, D>> nodes;
    Buf<int32_t> child_t;
    AqTree<T, D> t;
};

template <class T, int D>
int pack_tree(bt_context *ctx, const bt_aq_tree *p, Packed<T, D> &pk)
{
    constexpr int C = 1 << D;
    const int32_t nboxes = (int32_t) p->nboxes;
    BT_CHECK(pk.nodes.alloc(ctx->pool, nboxes));
    BT_CHECK(pk.child_t.alloc(ctx->pool, (int64_t) nboxes * C));
    pack_nodes_kernel<T, D><<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, p->aligned_nboxes, (const T *) p->box_centers, p->box_levels, p->box_flags,
            p->box_child_ids, pk.nodes.get(), pk.child_t.get());
    BT_HIP_CHECK(hipGetLastError());
    pk.t.nodes = pk.nodes.get();
    pk.t.child_t = pk.child_t.get();
    pk.t.parent = p->box_parent_ids;
    pk.t.nboxes = nboxes;
    pk.t.root_extent = (T) p->root_extent;
    for (int d = 0; d < D; ++d) pk.t.bbox_min[d] = (T) p->bbox_min[d];
    return BT_OK;
}

int check_tree_args(const bt_aq_tree *p, const char *who)
{
    if (!p || p->dims < 1 || p->dims > 3 || p->nboxes < 1 || p->nboxes >= (1 << 28)
            || p->nlevels < 1 || p->nlevels > BT_MAX_LEVELS || !p->box_centers || !p->box_levels
            || !p->box_flags || !p->box_child_ids
            || (p->coord_kind != BT_F32 && p->coord_kind != BT_F64)) {
        set_error("%s: invalid tree description", who);
        return BT_ERR_INVALID;
    }
    return BT_OK;
}

template <class T, int D>
int peer_lists_impl(bt_context *ctx, const bt_aq_tree *p, int64_t *n_entries)
{
    constexpr int P = pow3(D);
    static_assert(P <= 32, "compact_rows_kernel uses 32 lanes per row");
    if (!p->box_parent_ids || !p->level_start_box_nrs) {
        set_error("bt_peer_lists_build: box_parent_ids and level_start_box_nrs are required");
        return BT_ERR_INVALID;
    }
    const int32_t nboxes = (int32_t) p->nboxes;
    Packed<T, D> pk;
    BT_CHECK((pack_tree<T, D>(ctx, p, pk)));
    Buf<int32_t> rows, counts;
    BT_CHECK(rows.alloc(ctx->pool, (int64_t) nboxes * P));
    BT_CHECK(counts.alloc(ctx->pool, nboxes));
    BT_CHECK(reset_status(ctx));
    for (int lev = 0; lev < p->nlevels; ++lev) {
        const int32_t b0 = p->level_start_box_nrs[lev], b1 = p->level_start_box_nrs[lev + 1];
        if (b1 <= b0) continue;
        peer_level_kernel<T, D><<<(unsigned) div_up(b1 - b0, 256), 256, 0, ctx->stream>>>(
                pk.t, b0, b1, lev, rows.get(), counts.get(), ctx->d_status);
    }
    BT_HIP_CHECK(hipGetLastError());
    AqState *st = new AqState();
    bt_free_aq_state(ctx);
    ctx->aq = st;
    BT_CHECK(st->starts.alloc(ctx->pool, (int64_t) nboxes + 1));
    Buf<int64_t> d_total;
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, RowCount{counts.get()}, nboxes,
                                                      st->starts.get(), d_total.get(), true)));
    int64_t total = 0;
    BT_CHECK(bt::d2h(ctx, &total, d_total.get(), 8));
    BT_CHECK(check_status(ctx));          // syncs
    if (total > INT32_MAX) { set_error("peer lists: more than 2^31 entries"); return BT_ERR_INVALID; }
    BT_CHECK(st->lists.alloc(ctx->pool, total));
    compact_rows_kernel<P><<<(unsigned) div_up((int64_t) nboxes * 32, 256), 256, 0, ctx->stream>>>(
            nboxes, rows.get(), st->starts.get(), st->lists.get());
    BT_HIP_CHECK(hipGetLastError());
    st->nrows = nboxes;
    st->nentries = total;
    *n_entries = total;
    return BT_OK;
}

template <class T, int D>
int make_balls(const void *const *centers, const void *radii, int64_t n, Balls<T, D> &b)
{
    for (int d = 0; d < D; ++d) b.c[d] = (const T *) centers[d];
    b.radii = (const T *) radii;
    b.n = n;
    return BT_OK;
}

inline size_t walk_lds_bytes(int nlevels) { return (size_t) (nlevels + 1) * WALK_THREADS * 4; }

template <class T, int D>
int area_query_impl(bt_context *ctx, const bt_aq_tree *p, const int32_t *pl_starts,
                    const int32_t *pl_lists, int64_t nballs, const void *const *centers,
                    const void *radii, int64_t *n_entries)
{
    Packed<T, D> pk;
    BT_CHECK((pack_tree<T, D>(ctx, p, pk)));
    Balls<T, D> balls;
    make_balls<T, D>(centers, radii, nballs, balls);
    AqState *st = new AqState();
    bt_free_aq_state(ctx);
    ctx->aq = st;
    BT_CHECK(st->starts.alloc(ctx->pool, nballs + 1));
    Buf<int32_t> counts;
    BT_CHECK(counts.alloc(ctx->pool, nballs));
    const unsigned grid = (unsigned) div_up(nballs, WALK_THREADS);
    const size_t lds = walk_lds_bytes(p->nlevels);
    if (nballs > 0) {
        ball_kernel<T, D, 0><<<grid, WALK_THREADS, lds, ctx->stream>>>(
                pk.t, balls, pl_starts, pl_lists, counts.get(), nullptr);
        BT_HIP_CHECK(hipGetLastError());
    }
    Buf<int64_t> d_total;
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, RowCount{counts.get()}, nballs,
                                                      st->starts.get(), d_total.get(), true)));
    int64_t total = 0;
    BT_CHECK(bt::d2h(ctx, &total, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    if (total > INT32_MAX) { set_error("area query: more than 2^31 entries"); return BT_ERR_INVALID; }
    BT_CHECK(st->lists.alloc(ctx->pool, total));
    if (nballs > 0 && total > 0) {
        ball_kernel<T, D, 1><<<grid, WALK_THREADS, lds, ctx->stream>>>(
                pk.t, balls, pl_starts, pl_lists, st->starts.get(), st->lists.get());
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK(bt::sync_stream(ctx));    // pk's buffers die here
    st->nrows = nballs;
    st->nentries = total;
    *n_entries = total;
    return BT_OK;
}

template <class T, int D>
int space_invader_impl(bt_context *ctx, const bt_aq_tree *p, const int32_t *pl_starts,
                       const int32_t *pl_lists, int64_t nballs, const void *const *centers,
                       const void *radii, float *out)
{
    Packed<T, D> pk;
    BT_CHECK((pack_tree<T, D>(ctx, p, pk)));
    Balls<T, D> balls;
    make_balls<T, D>(centers, radii, nballs, balls);
    BT_HIP_CHECK(hipMemsetAsync(out, 0, (size_t) p->nboxes * 4, ctx->stream));
    if (nballs > 0) {
        ball_kernel<T, D, 2><<<(unsigned) div_up(nballs, WALK_THREADS), WALK_THREADS,
                               walk_lds_bytes(p->nlevels), ctx->stream>>>(
                pk.t, balls, pl_starts, pl_lists, nullptr, (int32_t *) out);
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

#define AQ_DISPATCH(p, CALL)                                                       \
    do {                                                                           \
        const bool f64_ = (p)->coord_kind == BT_F64;                               \
        switch ((p)->dims) {                                                       \
        case 1: return f64_ ? CALL(double, 1) : CALL(float, 1);                    \
        case 2: return f64_ ? CALL(double, 2) : CALL(float, 2);                    \
        default: return f64_ ? CALL(double, 3) : CALL(float, 3);                   \
        }                                                                          \
    } while (0)

}  // namespace

extern "C" {

int bt_peer_lists_build(bt_context *ctx, const bt_aq_tree *tree, int64_t *n_entries)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !n_entries) { set_error("bt_peer_lists_build: NULL argument"); return BT_ERR_INVALID; }
    BT_CHECK(check_tree_args(tree, "bt_peer_lists_build"));
    BT_HIP_CHECK(hipSetDevice(ctx->device));
#define CALL(T, D) peer_lists_impl<T, D>(ctx, tree, n_entries)
    AQ_DISPATCH(tree, CALL);
#undef CALL
}

int bt_area_query_build(bt_context *ctx, const bt_aq_tree *tree, const int32_t *peer_list_starts,
                        const int32_t *peer_lists, int64_t nballs,
                        const void *const *ball_centers, const void *ball_radii,
                        int64_t *n_entries)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !n_entries || !peer_list_starts || nballs < 0 || nballs > INT32_MAX
            || (nballs > 0 && (!ball_centers || !ball_radii))) {
        set_error("bt_area_query_build: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_CHECK(check_tree_args(tree, "bt_area_query_build"));
    BT_HIP_CHECK(hipSetDevice(ctx->device));
#define CALL(T, D) area_query_impl<T, D>(ctx, tree, peer_list_starts, peer_lists, nballs, \
                                         ball_centers, ball_radii, n_entries)
    AQ_DISPATCH(tree, CALL);
#undef CALL
}

int bt_csr_export(bt_context *ctx, int32_t *starts, int32_t *lists)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !ctx->aq) { set_error("bt_csr_export: nothing was built"); return BT_ERR_INVALID; }
    AqState *st = ctx->aq;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (starts)
        BT_HIP_CHECK(hipMemcpyAsync(starts, st->starts.get(), (size_t) (st->nrows + 1) * 4,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    if (lists && st->nentries > 0)
        BT_HIP_CHECK(hipMemcpyAsync(lists, st->lists.get(), (size_t) st->nentries * 4,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    BT_CHECK(bt::sync_stream(ctx));
    bt_free_aq_state(ctx);
    return BT_OK;
}

int bt_space_invader_query(bt_context *ctx, const bt_aq_tree *tree,
                           const int32_t *peer_list_starts, const int32_t *peer_lists,
                           int64_t nballs, const void *const *ball_centers,
                           const void *ball_radii, float *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !out || !peer_list_starts || nballs < 0
            || (nballs > 0 && (!ball_centers || !ball_radii))) {
        set_error("bt_space_invader_query: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_CHECK(check_tree_args(tree, "bt_space_invader_query"));
    BT_HIP_CHECK(hipSetDevice(ctx->device));
#define CALL(T, D) space_invader_impl<T, D>(ctx, tree, peer_list_starts, peer_lists, nballs, \
                                            ball_centers, ball_radii, out)
    AQ_DISPATCH(tree, CALL);
#undef CALL
}

int bt_leaves_to_balls(bt_context *ctx, int64_t nballs, int64_t nboxes,
                       const int32_t *leaves_near_ball_starts,
                       const int32_t *leaves_near_ball_lists, int64_t n_entries,
                       int32_t *balls_near_box_starts, int32_t *balls_near_box_lists)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nballs < 0 || nboxes < 1 || n_entries < 0 || !balls_near_box_starts
            || !leaves_near_ball_starts || (n_entries > 0 && (!leaves_near_ball_lists
                                                                || !balls_near_box_lists))) {
        set_error("bt_leaves_to_balls: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t *sorted_keys = nullptr;
    Buf<uint32_t> ka, kb, va, vb;
    if (n_entries > 0) {
        BT_CHECK(ka.alloc(ctx->pool, n_entries));
        BT_CHECK(kb.alloc(ctx->pool, n_entries));
        BT_CHECK(va.alloc(ctx->pool, n_entries));
        BT_CHECK(vb.alloc(ctx->pool, n_entries));
        BT_HIP_CHECK(hipMemcpyAsync(ka.get(), leaves_near_ball_lists, (size_t) n_entries * 4,
                                    hipMemcpyDeviceToDevice, ctx->stream));
        expand_starts_kernel<<<(unsigned) div_up(n_entries, 256), 256, 0, ctx->stream>>>(
                leaves_near_ball_starts, nballs, n_entries, va.get());
        BT_HIP_CHECK(hipGetLastError());
        int bits = 1;
        while (bits < 32 && (1ll << bits) < nboxes) ++bits;
        bool in_b = false;
        BT_CHECK(radix_sort_pairs<uint32_t>(ctx, ka.get(), va.get(), kb.get(), vb.get(), n_entries,
                                            0, bits, false, &in_b));
        sorted_keys = in_b ? kb.get() : ka.get();
        BT_HIP_CHECK(hipMemcpyAsync(balls_near_box_lists, in_b ? vb.get() : va.get(),
                                    (size_t) n_entries * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    key_starts_kernel<<<(unsigned) div_up(nboxes + 1, 256), 256, 0, ctx->stream>>>(
            sorted_keys, n_entries, nboxes, balls_near_box_starts);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"
