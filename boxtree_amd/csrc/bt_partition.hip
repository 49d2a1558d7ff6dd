// Work partition and local trees for the distributed FMM evaluation: the
// device side of boxtree/distributed/partition.py, local_tree.py and of the
// box selection in calculation.py.  Everything is masks over box numbers,
// prefix sums and index arithmetic on a tree that is already built.
#include "bt_common.hpp"
#include "bt_prims.hpp"

#include <vector>

using namespace bt;

namespace {

// ---- depth-first order (distributed/partition.py:39-57) ---------------------------

// boxes of one level, deepest level first: size of the subtree under each box
__global__ __launch_bounds__(256) void subtree_size_kernel(int32_t lo, int32_t hi, int nchildren,
        int64_t stride, const int32_t *child_ids, int32_t *size)
{
    const int32_t b = lo + (int32_t) (blockIdx.x * 256 + threadIdx.x);
    if (b >= hi) return;
    int32_t s = 1;
    for (int c = 0; c < nchildren; ++c) {
        const int32_t ch = child_ids[(int64_t) c * stride + b];
        if (ch > 0) s += size[ch];
    }
    size[b] = s;
}

// top level first: the reference pops children in descending child number
// (the stack is filled ascending, partition.py:52-56)
__global__ __launch_bounds__(256) void dfs_position_kernel(int32_t lo, int32_t hi, int nchildren,
        int64_t stride, const int32_t *child_ids, const int32_t *size, int32_t *pos,
        int32_t *order)
{
    const int32_t b = lo + (int32_t) (blockIdx.x * 256 + threadIdx.x);
    if (b >= hi) return;
    const int32_t p = pos[b];
    order[p] = b;
    int32_t run = p + 1;
    for (int c = nchildren - 1; c >= 0; --c) {
        const int32_t ch = child_ids[(int64_t) c * stride + b];
        if (ch > 0) {
            pos[ch] = run;
            run += size[ch];
        }
    }
}

// ---- partition_work (distributed/partition.py:60-121) ------------------------------

struct CostInDfsOrder {
    const double *cost;
    const int32_t *dfs;
    __device__ __forceinline__ double operator()(int64_t i) const { return cost[dfs[i]]; }
};

// first depth-first position whose running cost exceeds thr[s] (n if none);
// E is the exclusive scan with the total at E[n], so the running cost after
// position i is E[i+1]
__global__ void first_exceeding_kernel(int nthr, const double *thr, const double *E, int64_t n,
                                       int64_t *first)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nthr) return;
    const double t = thr[s];
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (E[mid + 1] > t) hi = mid; else lo = mid + 1;
    }
    first[s] = lo;
}

// ---- masks (distributed/partition.py:124-318) --------------------------------------

// every box of the mask walks to the root; it stops at the first ancestor that is
// already marked, whose marker has walked (or is walking) the rest of the chain
__global__ __launch_bounds__(256) void ancestor_mask_kernel(int64_t nboxes,
        const int32_t *parent_ids, const int8_t *mask, int8_t *ancestors)
{
    const int64_t b0 = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b0 >= nboxes || !mask[b0]) return;
    int32_t b = (int32_t) b0;
    while (b != 0) {
        b = parent_ids[b];
        // plain byte accesses: a stale 0 only costs a redundant walk
        if (ancestors[b]) break;
        ancestors[b] = 1;
    }
}

constexpr int ROW_LANES = 16;

__global__ __launch_bounds__(256) void mark_list_boxes_kernel(int64_t nrows,
        const int32_t *box_list, const int8_t *mask_a, const int8_t *mask_b,
        const int32_t *starts, const int32_t *lists, int8_t *out)
{
    const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t / ROW_LANES;
    const int lane = (int) (t % ROW_LANES);
    if (row >= nrows) return;
    const int32_t b = box_list[row];
    if (!(mask_a[b] || (mask_b && mask_b[b]))) return;
    const int32_t e = starts[row + 1];
    for (int32_t j = starts[row] + lane; j < e; j += ROW_LANES) out[lists[j]] = 1;
}

// ---- local particles (distributed/local_tree.py:198-283) ---------------------------

__global__ __launch_bounds__(256) void particle_mask_kernel(int64_t nboxes, const int8_t *box_mask,
        const int32_t *starts, const int32_t *counts_nonchild, uint8_t *pmask)
{
    const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t b = t / ROW_LANES;
    const int lane = (int) (t % ROW_LANES);
    if (b >= nboxes || !box_mask[b]) return;
    const int64_t s = starts[b], e = s + counts_nonchild[b];
    for (int64_t j = s + lane; j < e; j += ROW_LANES) pmask[j] = 1;
}

struct ByteCount {
    const uint8_t *f;
    __device__ __forceinline__ int32_t operator()(int64_t i) const { return f[i]; }
};

__global__ __launch_bounds__(256) void local_lists_kernel(int64_t nboxes, int64_t n,
        const int32_t *F, const int8_t *box_mask, const int32_t *starts,
        const int32_t *counts_nonchild, const int32_t *counts_cumul, int32_t *lstarts,
        int32_t *lnonchild, int32_t *lcumul)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    int64_t s = starts[b], e = s + counts_cumul[b];
    s = s < n ? s : n;
    e = e < n ? e : n;
    lstarts[b] = F[s];
    lnonchild[b] = box_mask[b] ? counts_nonchild[b] : 0;
    lcumul[b] = F[e] - F[s];
}

__global__ __launch_bounds__(256) void particle_idx_kernel(int64_t n, const int32_t *F,
        const uint8_t *pmask, int32_t *idx)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    if (pmask[j]) idx[F[j]] = (int32_t) j;
}

__global__ __launch_bounds__(256) void modify_target_flags_kernel(int64_t nboxes,
        const int32_t *nonchild, const int32_t *cumul, uint8_t *flags)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    uint8_t f = flags[b] & (uint8_t) ~(BT_BOX_IS_TARGET_BOX | BT_BOX_HAS_TARGET_CHILD_BOXES);
    if (nonchild[b]) f |= BT_BOX_IS_TARGET_BOX;
    if (nonchild[b] < cumul[b]) f |= BT_BOX_HAS_TARGET_CHILD_BOXES;
    flags[b] = f;
}

// ---- users of a box's multipole expansion (local_tree.py:368-399) ------------------

struct UserRankCount {
    const int8_t *masks;
    int64_t nboxes;
    int nranks;
    __device__ __forceinline__ int32_t operator()(int64_t b) const
    {
        int32_t c = 0;
        for (int r = 0; r < nranks; ++r) c += masks[(int64_t) r * nboxes + b] != 0;
        return c;
    }
};

__global__ __launch_bounds__(256) void user_rank_fill_kernel(int64_t nboxes, int nranks,
        const int8_t *masks, const int32_t *starts, int32_t *lists)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    int32_t w = starts[b];
    for (int r = 0; r < nranks; ++r)
        if (masks[(int64_t) r * nboxes + b]) lists[w++] = r;
}

// calculation.py:191-262
__global__ __launch_bounds__(256) void used_by_ranks_kernel(int64_t nboxes,
        const int8_t *contributing, int lo, int hi, const int32_t *ustarts,
        const int32_t *ulists, uint8_t *flag)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    uint8_t f = 0;
    if (contributing[b]) {
        const int32_t e = ustarts[b + 1];
        for (int32_t j = ustarts[b]; j < e; ++j) {
            const int32_t u = ulists[j];
            if (lo <= u && u < hi) { f = 1; break; }
        }
    }
    flag[b] = f;
}

}  // namespace

extern "C" {

int bt_dfs_order(bt_context *ctx, int nchildren, int nlevels, const int32_t *level_start_box_nrs,
                 int64_t nboxes, int64_t aligned_nboxes, const int32_t *box_child_ids,
                 int32_t *dfs_order)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nchildren < 2 || nlevels < 1 || !level_start_box_nrs || nboxes < 1
            || nboxes > INT32_MAX || aligned_nboxes < nboxes || !box_child_ids || !dfs_order
            || level_start_box_nrs[0] != 0 || level_start_box_nrs[nlevels] != nboxes) {
        set_error("bt_dfs_order: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<int32_t> size, pos;
    BT_CHECK(size.alloc(ctx->pool, nboxes));
    BT_CHECK(pos.alloc(ctx->pool, nboxes));
    BT_HIP_CHECK(hipMemsetAsync(pos.get(), 0, 4, ctx->stream));
    for (int l = nlevels - 1; l >= 0; --l) {
        const int32_t lo = level_start_box_nrs[l], hi = level_start_box_nrs[l + 1];
        if (hi <= lo) continue;
        subtree_size_kernel<<<(unsigned) div_up(hi - lo, 256), 256, 0, ctx->stream>>>(
                lo, hi, nchildren, aligned_nboxes, box_child_ids, size.get());
    }
    for (int l = 0; l < nlevels; ++l) {
        const int32_t lo = level_start_box_nrs[l], hi = level_start_box_nrs[l + 1];
        if (hi <= lo) continue;
        dfs_position_kernel<<<(unsigned) div_up(hi - lo, 256), 256, 0, ctx->stream>>>(
                lo, hi, nchildren, aligned_nboxes, box_child_ids, size.get(), pos.get(),
                dfs_order);
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_partition_work(bt_context *ctx, int64_t nboxes, const int32_t *dfs_order,
                      const double *cost_per_box, int nranks, int32_t *segments)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || nboxes > INT32_MAX || !dfs_order || !cost_per_box || nranks < 1
            || !segments) {
        set_error("bt_partition_work: invalid argument");
        return BT_ERR_INVALID;
    }
    if (nranks > nboxes) {      // partition.py:77-79, RuntimeError upstream
        set_error("bt_partition_work: fewer boxes than ranks");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<double> E, d_total, d_thr;
    Buf<int64_t> d_first;
    BT_CHECK(E.alloc(ctx->pool, nboxes + 1));
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<double, double>(ctx, CostInDfsOrder{cost_per_box, dfs_order},
                                                    nboxes, E.get(), d_total.get(), true)));
    double total = 0;
    BT_CHECK(bt::d2h(ctx, &total, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));

    std::vector<double> thr((size_t) nranks);
    std::vector<int64_t> first((size_t) nranks, nboxes);
    for (int s = 0; s < nranks; ++s)      // (segment_idx + 1) * total_workload / mpi_size
        thr[(size_t) s] = (double) (s + 1) * total / (double) nranks;
    BT_CHECK(d_thr.alloc(ctx->pool, nranks));
    BT_CHECK(d_first.alloc(ctx->pool, nranks));
    BT_HIP_CHECK(hipMemcpyAsync(d_thr.get(), thr.data(), 8 * (size_t) nranks,
                                hipMemcpyHostToDevice, ctx->stream));
    first_exceeding_kernel<<<(unsigned) div_up(nranks, 64), 64, 0, ctx->stream>>>(
            nranks, d_thr.get(), E.get(), nboxes, d_first.get());
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::d2h(ctx, first.data(), d_first.get(), 8 * (size_t) nranks));
    BT_CHECK(bt::sync_stream(ctx));

    // the reference's loop (partition.py:99-116) visits the boxes one by one and
    // lets a box end at most one segment; replayed here segment by segment
    for (int s = 0; s < nranks; ++s) segments[2 * s] = segments[2 * s + 1] = (int32_t) nboxes;
    int64_t i = 0, start = 0;
    for (int seg = 0; seg < nranks && i < nboxes; ++seg) {
        if (seg + 1 == nranks) {
            segments[2 * seg] = (int32_t) start;
            segments[2 * seg + 1] = (int32_t) nboxes;
            break;
        }
        int64_t p = first[(size_t) seg] > i ? first[(size_t) seg] : i;
        if (p > nboxes - 1) p = nboxes - 1;
        segments[2 * seg] = (int32_t) start;
        segments[2 * seg + 1] = (int32_t) (p + 1);
        start = i = p + 1;
    }
    return BT_OK;
}

int bt_ancestor_mask(bt_context *ctx, int64_t nboxes, const int32_t *box_parent_ids,
                     const int8_t *boxes_mask, int8_t *ancestors)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || !box_parent_ids || !boxes_mask || !ancestors) {
        set_error("bt_ancestor_mask: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_HIP_CHECK(hipMemsetAsync(ancestors, 0, (size_t) nboxes, ctx->stream));
    ancestor_mask_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, box_parent_ids, boxes_mask, ancestors);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_mark_list_boxes(bt_context *ctx, int64_t nrows, const int32_t *box_list,
                       const int8_t *mask_a, const int8_t *mask_b, const int32_t *starts,
                       const int32_t *lists, int8_t *out_mask)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nrows < 0 || !mask_a || !out_mask || (nrows > 0 && (!box_list || !starts))) {
        set_error("bt_mark_list_boxes: invalid argument");
        return BT_ERR_INVALID;
    }
    if (nrows == 0) return BT_OK;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    mark_list_boxes_kernel<<<(unsigned) div_up(nrows * ROW_LANES, 256), 256, 0, ctx->stream>>>(
            nrows, box_list, mask_a, mask_b, starts, lists, out_mask);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_local_particles(bt_context *ctx, int64_t nboxes, int64_t nparticles, const int8_t *box_mask,
                       const int32_t *box_particle_starts,
                       const int32_t *box_particle_counts_nonchild,
                       const int32_t *box_particle_counts_cumul, int32_t *local_starts,
                       int32_t *local_counts_nonchild, int32_t *local_counts_cumul,
                       int32_t *particle_idx, int64_t *nlocal)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || nparticles < 0 || nparticles > INT32_MAX || !box_mask
            || !box_particle_starts || !box_particle_counts_nonchild || !box_particle_counts_cumul
            || !local_starts || !local_counts_nonchild || !local_counts_cumul || !nlocal
            || (nparticles > 0 && !particle_idx)) {
        set_error("bt_local_particles: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<uint8_t> pmask;
    Buf<int32_t> F;
    Buf<int64_t> d_total;
    BT_CHECK(pmask.alloc(ctx->pool, nparticles));
    BT_CHECK(F.alloc(ctx->pool, nparticles + 1));
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    if (nparticles > 0) {
        BT_HIP_CHECK(hipMemsetAsync(pmask.get(), 0, (size_t) nparticles, ctx->stream));
        particle_mask_kernel<<<(unsigned) div_up(nboxes * ROW_LANES, 256), 256, 0, ctx->stream>>>(
                nboxes, box_mask, box_particle_starts, box_particle_counts_nonchild, pmask.get());
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, ByteCount{pmask.get()}, nparticles,
                                                      F.get(), d_total.get(), true)));
    local_lists_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, nparticles, F.get(), box_mask, box_particle_starts,
            box_particle_counts_nonchild, box_particle_counts_cumul, local_starts,
            local_counts_nonchild, local_counts_cumul);
    if (nparticles > 0)
        particle_idx_kernel<<<(unsigned) div_up(nparticles, 256), 256, 0, ctx->stream>>>(
                nparticles, F.get(), pmask.get(), particle_idx);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::d2h(ctx, nlocal, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_modify_target_flags(bt_context *ctx, int64_t nboxes, const int32_t *counts_nonchild,
                           const int32_t *counts_cumul, uint8_t *box_flags)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || !counts_nonchild || !counts_cumul || !box_flags) {
        set_error("bt_modify_target_flags: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    modify_target_flags_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, counts_nonchild, counts_cumul, box_flags);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_box_to_user_ranks(bt_context *ctx, int nranks, int64_t nboxes, const int8_t *masks,
                         int32_t *starts, int32_t *lists, int64_t *nentries)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nranks < 1 || nboxes < 1 || !masks || !starts || !nentries) {
        set_error("bt_box_to_user_ranks: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (!lists) {
        Buf<int64_t> d_total;
        BT_CHECK(d_total.alloc(ctx->pool, 1));
        BT_CHECK((device_exclusive_scan<int64_t, int32_t>(
                ctx, UserRankCount{masks, nboxes, nranks}, nboxes, starts, d_total.get(), true)));
        BT_CHECK(bt::d2h(ctx, nentries, d_total.get(), 8));
        BT_CHECK(bt::sync_stream(ctx));
        if (*nentries > INT32_MAX) {
            set_error("bt_box_to_user_ranks: list exceeds the int32 CSR limit");
            return BT_ERR_INVALID;
        }
        return BT_OK;
    }
    user_rank_fill_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, nranks, masks, starts, lists);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_boxes_used_by_ranks(bt_context *ctx, int64_t nboxes, const int8_t *contributing,
                           int rank_lo, int rank_hi, const int32_t *box_to_user_rank_starts,
                           const int32_t *box_to_user_rank_lists, int32_t *boxes, int64_t *n)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || !contributing || !box_to_user_rank_starts || !boxes || !n) {
        set_error("bt_boxes_used_by_ranks: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<uint8_t> flag;
    Buf<int32_t> F;
    Buf<int64_t> d_total;
    BT_CHECK(flag.alloc(ctx->pool, nboxes));
    BT_CHECK(F.alloc(ctx->pool, nboxes + 1));
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    used_by_ranks_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, contributing, rank_lo, rank_hi, box_to_user_rank_starts,
            box_to_user_rank_lists, flag.get());
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, ByteCount{flag.get()}, nboxes, F.get(),
                                                      d_total.get(), true)));
    particle_idx_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, F.get(), flag.get(), boxes);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::d2h(ctx, n, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"
