// Per-particle post-processing of a built tree: target filtering
// (boxtree/tree.py:1059-1243) and point-source linking (tree.py:772-949).
// All of it is prefix sums over the tree order plus index arithmetic.
#include "bt_common.hpp"
#include "bt_prims.hpp"

using namespace bt;

namespace {

// tree_order_flags[sorted_target_ids[k]] = flags[k] (tree.py:1184-1185) and the
// inverse permutation user_target_ids[sorted_target_ids[k]] = k (:1126-1129)
__global__ __launch_bounds__(256) void tree_order_flags_kernel(int64_t n, const int8_t *flags,
        const int32_t *sorted_target_ids, uint8_t *tflags, int32_t *inverse)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int32_t j = sorted_target_ids[k];
    tflags[j] = flags[k] != 0;
    if (inverse) inverse[j] = (int32_t) k;
}

struct FlagCount {
    const uint8_t *f;
    __device__ __forceinline__ int32_t operator()(int64_t i) const { return f[i]; }
};

// filtered index of unfiltered tree-order position j (tbk:1966); j == ntargets
// maps to nfiltered (the reference guards only the end of a box, tbk:2003-2011)
__device__ __forceinline__ int32_t filtered_at(const int32_t *F, int64_t j, int64_t n)
{
    return F[j <= n ? j : n];       // F has n+1 entries, F[n] = nfiltered
}

struct BoxFilteredCount {
    const int32_t *F, *starts, *counts;
    int64_t n;
    __device__ __forceinline__ int32_t operator()(int64_t b) const
    {
        const int64_t s = starts[b], c = counts[b];
        return c > 0 ? filtered_at(F, s + c, n) - filtered_at(F, s, n) : 0;
    }
};

// user-order lists: one wave per box, every flagged target lands at the box's
// list start plus its rank among the box's flagged targets
__global__ __launch_bounds__(256) void fill_user_lists_kernel(int64_t nboxes, int64_t n,
        const int32_t *F, const uint8_t *tflags, const int32_t *inverse,
        const int32_t *box_starts, const int32_t *box_counts, const int32_t *list_starts,
        int32_t *lists)
{
    const int64_t b = ((int64_t) blockIdx.x * 256 + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (b >= nboxes) return;
    const int64_t s = box_starts[b], c = box_counts[b];
    if (c <= 0) return;
    const int32_t base = list_starts[b] - F[s];
    for (int64_t j = s + lane; j < s + c; j += WAVE)
        if (tflags[j]) lists[base + F[j]] = inverse[j];
}

// tbk:1967-1968
__global__ __launch_bounds__(256) void unfiltered_from_filtered_kernel(int64_t n, const int32_t *F,
        const uint8_t *tflags, int32_t *out)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    if (tflags[j]) out[F[j]] = (int32_t) j;
}

// TREE_ORDER_TARGET_FILTER_INDEX_TPL: tbk:1990-2018
__global__ __launch_bounds__(256) void filtered_box_index_kernel(int64_t nboxes, int64_t n,
        const int32_t *F, const int32_t *box_starts, const int32_t *box_counts,
        int32_t *starts_f, int32_t *counts_f)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    const int64_t s = box_starts[b], c = box_counts[b];
    const int32_t fs = filtered_at(F, s, n);
    starts_f[b] = fs;
    counts_f[b] = c > 0 ? filtered_at(F, s + c, n) - fs : 0;
}

int tree_order_scan(bt_context *ctx, int64_t n, const int8_t *flags,
                    const int32_t *sorted_target_ids, Buf<uint8_t> &tflags, Buf<int32_t> &F,
                    Buf<int32_t> *inverse, int64_t *nfiltered)
{
    BT_CHECK(tflags.alloc(ctx->pool, n));
    BT_CHECK(F.alloc(ctx->pool, n + 1));
    if (inverse) BT_CHECK(inverse->alloc(ctx->pool, n));
    if (n > 0) {
        tree_order_flags_kernel<<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(
                n, flags, sorted_target_ids, tflags.get(), inverse ? inverse->get() : nullptr);
        BT_HIP_CHECK(hipGetLastError());
    }
    Buf<int64_t> d_total;
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, FlagCount{tflags.get()}, n, F.get(),
                                                      d_total.get(), true)));
    BT_CHECK(bt::d2h(ctx, nfiltered, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

// ---- point sources ---------------------------------------------------------------

struct PointSourceLength {      // tbk:1884-1887
    const int32_t *pss, *usi;
    __device__ __forceinline__ int32_t operator()(int64_t i) const
    {
        const int32_t u = usi[i];
        return pss[u + 1] - pss[u];
    }
};

__global__ __launch_bounds__(256) void point_source_counts_kernel(int64_t nsources,
        const int32_t *pss, const int32_t *usi, int32_t *counts)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= nsources) return;
    const int32_t u = usi[i];
    counts[i] = pss[u + 1] - pss[u];
}

// user_point_source_ids (tree.py:842-893): point p belongs to the last source i
// with tree_order_start[i] <= p and is number p - start[i] of that source
__global__ __launch_bounds__(256) void user_point_source_ids_kernel(int64_t npoints,
        int64_t nsources, const int32_t *to_starts, const int32_t *pss, const int32_t *usi,
        int32_t *ids)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= npoints) return;
    int64_t lo = 0, hi = nsources;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) to_starts[mid] <= p) lo = mid; else hi = mid;
    }
    ids[p] = pss[usi[lo]] + (int32_t) (p - to_starts[lo]);
}

// POINT_SOURCE_LINKING_BOX_POINT_SOURCES: tbk:1914-1947
__global__ __launch_bounds__(256) void box_point_sources_kernel(int64_t nboxes, int64_t nsources,
        int32_t npoints, const int32_t *box_source_starts, const int32_t *nonchild,
        const int32_t *cumul, const int32_t *to_starts, const int32_t *to_counts,
        int32_t *out_starts, int32_t *out_nonchild, int32_t *out_cumul)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    const int64_t s = box_source_starts[b];
    const int32_t ps = s < nsources ? to_starts[s] : npoints;
    out_starts[b] = ps;
    const int32_t cn = nonchild[b], cc = cumul[b];
    out_nonchild[b] = cn ? to_starts[s + cn - 1] + to_counts[s + cn - 1] - ps : 0;
    out_cumul[b] = cc ? to_starts[s + cc - 1] + to_counts[s + cc - 1] - ps : 0;
}

}  // namespace

extern "C" {

int bt_filter_targets_user_order(bt_context *ctx, int64_t nboxes, int64_t ntargets,
                                 const int8_t *user_order_flags, const int32_t *sorted_target_ids,
                                 const int32_t *box_target_starts,
                                 const int32_t *box_target_counts_nonchild,
                                 int32_t *target_starts, int32_t *target_lists,
                                 int64_t *nfiltered)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || ntargets < 0 || ntargets > INT32_MAX || !box_target_starts
            || !box_target_counts_nonchild || !target_starts || !nfiltered
            || (ntargets > 0 && (!user_order_flags || !sorted_target_ids || !target_lists))) {
        set_error("bt_filter_targets_user_order: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<uint8_t> tflags;
    Buf<int32_t> F, inverse;
    BT_CHECK(tree_order_scan(ctx, ntargets, user_order_flags, sorted_target_ids, tflags, F,
                             &inverse, nfiltered));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(
            ctx, BoxFilteredCount{F.get(), box_target_starts, box_target_counts_nonchild, ntargets},
            nboxes, target_starts, (int64_t *) nullptr, true)));
    if (*nfiltered > 0) {
        fill_user_lists_kernel<<<(unsigned) div_up(nboxes * WAVE, 256), 256, 0, ctx->stream>>>(
                nboxes, ntargets, F.get(), tflags.get(), inverse.get(), box_target_starts,
                box_target_counts_nonchild, target_starts, target_lists);
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_filter_targets_tree_order(bt_context *ctx, int64_t nboxes, int64_t ntargets,
                                 const int8_t *user_order_flags, const int32_t *sorted_target_ids,
                                 const int32_t *box_target_starts,
                                 const int32_t *box_target_counts_nonchild,
                                 int32_t *box_target_starts_filtered,
                                 int32_t *box_target_counts_nonchild_filtered,
                                 int32_t *unfiltered_from_filtered, int64_t *nfiltered)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nboxes < 1 || ntargets < 0 || ntargets > INT32_MAX || !box_target_starts
            || !box_target_counts_nonchild || !box_target_starts_filtered
            || !box_target_counts_nonchild_filtered || !nfiltered
            || (ntargets > 0 && (!user_order_flags || !sorted_target_ids
                                 || !unfiltered_from_filtered))) {
        set_error("bt_filter_targets_tree_order: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<uint8_t> tflags;
    Buf<int32_t> F;
    BT_CHECK(tree_order_scan(ctx, ntargets, user_order_flags, sorted_target_ids, tflags, F, nullptr,
                             nfiltered));
    if (ntargets > 0) {
        unfiltered_from_filtered_kernel<<<(unsigned) div_up(ntargets, 256), 256, 0, ctx->stream>>>(
                ntargets, F.get(), tflags.get(), unfiltered_from_filtered);
        BT_HIP_CHECK(hipGetLastError());
    }
    filtered_box_index_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, ntargets, F.get(), box_target_starts, box_target_counts_nonchild,
            box_target_starts_filtered, box_target_counts_nonchild_filtered);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_link_point_sources(bt_context *ctx, int64_t nsources, int64_t nboxes,
                          int64_t npoint_sources, const int32_t *point_source_starts,
                          const int32_t *user_source_ids, const int32_t *box_source_starts,
                          const int32_t *box_source_counts_nonchild,
                          const int32_t *box_source_counts_cumul,
                          int32_t *tree_order_point_source_starts,
                          int32_t *tree_order_point_source_counts,
                          int32_t *user_point_source_ids, int32_t *box_point_source_starts,
                          int32_t *box_point_source_counts_nonchild,
                          int32_t *box_point_source_counts_cumul)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nsources < 1 || nboxes < 1 || npoint_sources < 0 || npoint_sources > INT32_MAX
            || !point_source_starts || !user_source_ids || !box_source_starts
            || !box_source_counts_nonchild || !box_source_counts_cumul
            || !tree_order_point_source_starts || !tree_order_point_source_counts
            || !box_point_source_starts || !box_point_source_counts_nonchild
            || !box_point_source_counts_cumul || (npoint_sources > 0 && !user_point_source_ids)) {
        set_error("bt_link_point_sources: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<int64_t> d_total;
    BT_CHECK(d_total.alloc(ctx->pool, 1));
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(
            ctx, PointSourceLength{point_source_starts, user_source_ids}, nsources,
            tree_order_point_source_starts, d_total.get(), false)));
    point_source_counts_kernel<<<(unsigned) div_up(nsources, 256), 256, 0, ctx->stream>>>(
            nsources, point_source_starts, user_source_ids, tree_order_point_source_counts);
    BT_HIP_CHECK(hipGetLastError());
    int64_t total = 0;
    BT_CHECK(bt::d2h(ctx, &total, d_total.get(), 8));
    BT_CHECK(bt::sync_stream(ctx));
    if (total != npoint_sources) {
        set_error("bt_link_point_sources: point_source_starts describes %lld point sources, "
                  "the caller announced %lld", (long long) total, (long long) npoint_sources);
        return BT_ERR_INVALID;
    }
    if (npoint_sources > 0) {
        user_point_source_ids_kernel<<<(unsigned) div_up(npoint_sources, 256), 256, 0,
                                       ctx->stream>>>(
                npoint_sources, nsources, tree_order_point_source_starts, point_source_starts,
                user_source_ids, user_point_source_ids);
        BT_HIP_CHECK(hipGetLastError());
    }
    box_point_sources_kernel<<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
            nboxes, nsources, (int32_t) npoint_sources, box_source_starts,
            box_source_counts_nonchild, box_source_counts_cumul, tree_order_point_source_starts,
            tree_order_point_source_counts, box_point_source_starts,
            box_point_source_counts_nonchild, box_point_source_counts_cumul);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"
