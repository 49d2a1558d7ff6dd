// Device kernels of the constant-one expansion wrangler (boxtree/constant_one.py:49-237)
// that drive_fmm (boxtree/fmm.py:342-532) calls: every "expansion" is one float64
// per box, every translation a sum.  Used to check interaction lists for
// completeness at full problem size (each target must receive the total source
// weight exactly once).
#include "bt_common.hpp"
#include "bt_prims.hpp"

using namespace bt;

namespace {

// out[box] (+)= sum of values[start[box] : start[box]+count[box]]; one wave per listed box
__global__ __launch_bounds__(256) void box_particle_sum_kernel(int64_t n, const int32_t *boxes,
        const int32_t *starts, const int32_t *counts, const double *values, double *out,
        int accumulate)
{
    const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (i >= n) return;
    const int32_t b = boxes ? boxes[i] : (int32_t) i;
    const int64_t s = starts[b], c = counts[b];
    double acc = 0;
    for (int64_t j = lane; j < c; j += WAVE) acc += values[s + j];
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) out[b] = accumulate ? out[b] + acc : acc;
}

// out[i] = sum over row i of box_values[lists[.]]; 8 lanes per row
__global__ __launch_bounds__(256) void csr_sum_kernel(int64_t nrows, const int32_t *starts,
        const int32_t *lists, const double *box_values, double *out)
{
    const int64_t gid = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int64_t i = gid >> 3;
    const int lane = (int) (gid & 7);
    double acc = 0;
    if (i < nrows) {
        const int32_t s = starts[i], e = starts[i + 1];
        for (int32_t j = s + lane; j < e; j += 8) acc += box_values[lists[j]];
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 8);
    if (i < nrows && lane == 0) out[i] = acc;
}

// dst[row_boxes[i]] += row_values[i]
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int64_t nrows, const int32_t *row_boxes,
        const double *row_values, double *dst)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < nrows) dst[row_boxes[i]] += row_values[i];      // row boxes are distinct
}

// pot[start[b] : start[b]+count[b]] (+)= row_values[i] (or box_values[b]); one wave per row
__global__ __launch_bounds__(256) void box_to_particles_kernel(int64_t nrows, const int32_t *row_boxes,
        const int32_t *starts, const int32_t *counts, const double *row_values,
        const double *box_values, double *pot, int accumulate)
{
    const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (i >= nrows) return;
    const int32_t b = row_boxes[i];
    const double v = row_values ? row_values[i] : box_values[b];
    const int64_t s = starts[b], c = counts[b];
    for (int64_t j = lane; j < c; j += WAVE) pot[s + j] = accumulate ? pot[s + j] + v : v;
}

// coarsen_multipoles (constant_one.py:102-123): box += sum of its children
__global__ __launch_bounds__(256) void add_children_kernel(int64_t n, const int32_t *boxes,
        const int32_t *child_ids, int64_t aligned, int nchildren, double *vals)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = boxes[i];
    double acc = vals[b];
    for (int m = 0; m < nchildren; ++m) {
        const int32_t c = child_ids[(int64_t) m * aligned + b];
        if (c) acc += vals[c];
    }
    vals[b] = acc;
}

// refine_locals (constant_one.py:214-223): box += parent
__global__ __launch_bounds__(256) void add_parent_kernel(int64_t n, const int32_t *boxes,
        const int32_t *parent_ids, double *vals)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t b = boxes[i];
    vals[b] += vals[parent_ids[b]];
}


// TRANSLATION_CLASS_FINDER_TEMPLATE (translation_classes.py:62-189): one thread per
// list-2 entry
template <class T, int D>
__global__ __launch_bounds__(256) void translation_class_kernel(int64_t n, const int32_t *lists,
        const int32_t *starts, const int32_t *ttp_boxes, int64_t nttp, const T *box_centers,
        int64_t aligned, T root_extent, const uint8_t *box_levels, int nway, int per_level,
        int32_t *classes, int32_t *class_is_used, int32_t *error_flag)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int32_t source_box_id = lists[i];
    int64_t lo = 0, hi = nttp;          // last row with starts[row] <= i
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t) starts[mid] <= i) lo = mid; else hi = mid;
    }
    const int32_t target_box_id = ttp_boxes[lo];
    const int level = box_levels[source_box_id];
    if (level != (int) box_levels[target_box_id]) { atomicOr(error_flag, 1); return; }
    // get_normalized_translation_vector, :71-85
    const T diam = 2 * (root_extent * 1 / (T) (1 << (level + 1)));
    const int dim_bound = 2 * nway + 1;
    const int base = 4 * nway + 3;
    int result = 0, mult = 1;
    bool bad = false;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const T tc = box_centers[aligned * d + target_box_id];
        const T sc = box_centers[aligned * d + source_box_id];
        const int v = (int) rint((tc - sc) / diam);
        bad |= !(-dim_bound <= v && v <= dim_bound);           // :108-114
        result += (2 * nway + 1 + v) * mult;                    // :116-122
        mult *= base;
    }
    if (bad) { atomicOr(error_flag, 1); return; }
    if (per_level) result += level * mult;                      // :177-180 (mult == base^D)
    classes[i] = result;
    if (!class_is_used[result]) atomicOr(&class_is_used[result], 1);
}

unsigned blocks_for(int64_t threads) { return (unsigned) std::max<int64_t>(1, div_up(threads, 256)); }

}  // namespace

extern "C" {

int bt_fmm_box_particle_sums(bt_context *ctx, int64_t n, const int32_t *boxes,
                             const int32_t *box_starts, const int32_t *box_counts,
                             const double *values, double *out, int accumulate)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || !box_starts || !box_counts || !out || (n > 0 && !values)) {
        set_error("bt_fmm_box_particle_sums: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n > 0)
        box_particle_sum_kernel<<<blocks_for(n * WAVE), 256, 0, ctx->stream>>>(
            n, boxes, box_starts, box_counts, values, out, accumulate);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_fmm_csr_sum(bt_context *ctx, int64_t nrows, const int32_t *starts, const int32_t *lists,
                   const double *box_values, const int32_t *row_boxes, double *out,
                   int scatter_add)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nrows < 0 || !starts || !box_values || !out || (scatter_add && !row_boxes)) {
        set_error("bt_fmm_csr_sum: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (nrows > 0) {
        if (!scatter_add) {
            csr_sum_kernel<<<blocks_for(nrows * 8), 256, 0, ctx->stream>>>(nrows, starts, lists,
                                                                          box_values, out);
        } else {
            Buf<double> rows;
            BT_CHECK(rows.alloc(ctx->pool, nrows));
            csr_sum_kernel<<<blocks_for(nrows * 8), 256, 0, ctx->stream>>>(nrows, starts, lists,
                                                                          box_values, rows.get());
            scatter_add_rows_kernel<<<blocks_for(nrows), 256, 0, ctx->stream>>>(nrows, row_boxes,
                                                                               rows.get(), out);
            BT_CHECK(bt::sync_stream(ctx));
        }
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_fmm_box_to_particles(bt_context *ctx, int64_t nrows, const int32_t *row_boxes,
                            const int32_t *box_starts, const int32_t *box_counts,
                            const double *row_values, const double *box_values, double *pot,
                            int accumulate)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || nrows < 0 || !box_starts || !box_counts || !pot || (nrows > 0 && !row_boxes)
            || (!row_values && !box_values)) {
        set_error("bt_fmm_box_to_particles: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (nrows > 0)
        box_to_particles_kernel<<<blocks_for(nrows * WAVE), 256, 0, ctx->stream>>>(
            nrows, row_boxes, box_starts, box_counts, row_values, box_values, pot, accumulate);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_fmm_tree_sweep(bt_context *ctx, int64_t n, const int32_t *boxes, const int32_t *child_ids,
                      int64_t aligned_nboxes, int nchildren, const int32_t *parent_ids,
                      double *box_values)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || !box_values || (n > 0 && !boxes) || (!child_ids && !parent_ids)) {
        set_error("bt_fmm_tree_sweep: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n > 0) {
        if (child_ids)
            add_children_kernel<<<blocks_for(n), 256, 0, ctx->stream>>>(n, boxes, child_ids,
                                                                       aligned_nboxes, nchildren,
                                                                       box_values);
        else
            add_parent_kernel<<<blocks_for(n), 256, 0, ctx->stream>>>(n, boxes, parent_ids, box_values);
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_translation_classes(bt_context *ctx, int dims, int coord_kind, int64_t n_entries,
                           const int32_t *from_sep_siblings_lists,
                           const int32_t *from_sep_siblings_starts,
                           const int32_t *target_or_target_parent_boxes, int64_t nttp,
                           const void *box_centers, int64_t aligned_nboxes, double root_extent,
                           const uint8_t *box_levels, int well_sep_is_n_away, int per_level,
                           int nclasses, int32_t *classes, int32_t *class_is_used, int32_t *error)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || dims < 1 || dims > 3 || n_entries < 0 || nttp < 0 || nclasses < 1
            || !from_sep_siblings_starts || !box_centers || !box_levels || !class_is_used || !error
            || (n_entries > 0 && (!from_sep_siblings_lists || !target_or_target_parent_boxes
                                  || !classes))) {
        set_error("bt_translation_classes: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    Buf<int32_t> d_err;
    BT_CHECK(d_err.alloc(ctx->pool, 1));
    BT_HIP_CHECK(hipMemsetAsync(d_err.get(), 0, 4, ctx->stream));
    BT_HIP_CHECK(hipMemsetAsync(class_is_used, 0, (size_t) nclasses * 4, ctx->stream));
    if (n_entries > 0) {
        const unsigned blocks = blocks_for(n_entries);
#define TC_LAUNCH(T, D)                                                                      \
        translation_class_kernel<T, D><<<blocks, 256, 0, ctx->stream>>>(                      \
            n_entries, from_sep_siblings_lists, from_sep_siblings_starts,                    \
            target_or_target_parent_boxes, nttp, (const T *) box_centers, aligned_nboxes,    \
            (T) root_extent, box_levels, well_sep_is_n_away, per_level, classes,             \
            class_is_used, d_err.get())
        if (coord_kind == BT_F64) {
            if (dims == 1) TC_LAUNCH(double, 1); else if (dims == 2) TC_LAUNCH(double, 2);
            else TC_LAUNCH(double, 3);
        } else {
            if (dims == 1) TC_LAUNCH(float, 1); else if (dims == 2) TC_LAUNCH(float, 2);
            else TC_LAUNCH(float, 3);
        }
#undef TC_LAUNCH
        BT_HIP_CHECK(hipGetLastError());
    }
    BT_CHECK(bt::d2h(ctx, error, d_err.get(), 4));
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"
