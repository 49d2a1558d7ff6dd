// Kernels for the multi-GPU exchange step (boxtree_amd/distributed/__init__.py): top-level
// Morton cell of every particle + cell histogram, stable bucketing by owner rank
// (one digit pass of the radix sort), and the gather that fills the send buffers.
// The collectives themselves (RCCL all-reduce / all-to-all over xGMI) are issued
// from Python through torch.distributed.
#include "bt_common.hpp"
#include "bt_prims.hpp"
#include "bt_sort.hpp"

#include <algorithm>

using namespace bt;

namespace {

template <class T, int D>
struct CellArgs {
    const T *x[D];
    T bmin[D], bmax[D];
    int64_t n;
    int level;
};

// same float expression as the key kernel (tbk:374-376) at level `level`
template <class T, int D>
__global__ __launch_bounds__(256) void morton_cells_kernel(CellArgs<T, D> a, uint32_t *cells,
                                                           int32_t *hist)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    uint32_t cell = 0;
    const uint32_t top = (1u << a.level) - 1u;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const T gmin = a.bmin[ax];
        const T gext = a.bmax[ax] - gmin;
        uint32_t v = (uint32_t) (((a.x[ax][i] - gmin) / gext) * (T) (1u << a.level));
        v = v > top ? top : v;
        for (int b = 0; b < a.level; ++b)
            cell |= ((v >> b) & 1u) << (D * b + (D - 1 - ax));       // x most significant
    }
    cells[i] = cell;
    atomicAdd(&hist[cell], 1);
}

// same, with the histogram privatised in LDS (up to 2^15 cells = 128 KiB, one
// workgroup per CU): 1e8 global atomics on 32 K addresses cost ~5 ms otherwise
constexpr int CELLS_LDS_MAX = 1 << 15;

template <class T, int D>
__global__ __launch_bounds__(1024) void morton_cells_lds_kernel(CellArgs<T, D> a, uint32_t *cells,
                                                                int32_t *hist, int ncells)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    for (int c = threadIdx.x; c < ncells; c += 1024) s_hist[c] = 0;
    __syncthreads();
    const uint32_t top = (1u << a.level) - 1u;
    const int64_t stride = (int64_t) gridDim.x * 1024;
    for (int64_t i = (int64_t) blockIdx.x * 1024 + threadIdx.x; i < a.n; i += stride) {
        uint32_t cell = 0;
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            const T gmin = a.bmin[ax];
            const T gext = a.bmax[ax] - gmin;
            uint32_t v = (uint32_t) (((a.x[ax][i] - gmin) / gext) * (T) (1u << a.level));
            v = v > top ? top : v;
            for (int b = 0; b < a.level; ++b)
                cell |= ((v >> b) & 1u) << (D * b + (D - 1 - ax));
        }
        cells[i] = cell;
        atomicAdd(&s_hist[cell], 1u);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncells; c += 1024) {
        const uint32_t v = s_hist[c];
        if (v) atomicAdd(&hist[c], (int32_t) v);
    }
}

__global__ __launch_bounds__(256) void owner_keys_kernel(int64_t n, const uint32_t *cells,
        const int32_t *owner_of_cell, uint32_t *keys)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = (uint32_t) owner_of_cell[cells[i]];
}

template <class U>
__global__ __launch_bounds__(256) void gather_perm_kernel(int64_t n, const uint32_t *perm,
        const U *__restrict__ in, U *__restrict__ out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

// out[i*D + ax] = in[ax][perm[i]]: the D coordinates of a particle travel together
template <class U, int D>
struct PackArgs { const U *in[D]; };

template <class U, int D>
__global__ __launch_bounds__(256) void gather_pack_kernel(int64_t n, const uint32_t *perm,
        PackArgs<U, D> a, U *__restrict__ out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = perm[i];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) out[i * D + ax] = a.in[ax][j];
}

template <class U, int D>
struct UnpackArgs { U *out[D]; };

template <class U, int D>
__global__ __launch_bounds__(256) void unpack_kernel(int64_t n, const U *__restrict__ in,
                                                     UnpackArgs<U, D> a)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) a.out[ax][i] = in[i * D + ax];
}

template <class U, int D>
int pack_impl(bt_context *ctx, const void *const *in, const uint32_t *perm, int64_t n, void *out)
{
    PackArgs<U, D> a;
    for (int ax = 0; ax < D; ++ax) a.in[ax] = (const U *) in[ax];
    gather_pack_kernel<U, D><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(n, perm, a, (U *) out);
    return BT_OK;
}

template <class U, int D>
int unpack_impl(bt_context *ctx, const void *in, int64_t n, void *const *out)
{
    UnpackArgs<U, D> a;
    for (int ax = 0; ax < D; ++ax) a.out[ax] = (U *) out[ax];
    unpack_kernel<U, D><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(n, (const U *) in, a);
    return BT_OK;
}

template <class T, int D>
int cells_impl(bt_context *ctx, const void *const *coords, int64_t n, const double *bmin,
               const double *bmax, int level, uint32_t *cells, int32_t *hist)
{
    CellArgs<T, D> a;
    for (int ax = 0; ax < D; ++ax) {
        a.x[ax] = (const T *) coords[ax];
        a.bmin[ax] = (T) bmin[ax];
        a.bmax[ax] = (T) bmax[ax];
    }
    a.n = n;
    a.level = level;
    const int ncells = 1 << (D * level);
    if (n > 0 && ncells <= CELLS_LDS_MAX) {
        const unsigned blocks = (unsigned) std::min<int64_t>(div_up(n, 1024), ctx->num_cus);
        morton_cells_lds_kernel<T, D><<<blocks, 1024, (size_t) ncells * 4, ctx->stream>>>(
            a, cells, hist, ncells);
    } else if (n > 0) {
        morton_cells_kernel<T, D><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(a, cells, hist);
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

// ---- stable partition of the particles by owner rank, payload carried along -------------
//
// The send buffer of the exchange, made in one sweep over the coordinates: a wave owns 1024
// consecutive particles, counts them per owner (pp_count_kernel), a scan over (owner, wave)
// turns the counts into record offsets, and pp_scatter_kernel writes every particle's
// coordinates, interleaved, at its place -- the segment a rank keeps straight into the
// receive buffer.  Reads are sequential, writes go to `nranks` advancing runs.  (A
// permutation by owner followed by a gather reads the coordinate arrays in `nranks`
// interleaved strided passes: with 8 ranks every 64-byte line is fetched 8 times.)
constexpr int PP_MAX_RANKS = BT_MGPU_MAX_RANKS;
constexpr int PP_ITEMS = 16;                    // particles per lane
constexpr int PP_WAVE_ITEMS = 64 * PP_ITEMS;

// lanes of the wave whose value equals this lane's (match-any by ballots over `bits` bits)
__device__ __forceinline__ uint64_t pp_match(uint32_t d, int bits)
{
    uint64_t mask = ~0ull;
    for (int b = 0; b < bits; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        mask &= bit ? bal : ~bal;
    }
    return mask;
}

// counts[owner * nwaves + wave]
__global__ __launch_bounds__(256) void pp_count_kernel(const uint32_t *cells, int64_t n,
        const int32_t *owner_of_cell, int nranks, int bits, int64_t nwaves, int32_t *counts)
{
    __shared__ int32_t s_cnt[4][PP_MAX_RANKS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wave = (int64_t) blockIdx.x * 4 + w;
    for (int r = lane; r < nranks; r += 64) s_cnt[w][r] = 0;
    __builtin_amdgcn_wave_barrier();
    if (wave >= nwaves) return;
    const int64_t base = wave * PP_WAVE_ITEMS + lane;
    for (int j = 0; j < PP_ITEMS; ++j) {
        const int64_t i = base + (int64_t) j * 64;
        const bool in = i < n;
        // (lanes past the end form a group of their own)
        const uint32_t d = in ? (uint32_t) owner_of_cell[cells[i]] : (uint32_t) nranks;
        const uint64_t mask = pp_match(d, bits + 1);
        const bool leader = (mask & ((1ull << lane) - 1ull)) == 0ull;
        if (in && leader) s_cnt[w][d] += (int32_t) __popcll(mask);
        __builtin_amdgcn_wave_barrier();
    }
    for (int r = lane; r < nranks; r += 64) counts[(int64_t) r * nwaves + wave] = s_cnt[w][r];
}

struct PpScan {
    const int32_t *c;
    __device__ int64_t operator()(int64_t i) const { return (int64_t) c[i]; }
};

template <class U, int D>
struct PpArgs {
    const U *in[D];
    const uint32_t *cells;
    const int32_t *owner_of_cell;
    const int32_t *offsets;         // [nranks * nwaves] exclusive scan of the counts
    int64_t n, nwaves;
    int nranks, bits, self_rank;
    int64_t self_delta;             // receive offset of the own segment minus its send offset
    U *send, *recv;
};

template <class U, int D>
__global__ __launch_bounds__(256) void pp_scatter_kernel(PpArgs<U, D> a)
{
    __shared__ int32_t s_run[4][PP_MAX_RANKS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t wave = (int64_t) blockIdx.x * 4 + w;
    if (wave >= a.nwaves) return;
    for (int r = lane; r < a.nranks; r += 64) s_run[w][r] = a.offsets[(int64_t) r * a.nwaves + wave];
    __builtin_amdgcn_wave_barrier();
    const int64_t base = wave * PP_WAVE_ITEMS + lane;
    for (int j = 0; j < PP_ITEMS; ++j) {
        const int64_t i = base + (int64_t) j * 64;
        const bool in = i < a.n;
        const uint32_t d = in ? (uint32_t) a.owner_of_cell[a.cells[i]] : (uint32_t) a.nranks;
        U v[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) v[ax] = in ? a.in[ax][i] : (U) 0;
        const uint64_t mask = pp_match(d, a.bits + 1);
        const uint64_t below = mask & ((1ull << lane) - 1ull);
        int32_t old = 0;
        if (in) old = s_run[w][d];
        __builtin_amdgcn_wave_barrier();
        if (in && below == 0ull) s_run[w][d] = old + (int32_t) __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (in) {
            const int64_t pos = (int64_t) old + __popcll(below);      // record index, owner-major
            U *dst = ((int) d == a.self_rank) ? a.recv + (pos + a.self_delta) * D : a.send + pos * D;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) dst[ax] = v[ax];
        }
    }
}

template <class U, int D>
int partition_pack_impl(bt_context *ctx, const void *const *in, const uint32_t *cells, int64_t n,
                        const int32_t *owner_of_cell, int nranks, int self_rank,
                        int64_t self_send_offset, int64_t self_recv_offset, void *send, void *recv)
{
    const int64_t nwaves = div_up(n, PP_WAVE_ITEMS);
    int bits = 0;
    while ((1 << bits) < nranks) ++bits;
    Buf<int32_t> counts, offsets;
    BT_CHECK(counts.alloc(ctx->pool, (int64_t) nranks * nwaves));
    BT_CHECK(offsets.alloc(ctx->pool, (int64_t) nranks * nwaves + 1));
    pp_count_kernel<<<(unsigned) div_up(nwaves, 4), 256, 0, ctx->stream>>>(
        cells, n, owner_of_cell, nranks, bits, nwaves, counts.get());
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, PpScan{counts.get()}, (int64_t) nranks * nwaves,
                                                      offsets.get(), (int64_t *) nullptr, true)));
    PpArgs<U, D> a{};
    for (int ax = 0; ax < D; ++ax) a.in[ax] = (const U *) in[ax];
    a.cells = cells; a.owner_of_cell = owner_of_cell; a.offsets = offsets.get();
    a.n = n; a.nwaves = nwaves; a.nranks = nranks; a.bits = bits; a.self_rank = self_rank;
    a.self_delta = self_recv_offset - self_send_offset;
    a.send = (U *) send; a.recv = (U *) recv;
    pp_scatter_kernel<U, D><<<(unsigned) div_up(nwaves, 4), 256, 0, ctx->stream>>>(a);
    BT_HIP_CHECK(hipGetLastError());
    return bt::finish_call(ctx);
}

}  // namespace

extern "C" {

int bt_morton_cells(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                    int64_t n, const double *bbox_min, const double *bbox_max, int level,
                    uint32_t *cells_out, int32_t *hist_inout)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !coords || !bbox_min || !bbox_max || n < 0 || dims < 1 || dims > BT_MAX_DIMS
            || level < 1 || dims * level > 24 || (n > 0 && (!cells_out || !hist_inout))) {
        set_error("bt_morton_cells: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    const bool f64 = coord_kind == BT_F64;
    switch (dims) {
    case 1: return f64 ? cells_impl<double, 1>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout)
                       : cells_impl<float, 1>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout);
    case 2: return f64 ? cells_impl<double, 2>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout)
                       : cells_impl<float, 2>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout);
    default: return f64 ? cells_impl<double, 3>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout)
                        : cells_impl<float, 3>(ctx, coords, n, bbox_min, bbox_max, level, cells_out, hist_inout);
    }
}

int bt_bucket_permutation(bt_context *ctx, const uint32_t *cells, int64_t n,
                          const int32_t *owner_of_cell, int nranks, uint32_t *perm_out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || nranks < 1 || nranks > 256 || (n > 0 && (!cells || !owner_of_cell || !perm_out))) {
        set_error("bt_bucket_permutation: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return BT_OK;
    BT_CHECK(reset_status(ctx));
    Buf<uint32_t> ka, kb, vb;
    BT_CHECK(ka.alloc(ctx->pool, n));
    BT_CHECK(kb.alloc(ctx->pool, n));
    BT_CHECK(vb.alloc(ctx->pool, n));
    owner_keys_kernel<<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(n, cells, owner_of_cell, ka.get());
    int bits = 1;
    while ((1 << bits) < nranks) ++bits;
    bool in_b = false;
    // one stable digit pass over (owner, 0..n-1): perm = original indices grouped by owner
    BT_CHECK(radix_sort_pairs<uint32_t>(ctx, ka.get(), vb.get(), kb.get(), perm_out, n, 0, bits,
                                        true, &in_b));
    if (!in_b)
        BT_HIP_CHECK(hipMemcpyAsync(perm_out, vb.get(), (size_t) n * 4, hipMemcpyDeviceToDevice,
                                    ctx->stream));
    return check_status(ctx);
}

int bt_partition_pack(bt_context *ctx, int dims, int elem_size, const void *const *in,
                      const uint32_t *cells, int64_t n, const int32_t *owner_of_cell, int nranks,
                      int self_rank, int64_t self_send_offset, int64_t self_recv_offset,
                      void *send, void *recv)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || dims < 1 || dims > BT_MAX_DIMS || (elem_size != 4 && elem_size != 8)
            || nranks < 1 || nranks > PP_MAX_RANKS || self_rank < 0 || self_rank >= nranks
            || (n > 0 && (!in || !cells || !owner_of_cell))) {
        // (send / recv may be NULL for a rank that keeps nothing / sends nothing)
        set_error("bt_partition_pack: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return BT_OK;
    if (n >= ((int64_t) 1 << 31)) {
        set_error("bt_partition_pack: n=%lld exceeds 2^31-1", (long long) n);
        return BT_ERR_INVALID;
    }
#define BT_PP(U, D) partition_pack_impl<U, D>(ctx, in, cells, n, owner_of_cell, nranks, self_rank, \
                                              self_send_offset, self_recv_offset, send, recv)
    if (elem_size == 8) return dims == 1 ? BT_PP(uint64_t, 1) : dims == 2 ? BT_PP(uint64_t, 2) : BT_PP(uint64_t, 3);
    return dims == 1 ? BT_PP(uint32_t, 1) : dims == 2 ? BT_PP(uint32_t, 2) : BT_PP(uint32_t, 3);
#undef BT_PP
}

int bt_gather(bt_context *ctx, int elem_size, const void *in, const uint32_t *perm, int64_t n,
              void *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || (elem_size != 4 && elem_size != 8) || (n > 0 && (!in || !perm || !out))) {
        set_error("bt_gather: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return BT_OK;
    const unsigned blocks = (unsigned) div_up(n, 256);
    if (elem_size == 8)
        gather_perm_kernel<uint64_t><<<blocks, 256, 0, ctx->stream>>>(n, perm, (const uint64_t *) in, (uint64_t *) out);
    else
        gather_perm_kernel<uint32_t><<<blocks, 256, 0, ctx->stream>>>(n, perm, (const uint32_t *) in, (uint32_t *) out);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}


int bt_gather_pack(bt_context *ctx, int dims, int elem_size, const void *const *in,
                   const uint32_t *perm, int64_t n, void *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || dims < 1 || dims > BT_MAX_DIMS || (elem_size != 4 && elem_size != 8)
            || (n > 0 && (!in || !perm || !out))) {
        set_error("bt_gather_pack: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return BT_OK;
    int s = BT_OK;
    if (elem_size == 8) {
        s = dims == 1 ? pack_impl<uint64_t, 1>(ctx, in, perm, n, out)
          : dims == 2 ? pack_impl<uint64_t, 2>(ctx, in, perm, n, out)
                      : pack_impl<uint64_t, 3>(ctx, in, perm, n, out);
    } else {
        s = dims == 1 ? pack_impl<uint32_t, 1>(ctx, in, perm, n, out)
          : dims == 2 ? pack_impl<uint32_t, 2>(ctx, in, perm, n, out)
                      : pack_impl<uint32_t, 3>(ctx, in, perm, n, out);
    }
    BT_CHECK(s);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_unpack(bt_context *ctx, int dims, int elem_size, const void *in, int64_t n, void *const *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || dims < 1 || dims > BT_MAX_DIMS || (elem_size != 4 && elem_size != 8)
            || (n > 0 && (!in || !out))) {
        set_error("bt_unpack: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return BT_OK;
    int s = BT_OK;
    if (elem_size == 8) {
        s = dims == 1 ? unpack_impl<uint64_t, 1>(ctx, in, n, out)
          : dims == 2 ? unpack_impl<uint64_t, 2>(ctx, in, n, out)
                      : unpack_impl<uint64_t, 3>(ctx, in, n, out);
    } else {
        s = dims == 1 ? unpack_impl<uint32_t, 1>(ctx, in, n, out)
          : dims == 2 ? unpack_impl<uint32_t, 2>(ctx, in, n, out)
                      : unpack_impl<uint32_t, 3>(ctx, in, n, out);
    }
    BT_CHECK(s);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Local essential tree (LET) of a sharded traversal: the shared top levels, the
// rank's own subtrees and the subtrees of the neighbouring cells of other ranks,
// assembled from Morton paths (boxtree_amd/distributed/__init__.py step 6).
// ---------------------------------------------------------------------------

namespace {

// Morton path of a box from its centre: centres sit at (i + 1/2) / 2^level of the
// root box, so floor((c - min) / extent * 2^level) recovers i exactly.
template <class T, int D>
__global__ __launch_bounds__(256) void box_paths_kernel(int64_t nboxes, int64_t aligned,
        const T *centers, const uint8_t *levels, double bmin0, double bmin1, double bmin2,
        double root_extent, uint64_t *paths)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    const int level = levels[b];
    const double bmin[3] = {bmin0, bmin1, bmin2};
    uint64_t path = 0;
    const double scale = (double) (1ull << level);
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const double t = ((double) centers[(int64_t) ax * aligned + b] - bmin[ax]) / root_extent * scale;
        int64_t v = (int64_t) floor(t);
        v = v < 0 ? 0 : (v >= (int64_t) scale ? (int64_t) scale - 1 : v);
        for (int bit = 0; bit < level; ++bit)
            path |= (uint64_t) ((v >> bit) & 1) << (D * bit + (D - 1 - ax));
    }
    paths[b] = path;
}

// position of `key` in the ascending array a[lo, hi), or -1
__device__ __forceinline__ int64_t find_path(const uint64_t *a, int64_t lo, int64_t hi, uint64_t key)
{
    int64_t l = lo, h = hi;
    while (l < h) {
        const int64_t mid = (l + h) >> 1;
        if (a[mid] < key) l = mid + 1; else h = mid;
    }
    return (l < hi && a[l] == key) ? l : -1;
}

// one thread per box: the parent by path lookup in the level above; the box then
// enters itself into its parent's child row (rows are cleared beforehand, a slot
// nobody claims stays 0)
template <int D>
__global__ __launch_bounds__(256) void let_link_kernel(int32_t b0, int32_t b1, int32_t prev0,
        int64_t aligned, const uint64_t *paths, int32_t *parent_ids, int32_t *child_ids,
        int32_t *missing_parent)
{
    constexpr int C = 1 << D;
    const int32_t b = b0 + (int32_t) (blockIdx.x * 256 + threadIdx.x);
    if (b >= b1) return;
    if (b == 0) { parent_ids[0] = 0; return; }
    const uint64_t p = paths[b];
    const int64_t par = find_path(paths, prev0, b0, p >> D);
    parent_ids[b] = par < 0 ? 0 : (int32_t) par;
    if (par < 0) { atomicExch(missing_parent, 1); return; }
    child_ids[(int64_t) (p & (uint64_t) (C - 1)) * aligned + par] = b;
}

// child centre = parent centre +/- root_extent / 2^(1+level): the builder's own
// chain of roundings (tree_build_kernels.py:698-705), level by level
template <class T, int D>
__global__ __launch_bounds__(256) void let_centers_kernel(int32_t b0, int32_t b1, int level,
        int64_t aligned, const uint64_t *paths, const int32_t *parent_ids, T root_extent,
        T *centers)
{
    const int32_t b = b0 + blockIdx.x * 256 + threadIdx.x;
    if (b >= b1) return;
    const int32_t par = parent_ids[b];
    const int m = (int) (paths[b] & ((1u << D) - 1));
    const T radius = (root_extent * 1 / (T) (1ull << (1 + level)));
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const bool has_bit = (m >> (D - 1 - ax)) & 1;
        const T pc = centers[(int64_t) ax * aligned + par];
        centers[(int64_t) ax * aligned + b] = has_bit ? pc + radius : pc - radius;
    }
}

template <class T, int D>
int let_build_impl(bt_context *ctx, int nlevels, const int32_t *level_starts, const uint64_t *paths,
                   int64_t aligned, const double *bbox_min, const double *bbox_max,
                   double root_extent, int32_t *parent_ids, int32_t *child_ids, T *centers)
{
    constexpr int C = 1 << D;
    const int32_t nboxes = level_starts[nlevels];
    Buf<int32_t> d_missing;
    BT_CHECK(d_missing.alloc(ctx->pool, 1));
    BT_HIP_CHECK(hipMemsetAsync(d_missing.get(), 0, 4, ctx->stream));
    BT_HIP_CHECK(hipMemsetAsync(child_ids, 0, (size_t) C * (size_t) aligned * 4, ctx->stream));
    for (int lev = 0; lev < nlevels; ++lev) {
        const int32_t b0 = level_starts[lev], b1 = level_starts[lev + 1];
        if (b1 <= b0) continue;
        const int32_t prev0 = lev > 0 ? level_starts[lev - 1] : 0;
        let_link_kernel<D><<<(unsigned) div_up(b1 - b0, 256), 256, 0, ctx->stream>>>(
            b0, b1, prev0, aligned, paths, parent_ids, child_ids, d_missing.get());
    }
    // root centre: tree_build.py:585-590
    T root[D];
    for (int ax = 0; ax < D; ++ax) {
        const T mn = (T) bbox_min[ax], mx = (T) bbox_max[ax];
        root[ax] = mn + (mx - mn) / 2;
        BT_HIP_CHECK(hipMemcpyAsync(centers + (int64_t) ax * aligned, &root[ax], sizeof(T),
                                    hipMemcpyHostToDevice, ctx->stream));
    }
    BT_CHECK(bt::sync_stream(ctx));        // `root` goes out of scope
    for (int lev = 1; lev < nlevels; ++lev) {
        const int32_t b0 = level_starts[lev], b1 = level_starts[lev + 1];
        if (b1 <= b0) continue;
        let_centers_kernel<T, D><<<(unsigned) div_up(b1 - b0, 256), 256, 0, ctx->stream>>>(
            b0, b1, lev, aligned, paths, parent_ids, (T) root_extent, centers);
    }
    BT_HIP_CHECK(hipGetLastError());
    int32_t missing = 0;
    BT_CHECK(bt::d2h(ctx, &missing, d_missing.get(), 4));
    BT_CHECK(bt::sync_stream(ctx));
    if (missing) {
        set_error("bt_let_build: a box has no parent among the boxes of the level above "
                  "(%d boxes)", nboxes);
        return BT_ERR_INVALID;
    }
    return BT_OK;
}

}  // namespace

extern "C" {

int bt_box_morton_paths(bt_context *ctx, int dims, int coord_kind, int64_t nboxes,
                        int64_t aligned_nboxes, const void *box_centers, const uint8_t *box_levels,
                        const double *bbox_min, double root_extent, uint64_t *paths)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || dims < 1 || dims > 3 || nboxes < 0 || !box_centers || !box_levels || !bbox_min
            || !paths || (coord_kind != BT_F32 && coord_kind != BT_F64)) {
        set_error("bt_box_morton_paths: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (nboxes == 0) return BT_OK;
    const unsigned blocks = (unsigned) div_up(nboxes, 256);
    const double m0 = bbox_min[0], m1 = dims > 1 ? bbox_min[1] : 0, m2 = dims > 2 ? bbox_min[2] : 0;
#define BP(T, D) box_paths_kernel<T, D><<<blocks, 256, 0, ctx->stream>>>(                    \
        nboxes, aligned_nboxes, (const T *) box_centers, box_levels, m0, m1, m2, root_extent, paths)
    if (coord_kind == BT_F64) { if (dims == 1) BP(double, 1); else if (dims == 2) BP(double, 2); else BP(double, 3); }
    else { if (dims == 1) BP(float, 1); else if (dims == 2) BP(float, 2); else BP(float, 3); }
#undef BP
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

int bt_let_build(bt_context *ctx, int dims, int coord_kind, int nlevels,
                 const int32_t *level_start_box_nrs, const uint64_t *paths, int64_t aligned_nboxes,
                 const double *bbox_min, const double *bbox_max, double root_extent,
                 int32_t *box_parent_ids, int32_t *box_child_ids, void *box_centers)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || dims < 1 || dims > 3 || nlevels < 1 || nlevels > BT_MAX_LEVELS
            || !level_start_box_nrs || !paths || !bbox_min || !bbox_max || !box_parent_ids
            || !box_child_ids || !box_centers || (coord_kind != BT_F32 && coord_kind != BT_F64)) {
        set_error("bt_let_build: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
#define LB(T, D) return let_build_impl<T, D>(ctx, nlevels, level_start_box_nrs, paths,            \
        aligned_nboxes, bbox_min, bbox_max, root_extent, box_parent_ids, box_child_ids,           \
        (T *) box_centers)
    if (coord_kind == BT_F64) { if (dims == 1) LB(double, 1); else if (dims == 2) LB(double, 2); else LB(double, 3); }
    else { if (dims == 1) LB(float, 1); else if (dims == 2) LB(float, 2); else LB(float, 3); }
#undef LB
}

}  // extern "C"
