// Device-side building blocks shared by the tree and traversal kernels:
// wave64 scan/reduce helpers and a device-wide exclusive scan
// (reduce -> scan of tile sums -> rescan; 2 reads + 1 write of the input).
#pragma once

#include "bt_common.hpp"

#include <type_traits>

namespace bt {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set bits of `mask` strictly below the calling lane
__device__ __forceinline__ int mask_popc_lt(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

template <class T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = __shfl_up(v, off, 64);
        if (lane_id() >= off) v += o;
    }
    return v;
}

template <class T>
__device__ __forceinline__ T wave_reduce_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide exclusive scan of one value per thread (blockDim.x == THREADS).
// s_tmp must hold THREADS/64 + 1 elements.  Returns exclusive prefix; *total
// (if non-null) receives the block sum in every thread.
template <class T, int THREADS>
__device__ __forceinline__ T block_exclusive_scan(T v, T *s_tmp, T *total)
{
    constexpr int NW = THREADS / 64;
    const int w = threadIdx.x >> 6;
    T incl = wave_inclusive_scan(v);
    if (lane_id() == 63) s_tmp[w] = incl;
    __syncthreads();
    T woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        T s = s_tmp[i];
        if (i < w) woff += s;
        tot += s;
    }
    __syncthreads();
    if (total) *total = tot;
    return woff + incl - v;
}

// ---------------------------------------------------------------------------
// device-wide exclusive scan:  out[i] = sum_{j<i} f(j),  total -> *d_total
// f is a device functor  AccT operator()(int64_t i) const.
// ---------------------------------------------------------------------------

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class AccT, class F>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(F f, int64_t n, AccT *tile_sums)
{
    __shared__ AccT s_tmp[SCAN_THREADS / 64 + 1];
    const int64_t base = (int64_t) blockIdx.x * SCAN_TILE;
    AccT acc = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + (int64_t) k * SCAN_THREADS + threadIdx.x;
        if (i < n) acc += f(i);
    }
    acc = wave_reduce_sum(acc);
    if (lane_id() == 0) s_tmp[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        AccT t = 0;
        for (int i = 0; i < SCAN_THREADS / 64; ++i) t += s_tmp[i];
        tile_sums[blockIdx.x] = t;
    }
}

template <class AccT>
__global__ __launch_bounds__(1024) void scan_tile_sums_kernel(AccT *tile_sums, int64_t ntiles,
                                                             AccT *d_total)
{
    __shared__ AccT s_tmp[1024 / 64 + 1];
    AccT carry = 0;
    for (int64_t base = 0; base < ntiles; base += 1024) {
        int64_t i = base + threadIdx.x;
        AccT v = (i < ntiles) ? tile_sums[i] : (AccT) 0;
        AccT tot;
        AccT ex = block_exclusive_scan<AccT, 1024>(v, s_tmp, &tot);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && d_total) *d_total = carry;
}

template <class AccT, class OutT, bool NARROW = (std::is_integral<OutT>::value
                                                && sizeof(OutT) == 4 && sizeof(AccT) == 8)>
struct ScanIntraTile { using type = AccT; };
template <class AccT, class OutT>
struct ScanIntraTile<AccT, OutT, true> { using type = uint32_t; };

// Elements are assigned to lanes wave-striped (item k of lane l = wave chunk + 64 k + l):
// every load and store instruction of a wave covers one contiguous 256-byte (int32)
// stretch.  (A blocked assignment -- 16 consecutive elements per thread -- makes each
// instruction touch 64 different lines; measured 0.57 TB/s on a 5*10^7-element scan.)
template <class AccT, class OutT, class F>
__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(F f, int64_t n,
        const AccT *tile_sums, OutT *out, bool write_total_at_n)
{
    constexpr int NW = SCAN_THREADS / 64;
    // Inside a tile the sums are carried in the width of the output: with 32-bit
    // outputs and 64-bit accumulation (list lengths checked against the int32 CSR
    // limit) a tile's own partial sums fit 32 bits whenever the result is
    // representable at all, and the wave scans cost half the shuffles.
    using W = typename ScanIntraTile<AccT, OutT>::type;
    __shared__ W s_wave[NW];
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int64_t wave_base = (int64_t) blockIdx.x * SCAN_TILE + (int64_t) w * (64 * SCAN_ITEMS);
    W v[SCAN_ITEMS], ex[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        v[k] = (i < n) ? (W) f(i) : (W) 0;
    }
    W carry = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const W incl = wave_inclusive_scan(v[k]);
        ex[k] = carry + incl - v[k];
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_wave[w] = carry;
    __syncthreads();
    AccT run = tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if (i < w) run += (AccT) s_wave[i];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        if (i < n) out[i] = (OutT) (run + (AccT) ex[k]);
        if (write_total_at_n && i + 1 == n) out[n] = (OutT) (run + (AccT) ex[k] + (AccT) v[k]);
    }
}

// out must have n (+1 if write_total_at_n) elements.  d_total may be null.
// tile_sums scratch: div_up(n, SCAN_TILE) AccT elements.
template <class AccT, class OutT, class F>
int device_exclusive_scan(bt_context *ctx, F f, int64_t n, OutT *out, AccT *d_total,
                          bool write_total_at_n = false)
{
    if (n <= 0) {
        if (d_total) BT_HIP_CHECK(hipMemsetAsync(d_total, 0, sizeof(AccT), ctx->stream));
        if (write_total_at_n) BT_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(OutT), ctx->stream));
        return BT_OK;
    }
    const int64_t ntiles = div_up(n, SCAN_TILE);
    Buf<AccT> sums;
    BT_CHECK(sums.alloc(ctx->pool, ntiles));
    scan_reduce_kernel<AccT, F><<<(unsigned) ntiles, SCAN_THREADS, 0, ctx->stream>>>(f, n, sums.get());
    scan_tile_sums_kernel<AccT><<<1, 1024, 0, ctx->stream>>>(sums.get(), ntiles, d_total);
    scan_final_kernel<AccT, OutT, F><<<(unsigned) ntiles, SCAN_THREADS, 0, ctx->stream>>>(
        f, n, sums.get(), out, write_total_at_n);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
    // note: `sums` returns to the pool here; the pool never hands memory to
    // another stream and all work is stream-ordered, so reuse is safe.
}

}  // namespace bt
