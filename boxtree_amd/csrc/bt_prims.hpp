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
// Scans of more than this many 4096-element tiles use 16384-element tiles (1024 threads): the
// look-back chain crosses XCDs, a tile of either size costs it about the same, so larger tiles
// pay as soon as they still fill the CUs (LAB_NOTES.md section 8)
inline int64_t scan_big_threshold() { return 256; }

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

// ---------------------------------------------------------------------------
// single-pass variant (chained scan with decoupled look-back): ONE launch, the
// input is read once.  Tiles take tickets from a counter that is never reset (the
// host passes the value it has at launch: stream order makes that exact), so a tile
// only ever waits for tiles that are resident or finished.  A tile publishes one
// relaxed agent-scope 64-bit word {generation:22 | flag:2 | value:40} -- the data is
// the flag -- first its aggregate, then its inclusive prefix; the generation tag
// makes stale words of earlier scans in the same (persistent) array invisible, so
// nothing is cleared between calls.  Values saturate at 2^40-1: every caller that
// scans into 32-bit outputs treats totals >= 2^31 as an error anyway.
// ---------------------------------------------------------------------------

constexpr uint64_t SP_VAL_MASK = (1ull << 40) - 1;
constexpr uint32_t SP_AGG = 1, SP_PREFIX = 2;
constexpr uint32_t SP_GEN_MASK = (1u << 22) - 1;
constexpr uint32_t SP_SPIN_LIMIT = 1u << 24;

__device__ __forceinline__ uint64_t sp_pack(uint32_t gen, uint32_t flag, uint64_t v)
{
    if (v > SP_VAL_MASK) v = SP_VAL_MASK;
    return ((uint64_t) gen << 42) | ((uint64_t) flag << 40) | v;
}

__device__ __forceinline__ uint64_t sp_add_sat(uint64_t a, uint64_t b)
{
    const uint64_t r = a + b;
    return r > SP_VAL_MASK ? SP_VAL_MASK : r;
}

// THREADS = 256 for small inputs; 1024 (tiles of 16384 elements) for large ones: the
// look-back chain advances by about 64 tiles per poll round trip (~1.5 us under load),
// i.e. some 40 tiles/us -- 4096-element tiles cap a 32-bit scan at ~1.3 TB/s (measured:
// 1.6*10^8 elements in 2.2 ms), four times larger tiles leave the chain idle.
// one tile of a scan whose descriptors start at desc (tile = ticket within the scan)
template <class AccT, class OutT, class F, int THREADS>
__device__ __forceinline__ void scan_tile(const F &f, int64_t n, OutT *out, AccT *d_total,
        bool write_total_at_n, uint64_t *desc, uint32_t tile, uint32_t gen, DeviceStatus *status)
{
    constexpr int NW = THREADS / 64;
    constexpr int64_t TILE = (int64_t) THREADS * SCAN_ITEMS;
    using W = uint32_t;
    __shared__ W s_wave[NW];
    __shared__ uint64_t s_wave_tot[NW];
    __shared__ uint64_t s_excl;
    const int w = threadIdx.x >> 6, lane = lane_id();
    const int64_t wave_base = (int64_t) tile * TILE + (int64_t) w * (64 * SCAN_ITEMS);
    W v[SCAN_ITEMS], ex[SCAN_ITEMS];
    uint64_t mine = 0;                 // exact (64-bit) sum of this lane's items
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        const AccT x = (i < n) ? (AccT) f(i) : (AccT) 0;
        v[k] = (W) x;
        mine += (uint64_t) x;
    }
    W carry = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const W incl = wave_inclusive_scan(v[k]);
        ex[k] = carry + incl - v[k];
        carry += __shfl(incl, 63, 64);
    }
    mine = wave_reduce_sum(mine);
    if (lane == 0) { s_wave[w] = carry; s_wave_tot[w] = mine; }
    __syncthreads();
    if (w == 0) {
        uint64_t agg = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) agg += s_wave_tot[i];
        uint64_t excl = 0;
        if (tile == 0) {
            if (lane == 0)
                __hip_atomic_store(desc, sp_pack(gen, SP_PREFIX, agg), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0)
                __hip_atomic_store(desc + tile, sp_pack(gen, SP_AGG, agg), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            int64_t look = (int64_t) tile - 1;
            uint32_t spins = 0;
            while (true) {
                const int64_t idx = look - lane;
                uint64_t word = sp_pack(gen, SP_PREFIX, 0);     // before tile 0: prefix 0
                if (idx >= 0)
                    word = __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t) (word >> 40) & 3u;
                const bool valid = (uint32_t) (word >> 42) == gen && flag != 0;
                const uint64_t is_prefix = __ballot(valid && flag == SP_PREFIX);
                const uint64_t invalid = __ballot(!valid);
                const int first_prefix = is_prefix ? __builtin_ctzll(is_prefix) : 64;
                const int first_invalid = invalid ? __builtin_ctzll(invalid) : 64;
                if (first_invalid <= first_prefix && first_invalid < 64) {
                    // a descriptor this window needs is not there yet
                    if (++spins > SP_SPIN_LIMIT) {
                        if (lane == 0) atomicExch(&status->internal, 77);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                uint64_t val = (lane <= first_prefix) ? (word & SP_VAL_MASK) : 0ull;
                val = wave_reduce_sum(val);
                excl = sp_add_sat(excl, val);
                if (first_prefix < 64) break;
                look -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(desc + tile, sp_pack(gen, SP_PREFIX, sp_add_sat(excl, agg)),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
        const int64_t ntiles = (n + TILE - 1) / TILE;
        if (lane == 0 && (int64_t) tile == ntiles - 1) {
            const uint64_t tot = sp_add_sat(excl, agg);
            if (d_total) *d_total = (AccT) tot;
            if (write_total_at_n) out[n] = (OutT) tot;
        }
    }
    __syncthreads();
    uint64_t run = s_excl;
#pragma unroll
    for (int i = 0; i < NW; ++i)
        if (i < w) run += (uint64_t) s_wave[i];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = wave_base + k * 64 + lane;
        if (i < n) out[i] = (OutT) (run + (uint64_t) ex[k]);
    }
}

template <class AccT, class OutT, class F, int THREADS>
__global__ __launch_bounds__(THREADS) void scan_single_pass_kernel(F f, int64_t n, OutT *out,
        AccT *d_total, bool write_total_at_n, uint64_t *desc, uint32_t *ticket_counter,
        uint32_t ticket_base, uint32_t gen, DeviceStatus *status)
{
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket_counter, 1u) - ticket_base;
    __syncthreads();
    scan_tile<AccT, OutT, F, THREADS>(f, n, out, d_total, write_total_at_n, desc, s_tile, gen, status);
}

// Up to K scans of the same kind in one launch: the tiles of scan k are the tickets
// [first_tile[k], first_tile[k + 1]), and its look-back stops at its own first descriptor.
// Tickets are handed out in order, so a tile only ever waits for tiles that already run.
template <class AccT, class OutT, class F, int K>
struct ScanBatch {
    F f[K];
    int64_t n[K];
    OutT *out[K];
    AccT *total[K];
    uint32_t first_tile[K + 1];
    int count;
};

template <class AccT, class OutT, class F, int THREADS, int K>
__global__ __launch_bounds__(THREADS) void scan_batch_kernel(ScanBatch<AccT, OutT, F, K> b,
        bool write_total_at_n, uint64_t *desc, uint32_t *ticket_counter, uint32_t ticket_base,
        uint32_t gen, DeviceStatus *status)
{
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket_counter, 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    int k = 0;
#pragma unroll
    for (int i = 1; i < K; ++i)
        if (i < b.count && tile >= b.first_tile[i]) k = i;
    scan_tile<AccT, OutT, F, THREADS>(b.f[k], b.n[k], b.out[k], b.total[k], write_total_at_n,
                                      desc + b.first_tile[k], tile - b.first_tile[k], gen, status);
}

int scan_prepare(bt_context *ctx, int64_t ntiles, uint32_t *gen, uint32_t *ticket_base);   // bt_core.hip

// out must have n (+1 if write_total_at_n) elements.  d_total may be null.
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
    if constexpr (std::is_integral<AccT>::value && sizeof(OutT) == 4) {
        uint32_t gen = 0, base = 0;
        if (ntiles > scan_big_threshold()) {
            const int64_t nbig = div_up(n, (int64_t) 1024 * SCAN_ITEMS);
            BT_CHECK(scan_prepare(ctx, nbig, &gen, &base));
            scan_single_pass_kernel<AccT, OutT, F, 1024><<<(unsigned) nbig, 1024, 0, ctx->stream>>>(
                f, n, out, d_total, write_total_at_n, ctx->scan_desc, ctx->scan_ticket, base, gen,
                ctx->d_status);
        } else {
            BT_CHECK(scan_prepare(ctx, ntiles, &gen, &base));
            scan_single_pass_kernel<AccT, OutT, F, SCAN_THREADS>
                <<<(unsigned) ntiles, SCAN_THREADS, 0, ctx->stream>>>(
                    f, n, out, d_total, write_total_at_n, ctx->scan_desc, ctx->scan_ticket, base,
                    gen, ctx->d_status);
        }
        BT_HIP_CHECK(hipGetLastError());
        return BT_OK;
    } else {
        // wide outputs (weight prefix sums, floating-point costs): reduce, scan the
        // tile sums, rescan
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
}

// count <= K scans into 32-bit outputs in one launch (see scan_batch_kernel); same contract
// per scan as device_exclusive_scan.
template <class AccT, class OutT, class F, int K>
int device_exclusive_scan_batch(bt_context *ctx, int count, const F *f, const int64_t *n,
                                OutT *const *out, AccT *const *d_total, bool write_total_at_n)
{
    static_assert(std::is_integral<AccT>::value && sizeof(OutT) == 4, "32-bit outputs only");
    ScanBatch<AccT, OutT, F, K> b{};
    bool big = false;
    for (int k = 0; k < count; ++k) {
        if (n[k] <= 0) {
            if (d_total && d_total[k])
                BT_HIP_CHECK(hipMemsetAsync(d_total[k], 0, sizeof(AccT), ctx->stream));
            if (write_total_at_n) BT_HIP_CHECK(hipMemsetAsync(out[k], 0, sizeof(OutT), ctx->stream));
            continue;
        }
        const int c = b.count++;
        b.f[c] = f[k]; b.n[c] = n[k]; b.out[c] = out[k];
        b.total[c] = d_total ? d_total[k] : nullptr;
        big = big || div_up(n[k], SCAN_TILE) > scan_big_threshold();
    }
    if (b.count == 0) return BT_OK;
    const int64_t tile = big ? (int64_t) 1024 * SCAN_ITEMS : SCAN_TILE;
    int64_t ntiles = 0;
    for (int c = 0; c < b.count; ++c) {
        b.first_tile[c] = (uint32_t) ntiles;
        ntiles += div_up(b.n[c], tile);
    }
    b.first_tile[b.count] = (uint32_t) ntiles;
    uint32_t gen = 0, base = 0;
    BT_CHECK(scan_prepare(ctx, ntiles, &gen, &base));
    if (big)
        scan_batch_kernel<AccT, OutT, F, 1024, K><<<(unsigned) ntiles, 1024, 0, ctx->stream>>>(
            b, write_total_at_n, ctx->scan_desc, ctx->scan_ticket, base, gen, ctx->d_status);
    else
        scan_batch_kernel<AccT, OutT, F, SCAN_THREADS, K><<<(unsigned) ntiles, SCAN_THREADS, 0, ctx->stream>>>(
            b, write_total_at_n, ctx->scan_desc, ctx->scan_ticket, base, gen, ctx->d_status);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

}  // namespace bt
