// Onesweep-style LSD radix sort for gfx950 (wave64).
//
// One up-front kernel histograms every digit of every key (keys read once).
// Each digit pass is then ONE kernel: a workgroup takes the next tile (ticket
// from an atomic counter, so all predecessor tiles are resident or done),
// ranks its keys with wave-wide ballot matching (no per-thread counters),
// obtains the global offset of each of its 256 digit bins by decoupled
// look-back over the preceding tiles' published bin counts, reorders the tile
// through LDS so that equal digits are contiguous, and writes coalesced runs.
// Algorithmic traffic per pass: (sizeof(key)+4) bytes read + written per pair.
//
// Cross-workgroup hand-off is a single relaxed agent-scope 64-bit word per
// (tile, digit): {generation|flag, count}.  The data IS the flag (one aligned
// 8-byte sc1 store), so no fence is needed and placement on XCDs is irrelevant.
#include "bt_sort.hpp"
#include "bt_prims.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace bt {

constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;
constexpr uint32_t LB_AGG = 1, LB_PREFIX = 2;
constexpr uint32_t LOOKBACK_SPIN_LIMIT = 1u << 22;

// Tile shape of the pair sort (threads x keys per thread; MINW = waves per SIMD the register
// allocator must leave room for).  Twelve other shapes (256 ... 1024 threads, 8 ... 24 keys) were
// measured in rounds 1-3 (LAB_NOTES.md section 4) and dropped with their switch.
struct Cfg0 { static constexpr int THREADS = 512, ITEMS = 16, MINW = 4; };

static int sort_dbg()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("BT_SORT_DBG"); v = e ? atoi(e) : 0; }
    return v;
}

// Look-back words.  64-bit: {generation | flag, count}, one array shared by all passes
// (the generation tells them apart).  32-bit (n < 2^30): {flag:2 | count:30}, one
// zeroed array per pass -- half the bytes every poll of a predecessor row moves (the
// polls were ~0.25 GB of the 2.89 GB a pass over 10^8 pairs moved).
template <class W> struct Lb;
template <> struct Lb<uint64_t> {
    static __device__ __forceinline__ uint64_t pack(uint32_t gen, uint32_t flag, uint32_t v)
    {
        return ((uint64_t) ((gen << 2) | flag) << 32) | v;
    }
    static __device__ __forceinline__ uint32_t tag(uint64_t w) { return (uint32_t) (w >> 32); }
    static __device__ __forceinline__ uint32_t want(uint32_t gen, uint32_t flag) { return (gen << 2) | flag; }
    static __device__ __forceinline__ uint32_t value(uint64_t w) { return (uint32_t) w; }
};
template <> struct Lb<uint32_t> {
    static __device__ __forceinline__ uint32_t pack(uint32_t, uint32_t flag, uint32_t v)
    {
        return (flag << 30) | v;
    }
    static __device__ __forceinline__ uint32_t tag(uint32_t w) { return w >> 30; }
    static __device__ __forceinline__ uint32_t want(uint32_t, uint32_t flag) { return flag; }
    static __device__ __forceinline__ uint32_t value(uint32_t w) { return w & 0x3fffffffu; }
};

template <class W>
__device__ __forceinline__ void lb_store(W *p, W v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class W>
__device__ __forceinline__ W lb_load(const W *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-wide match of a digit (the "match-any" of the ranking phase): *below = lanes below
// the caller with the same digit, *cnt = lanes with the same digit.  Per digit bit: the
// bit sign-extended over a register (v_bfe_i32), its ballot (v_cmp), and one
// v_bitop3_b32 (gfx950) per half of the lane mask, mask & ~(ballot ^ bit) -- 4 vector
// instructions per bit where select-and-mask code compiles to 9.
template <int BITS>
__device__ __forceinline__ void match_digit(uint32_t d, uint32_t *below, uint32_t *cnt)
{
    uint32_t mlo = ~0u, mhi = ~0u;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const uint32_t e = (uint32_t) __builtin_amdgcn_sbfe((int) d, b, 1);
        const uint64_t bal = __builtin_amdgcn_uicmp(e, 0u, 33 /* ICMP_NE */);
        mlo = __builtin_amdgcn_bitop3_b32(mlo, (uint32_t) bal, e, 0x90);
        mhi = __builtin_amdgcn_bitop3_b32(mhi, (uint32_t) (bal >> 32), e, 0x90);
    }
    *below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
    *cnt = (uint32_t) (__popc(mlo) + __popc(mhi));
}

// ---- up-front histogram of all digits -------------------------------------

template <class KeyT, int MAXP>
__global__ __launch_bounds__(256) void sort_hist_kernel(const KeyT *keys, uint32_t n,
        int begin_bit, int npasses, uint32_t *ghist /* [npasses][RADIX] */,
        KeyT *copy_out /* or null: a copy of the keys */)
{
    __shared__ uint32_t s_h[MAXP * RADIX];
    for (int i = threadIdx.x; i < npasses * RADIX; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        KeyT k = keys[i];
        if (copy_out) copy_out[i] = k;
        k >>= begin_bit;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            if (p < npasses) {
                atomicAdd(&s_h[p * RADIX + (uint32_t) (k & 0xFF)], 1u);
                k >>= RADIX_BITS;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npasses * RADIX; i += 256) {
        uint32_t c = s_h[i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

// exclusive scan of each pass's 256 bins (in place)
__global__ __launch_bounds__(RADIX) void sort_hist_scan_kernel(uint32_t *ghist)
{
    __shared__ uint32_t s_tmp[RADIX / 64 + 1];
    uint32_t *h = ghist + blockIdx.x * RADIX;
    uint32_t v = h[threadIdx.x];
    uint32_t ex = block_exclusive_scan<uint32_t, RADIX>(v, s_tmp, (uint32_t *) nullptr);
    h[threadIdx.x] = ex;
}

// ---- look-back seeding --------------------------------------------------------
// All workgroups of the first residency wave start together; without help tile j
// would walk back over ~j/2 unfinished predecessors (hundreds of dependent
// ~1 us polls, measured 0.25 ms of a 0.68 ms pass at 1e8 keys).  The digit
// counts of the first `nseed` tiles are therefore computed up front (33 MB of
// keys) and published as inclusive prefixes before the pass starts.

template <class KeyT, int TILE>
__global__ __launch_bounds__(256) void seed_hist_kernel(const KeyT *__restrict__ keys, uint32_t n,
        int shift, uint32_t *tile_hist /* [nseed][RADIX] */)
{
    __shared__ uint32_t s_h[RADIX];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t) TILE;
    for (uint32_t i = threadIdx.x; i < (uint32_t) TILE; i += 256) {
        const uint32_t g = base + i;
        if (g < n) atomicAdd(&s_h[(uint32_t) (keys[g] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
    tile_hist[blockIdx.x * RADIX + threadIdx.x] = s_h[threadIdx.x];
}

constexpr int MAX_SEED = 1024;

// one workgroup per digit scans that digit's counts over the seeded tiles
template <class W>
__global__ __launch_bounds__(MAX_SEED) void seed_scan_kernel(const uint32_t *tile_hist,
        uint32_t nseed, W *lookback, uint32_t gen)
{
    __shared__ uint32_t s_tmp[MAX_SEED / 64 + 1];
    const uint32_t d = blockIdx.x, t = threadIdx.x;
    const uint32_t c = (t < nseed) ? tile_hist[t * RADIX + d] : 0u;
    const uint32_t ex = block_exclusive_scan<uint32_t, MAX_SEED>(c, s_tmp, (uint32_t *) nullptr);
    if (t < nseed) lookback[(uint64_t) t * RADIX + d] = Lb<W>::pack(gen, LB_PREFIX, ex + c);
}

// ---- one digit pass ---------------------------------------------------------

template <class KeyT, class W, int THREADS, int ITEMS, int MINW, bool IDENTITY_VALS>
__global__ __launch_bounds__(THREADS, MINW) void onesweep_kernel(
        const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
        KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out,
        uint32_t n, int shift, const uint32_t *__restrict__ digit_start,
        W *lookback, uint32_t *tile_counter, uint32_t gen, DeviceStatus *status, int dbg,
        uint32_t nseed)
{
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVE_ITEMS = 64 * ITEMS;

    __shared__ uint32_t s_hist[NW * RADIX];
    __shared__ uint32_t s_digit_base[RADIX];
    __shared__ uint32_t s_global_base[RADIX];
    __shared__ uint32_t s_tmp[RADIX / 64 + 1];
    __shared__ uint32_t s_tile;
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[TILE * sizeof(KeyT)];
    KeyT *s_keys = reinterpret_cast<KeyT *>(s_raw);
    uint32_t *s_vals = reinterpret_cast<uint32_t *>(s_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
    for (int i = tid; i < NW * RADIX; i += THREADS) s_hist[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (uint32_t) TILE;
    const uint32_t valid = min((uint32_t) TILE, n - base);

    // ---- load (wave-striped: item j of lane l = wave chunk + j*64 + l) ----
    KeyT key[ITEMS];
    uint32_t val[ITEMS];
    const uint32_t wbase = base + wave * WAVE_ITEMS + lane;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wbase + j * 64;
        key[j] = (i < n) ? keys_in[i] : ~(KeyT) 0;
    }
    // (values are loaded after the ranking phase: fewer live registers while
    // ranking, and the loads overlap the look-back)

    // ---- rank within the wave by ballot matching ---------------------------
    uint32_t rank[ITEMS];
    uint32_t *wh = s_hist + wave * RADIX;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = (uint32_t) (key[j] >> shift) & (RADIX - 1);
        uint32_t below, cnt;
        match_digit<RADIX_BITS>(d, &below, &cnt);
        const uint32_t old = wh[d];
        if (below == 0) wh[d] = old + cnt;
        rank[j] = old + below;
    }
    __syncthreads();

    // ---- per-digit totals, wave offsets, look-back --------------------------
    uint32_t tot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint32_t c = s_hist[w * RADIX + tid];
            s_hist[w * RADIX + tid] = tot;
            tot += c;
        }
    }
    uint32_t count = tot;
    if (tid == RADIX - 1) count -= ((uint32_t) TILE - valid);   // padding keys
    W *lb = lookback + (uint64_t) tile * RADIX + tid;
    if (tid < RADIX && tile >= nseed)
        lb_store<W>(lb, Lb<W>::pack(gen, tile == 0 ? LB_PREFIX : LB_AGG, count));

#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wbase + j * 64;
        if (IDENTITY_VALS) val[j] = i;
        else val[j] = (i < n) ? vals_in[i] : 0u;
    }

    // exclusive scan of the tile's digit counts (threads >= RADIX contribute 0)
    uint32_t dbase;
    {
        uint32_t incl = wave_inclusive_scan(tot);
        if (lane == 63 && wave < RADIX / 64) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int i = 0; i < RADIX / 64; ++i)
            if (i < wave) woff += s_tmp[i];
        dbase = woff + incl - tot;
    }

    // ---- reorder the tile through LDS (needs only tile-local offsets, so it runs
    // while the predecessors make progress on their prefixes) ----------------------
    if (tid < RADIX) s_digit_base[tid] = dbase;
    __syncthreads();
    uint32_t pos[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = (uint32_t) (key[j] >> shift) & (RADIX - 1);
        pos[j] = s_digit_base[d] + wh[d] + rank[j];
        s_keys[pos[j]] = key[j];
    }

    if (tid < RADIX) {
        uint32_t excl = 0;
        if (tile < nseed) {
            // inclusive prefix was published before the pass started
            excl = Lb<W>::value(lb_load<W>(lb)) - count;
        } else if (tile > 0 && !(dbg & 2)) {
            // serial walk back over the predecessors (issuing several polls at once
            // was measured slower: the polls themselves are the scarce resource)
            int64_t t = (int64_t) tile - 1;
            uint32_t spins = 0;
            const uint32_t want_p = Lb<W>::want(gen, LB_PREFIX);
            const uint32_t want_a = Lb<W>::want(gen, LB_AGG);
            while (true) {
                const W w = lb_load<W>(lookback + (uint64_t) t * RADIX + tid);
                const uint32_t hi = Lb<W>::tag(w);
                if (hi == want_p) { excl += Lb<W>::value(w); break; }
                if (hi == want_a) { excl += Lb<W>::value(w); --t; continue; }
                if (++spins > LOOKBACK_SPIN_LIMIT) {
                    atomicExch(&status->lookback_timeout, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            lb_store<W>(lb, Lb<W>::pack(gen, LB_PREFIX, excl + count));
        }
        s_global_base[tid] = (dbg & 1) ? base : digit_start[tid] + excl - dbase;
    }
    __syncthreads();

    // ---- write coalesced runs ----------------------------------------------------

    uint32_t gpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t p = k * THREADS + tid;
        const KeyT kk = s_keys[p];
        const uint32_t d = (uint32_t) (kk >> shift) & (RADIX - 1);
        gpos[k] = s_global_base[d] + p;
        if (p < valid) keys_out[gpos[k]] = kk;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) s_vals[pos[j]] = val[j];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t p = k * THREADS + tid;
        if (p < valid) vals_out[gpos[k]] = s_vals[p];
    }
}

template <class KeyT, class Tr, class W>
int radix_sort_pairs_w(bt_context *ctx, KeyT *ka, uint32_t *va, KeyT *kb, uint32_t *vb,
                       int64_t n, int begin_bit, int end_bit, bool identity_vals, bool *in_b)
{
    // W = uint32_t: one zeroed look-back array per pass; uint64_t: one array, passes
    // told apart by the generation in the word
    constexpr bool PER_PASS = sizeof(W) == 4;
    constexpr int TILE = Tr::THREADS * Tr::ITEMS;
    constexpr int MAXP = (int) sizeof(KeyT);   // at most one pass per key byte

    *in_b = false;
    KeyT *copy_out = (KeyT *) ctx->sort_copy_keys;
    ctx->sort_copy_keys = nullptr;
    ctx->last_sort_passes = 0;
    ctx->last_sort_pass_ms = 0.f;
    if (n >= ((int64_t) 1 << 31)) {
        set_error("radix sort: n=%lld exceeds the 2^31-1 limit of 32-bit ids", (long long) n);
        return BT_ERR_INVALID;
    }
    if (begin_bit < 0 || end_bit > (int) sizeof(KeyT) * 8 || begin_bit > end_bit) {
        set_error("radix sort: bad bit range [%d, %d)", begin_bit, end_bit);
        return BT_ERR_INVALID;
    }
    const int npasses = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
    if (n == 0) return BT_OK;
    if (copy_out && npasses == 0) {
        set_error("radix sort: a key copy needs at least one pass");
        return BT_ERR_INVALID;
    }
    if (npasses == 0) {
        if (identity_vals) {
            set_error("radix sort: identity values need at least one pass");
            return BT_ERR_INVALID;
        }
        return BT_OK;
    }

    const uint32_t ntiles = (uint32_t) div_up(n, TILE);
    Buf<uint32_t> hist;      // [npasses][RADIX] then tile counters [npasses]
    Buf<W> lookback;         // [ntiles][RADIX] (x npasses for 32-bit words)
    const int64_t lb_per_pass = (int64_t) ntiles * RADIX;
    // (the histograms come from the call's zeroed block when it has room)
    const int64_t hist_words = (int64_t) npasses * RADIX + MAXP;
    uint32_t *hist_zero = (uint32_t *) zero_alloc(ctx, (size_t) hist_words * 4);
    if (hist_zero) hist.set_external(hist_zero, hist_words);
    else BT_CHECK(hist.alloc(ctx->pool, hist_words));
    BT_CHECK(lookback.alloc(ctx->pool, lb_per_pass * (PER_PASS ? npasses : 1)));
    uint32_t *tile_counters = hist.get() + (int64_t) npasses * RADIX;

    // timing of the 64-bit-key sorts (the roofline figure is quoted on them): four
    // events per context, recorded here and read in bt_get_sort_stats -- the sort itself
    // never waits for the device
    const bool timed = sizeof(KeyT) == 8;
    hipEvent_t *ev = (hipEvent_t *) ctx->sort_ev;
    if (timed && !ev[0])
        for (int i = 0; i < 4; ++i) BT_HIP_CHECK(hipEventCreate(&ev[i]));

    if (!hist_zero)
        BT_HIP_CHECK(hipMemsetAsync(hist.get(), 0, ((size_t) npasses * RADIX + MAXP) * 4, ctx->stream));
    BT_HIP_CHECK(hipMemsetAsync(lookback.get(), 0,
                                (size_t) lb_per_pass * (PER_PASS ? npasses : 1) * sizeof(W), ctx->stream));

    if (timed) BT_HIP_CHECK(hipEventRecord(ev[0], ctx->stream));
    {
        int64_t blocks = div_up(n, 256 * 16);
        int64_t cap = (int64_t) ctx->num_cus * 8;
        if (blocks > cap) blocks = cap;
        sort_hist_kernel<KeyT, MAXP><<<(unsigned) blocks, 256, 0, ctx->stream>>>(
            ka, (uint32_t) n, begin_bit, npasses, hist.get(), copy_out);
        sort_hist_scan_kernel<<<npasses, RADIX, 0, ctx->stream>>>(hist.get());
    }
    if (timed) BT_HIP_CHECK(hipEventRecord(ev[1], ctx->stream));

    // seeding measured neutral at 1e8 keys; off unless BT_SORT_DBG & 8
    const uint32_t nseed = (sort_dbg() & 8) ? std::min<uint32_t>(ntiles, (uint32_t) MAX_SEED) : 0u;
    Buf<uint32_t> tile_hist;
    BT_CHECK(tile_hist.alloc(ctx->pool, (int64_t) nseed * RADIX));

    KeyT *kin = ka, *kout = kb;
    uint32_t *vin = va, *vout = vb;
    for (int p = 0; p < npasses; ++p) {
        const int shift = begin_bit + p * RADIX_BITS;
        W *lbp = lookback.get() + (PER_PASS ? lb_per_pass * p : 0);
        if (nseed > 0) {
            seed_hist_kernel<KeyT, TILE><<<nseed, 256, 0, ctx->stream>>>(
                kin, (uint32_t) n, shift, tile_hist.get());
            seed_scan_kernel<W><<<RADIX, MAX_SEED, 0, ctx->stream>>>(tile_hist.get(), nseed, lbp,
                                                                    (uint32_t) (p + 1));
        }
        if (p == 0 && identity_vals) {
            onesweep_kernel<KeyT, W, Tr::THREADS, Tr::ITEMS, Tr::MINW, true>
                <<<ntiles, Tr::THREADS, 0, ctx->stream>>>(
                    kin, vin, kout, vout, (uint32_t) n, shift, hist.get() + p * RADIX,
                    lbp, tile_counters + p, (uint32_t) (p + 1), ctx->d_status, sort_dbg(), nseed);
        } else {
            onesweep_kernel<KeyT, W, Tr::THREADS, Tr::ITEMS, Tr::MINW, false>
                <<<ntiles, Tr::THREADS, 0, ctx->stream>>>(
                    kin, vin, kout, vout, (uint32_t) n, shift, hist.get() + p * RADIX,
                    lbp, tile_counters + p, (uint32_t) (p + 1), ctx->d_status, sort_dbg(), nseed);
        }
        KeyT *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
        if (timed && p == 0) BT_HIP_CHECK(hipEventRecord(ev[2], ctx->stream));
    }
    BT_HIP_CHECK(hipGetLastError());
    if (timed) {
        BT_HIP_CHECK(hipEventRecord(ev[3], ctx->stream));
        ctx->sort_ev_pending = true;
        ctx->sort_ev_passes = npasses;
        ctx->sort_ev_first_identity = identity_vals ? 1 : 0;
        ctx->sort_ev_n = n;
        ctx->sort_ev_key_bytes = (int) sizeof(KeyT) + 4;     // a pair: key + 32-bit value
        ctx->sort_ev_digit_bits = RADIX_BITS;
    }
    ctx->last_sort_passes = npasses;
    ctx->last_sort_n = n;
    *in_b = (npasses & 1) != 0;
    return BT_OK;
}


// ---- keys-only passes: the value rides in the low bits of the key ------------------------
// A tree build of n < 2^idbits point particles sorts ONE 64-bit word per particle,
// (Morton path of the levels the tree can reach) << idbits | user id: a stable LSD sort
// over the path bits alone leaves equal paths in id order, which is exactly what the
// pair sort's stability delivers, and a pass moves 16 bytes per particle instead of 24.
// The digit width RB is 8 or 9 bits (36 path bits = 12 levels in 3D are 4 passes of 9).
// Same structure as onesweep_kernel: ballot ranking, per-wave digit counters (16 bits:
// a wave holds at most 64 * ITEMS keys), decoupled look-back, LDS reorder, runs written
// in order.

template <int RB>
__global__ __launch_bounds__(256) void keys_hist_kernel(const uint64_t *keys, uint32_t n,
        int begin_bit, int npasses, uint32_t *ghist /* [npasses][1 << RB] */)
{
    constexpr int R = 1 << RB;
    constexpr int MAXP = 8;
    __shared__ uint32_t s_h[MAXP * R];
    for (int i = threadIdx.x; i < npasses * R; i += 256) s_h[i] = 0;
    __syncthreads();
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint64_t k = keys[i] >> begin_bit;
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
            if (p < npasses) {
                atomicAdd(&s_h[p * R + (uint32_t) (k & (R - 1))], 1u);
                k >>= RB;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < npasses * R; i += 256) {
        uint32_t c = s_h[i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

template <int R>
__global__ __launch_bounds__(R) void keys_hist_scan_kernel(uint32_t *ghist)
{
    __shared__ uint32_t s_tmp[R / 64 + 1];
    uint32_t *h = ghist + blockIdx.x * R;
    uint32_t v = h[threadIdx.x];
    uint32_t ex = block_exclusive_scan<uint32_t, R>(v, s_tmp, (uint32_t *) nullptr);
    h[threadIdx.x] = ex;
}

template <class W, int THREADS, int ITEMS, int MINW, int RB>
__global__ __launch_bounds__(THREADS, MINW) void onesweep_keys_kernel(
        const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys_out,
        uint32_t n, int shift, const uint32_t *__restrict__ digit_start,
        W *lookback, uint32_t *tile_counter, uint32_t gen, DeviceStatus *status)
{
    constexpr int R = 1 << RB;
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * ITEMS;
    constexpr int WAVE_ITEMS = 64 * ITEMS;
    static_assert(R <= THREADS, "one thread per digit");
    static_assert(WAVE_ITEMS < 65536, "16-bit per-wave digit counters");

    __shared__ uint16_t s_hist[NW * R];
    __shared__ uint32_t s_digit_base[R];
    __shared__ uint32_t s_global_base[R];
    __shared__ uint32_t s_tmp[R / 64 + 1];
    __shared__ uint32_t s_tile;
    __shared__ __attribute__((aligned(16))) uint64_t s_keys[TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);
    for (int i = tid; i < NW * R; i += THREADS) s_hist[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t base = tile * (uint32_t) TILE;
    const uint32_t valid = min((uint32_t) TILE, n - base);

    uint64_t key[ITEMS];
    const uint32_t wbase = base + wave * WAVE_ITEMS + lane;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        uint32_t i = wbase + j * 64;
        key[j] = (i < n) ? keys_in[i] : ~(uint64_t) 0;
    }

    // ---- rank within the wave by ballot matching ---------------------------
    uint32_t rank[ITEMS];
    uint16_t *wh = s_hist + wave * R;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = (uint32_t) (key[j] >> shift) & (R - 1);
        uint32_t below, cnt;
        match_digit<RB>(d, &below, &cnt);
        const uint32_t old = wh[d];
        if (below == 0) wh[d] = (uint16_t) (old + cnt);
        rank[j] = old + below;
    }
    __syncthreads();

    // ---- per-digit totals, wave offsets, look-back --------------------------
    uint32_t tot = 0;
    if (tid < R) {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint32_t c = s_hist[w * R + tid];
            s_hist[w * R + tid] = (uint16_t) tot;
            tot += c;
        }
    }
    uint32_t count = tot;
    if (tid == R - 1) count -= ((uint32_t) TILE - valid);   // padding keys
    W *lb = lookback + (uint64_t) tile * R + tid;
    if (tid < R)
        lb_store<W>(lb, Lb<W>::pack(gen, tile == 0 ? LB_PREFIX : LB_AGG, count));

    uint32_t dbase;
    {
        uint32_t incl = wave_inclusive_scan(tot);
        if (lane == 63 && wave < R / 64) s_tmp[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int i = 0; i < R / 64; ++i)
            if (i < wave) woff += s_tmp[i];
        dbase = woff + incl - tot;
    }

    if (tid < R) s_digit_base[tid] = dbase;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const uint32_t d = (uint32_t) (key[j] >> shift) & (R - 1);
        s_keys[s_digit_base[d] + wh[d] + rank[j]] = key[j];
    }

    if (tid < R) {
        uint32_t excl = 0;
        if (tile > 0) {
            int64_t t = (int64_t) tile - 1;
            uint32_t spins = 0;
            const uint32_t want_p = Lb<W>::want(gen, LB_PREFIX);
            const uint32_t want_a = Lb<W>::want(gen, LB_AGG);
            while (true) {
                const W w = lb_load<W>(lookback + (uint64_t) t * R + tid);
                const uint32_t hi = Lb<W>::tag(w);
                if (hi == want_p) { excl += Lb<W>::value(w); break; }
                if (hi == want_a) { excl += Lb<W>::value(w); --t; continue; }
                if (++spins > LOOKBACK_SPIN_LIMIT) {
                    atomicExch(&status->lookback_timeout, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            lb_store<W>(lb, Lb<W>::pack(gen, LB_PREFIX, excl + count));
        }
        s_global_base[tid] = digit_start[tid] + excl - dbase;
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t p = k * THREADS + tid;
        const uint64_t kk = s_keys[p];
        const uint32_t d = (uint32_t) (kk >> shift) & (R - 1);
        if (p < valid) keys_out[s_global_base[d] + p] = kk;
    }
}

template <class W, class Tr, int RB>
int radix_sort_keys_w(bt_context *ctx, uint64_t *ka, uint64_t *kb, int64_t n, int begin_bit,
                      int npasses, bool *in_b)
{
    constexpr bool PER_PASS = sizeof(W) == 4;
    constexpr int R = 1 << RB;
    constexpr int TILE = Tr::THREADS * Tr::ITEMS;
    const uint32_t ntiles = (uint32_t) div_up(n, TILE);
    Buf<uint32_t> hist;
    Buf<W> lookback;
    const int64_t lb_per_pass = (int64_t) ntiles * R;
    const int64_t hist_words = (int64_t) npasses * R + 8;
    uint32_t *hist_zero = (uint32_t *) zero_alloc(ctx, (size_t) hist_words * 4);
    if (hist_zero) hist.set_external(hist_zero, hist_words);
    else BT_CHECK(hist.alloc(ctx->pool, hist_words));
    BT_CHECK(lookback.alloc(ctx->pool, lb_per_pass * (PER_PASS ? npasses : 1)));
    uint32_t *tile_counters = hist.get() + (int64_t) npasses * R;
    hipEvent_t *ev = (hipEvent_t *) ctx->sort_ev;
    if (!ev[0])
        for (int i = 0; i < 4; ++i) BT_HIP_CHECK(hipEventCreate(&ev[i]));
    if (!hist_zero)
        BT_HIP_CHECK(hipMemsetAsync(hist.get(), 0, (size_t) hist_words * 4, ctx->stream));
    BT_HIP_CHECK(hipMemsetAsync(lookback.get(), 0,
                                (size_t) lb_per_pass * (PER_PASS ? npasses : 1) * sizeof(W), ctx->stream));
    BT_HIP_CHECK(hipEventRecord(ev[0], ctx->stream));
    {
        int64_t blocks = div_up(n, 256 * 16);
        int64_t cap = (int64_t) ctx->num_cus * 8;
        if (blocks > cap) blocks = cap;
        keys_hist_kernel<RB><<<(unsigned) blocks, 256, 0, ctx->stream>>>(
            ka, (uint32_t) n, begin_bit, npasses, hist.get());
        keys_hist_scan_kernel<R><<<npasses, R, 0, ctx->stream>>>(hist.get());
    }
    BT_HIP_CHECK(hipEventRecord(ev[1], ctx->stream));
    uint64_t *kin = ka, *kout = kb;
    for (int p = 0; p < npasses; ++p) {
        const int shift = begin_bit + p * RB;
        W *lbp = lookback.get() + (PER_PASS ? lb_per_pass * p : 0);
        onesweep_keys_kernel<W, Tr::THREADS, Tr::ITEMS, Tr::MINW, RB>
            <<<ntiles, Tr::THREADS, 0, ctx->stream>>>(
                kin, kout, (uint32_t) n, shift, hist.get() + p * R, lbp, tile_counters + p,
                (uint32_t) (p + 1), ctx->d_status);
        std::swap(kin, kout);
        if (p == 0) BT_HIP_CHECK(hipEventRecord(ev[2], ctx->stream));
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_HIP_CHECK(hipEventRecord(ev[3], ctx->stream));
    ctx->sort_ev_pending = true;
    ctx->sort_ev_passes = npasses;
    ctx->sort_ev_first_identity = 0;
    ctx->sort_ev_n = n;
    ctx->sort_ev_key_bytes = 8;       // keys only: 16 bytes per particle and pass
    ctx->sort_ev_digit_bits = RB;
    ctx->last_sort_passes = npasses;
    ctx->last_sort_n = n;
    *in_b = (npasses & 1) != 0;
    return BT_OK;
}

// Tile shape of the keys-only pass: 1024 x 16 keys = 128 KiB of
// LDS, one workgroup per CU -- measured at 10^8 keys, 9-bit digits: 0.468 ms per pass
// against 0.520 for 512 x 16 (two workgroups per CU), 0.53-0.76 for the smaller tiles:
// digit runs twice as long (32 keys = 256 bytes at 512 digits) and half the look-back rows
struct KCfg0 { static constexpr int THREADS = 1024, ITEMS = 16, MINW = 4; };

template <class Tr>
static int radix_sort_keys_cfg(bt_context *ctx, uint64_t *ka, uint64_t *kb, int64_t n,
                               int begin_bit, int npasses, int rb, bool *in_b)
{
    const bool w32 = n < ((int64_t) 1 << 30) && !(sort_dbg() & 16);
    if (rb == 8)
        return w32 ? radix_sort_keys_w<uint32_t, Tr, 8>(ctx, ka, kb, n, begin_bit, npasses, in_b)
                   : radix_sort_keys_w<uint64_t, Tr, 8>(ctx, ka, kb, n, begin_bit, npasses, in_b);
    return w32 ? radix_sort_keys_w<uint32_t, Tr, 9>(ctx, ka, kb, n, begin_bit, npasses, in_b)
               : radix_sort_keys_w<uint64_t, Tr, 9>(ctx, ka, kb, n, begin_bit, npasses, in_b);
}

int radix_sort_keys_plan(int nbits, int *digit_bits)
{
    // fewest passes; at equal pass counts the narrower digit (longer runs per digit)
    const int p8 = (nbits + 7) / 8, p9 = (nbits + 8) / 9;
    static const int force = [] { const char *e = getenv("BT_SORT_KEYS_RB"); return e ? atoi(e) : 0; }();
    int rb = p9 < p8 ? 9 : 8;
    if (force == 8 || force == 9) rb = force;
    *digit_bits = rb;
    return (nbits + rb - 1) / rb;
}

int radix_sort_keys(bt_context *ctx, uint64_t *ka, uint64_t *kb, int64_t n, int begin_bit,
                    int end_bit, bool *in_b)
{
    *in_b = false;
    ctx->last_sort_passes = 0;
    if (n >= ((int64_t) 1 << 31)) {
        set_error("radix sort: n=%lld exceeds the 2^31-1 limit of 32-bit ids", (long long) n);
        return BT_ERR_INVALID;
    }
    if (begin_bit < 0 || end_bit > 64 || begin_bit > end_bit) {
        set_error("radix sort: bad bit range [%d, %d)", begin_bit, end_bit);
        return BT_ERR_INVALID;
    }
    if (n == 0 || end_bit == begin_bit) return BT_OK;
    int rb = 8;
    const int npasses = radix_sort_keys_plan(end_bit - begin_bit, &rb);
    return radix_sort_keys_cfg<KCfg0>(ctx, ka, kb, n, begin_bit, npasses, rb, in_b);
}

template <class KeyT, class Tr>
int radix_sort_pairs_cfg(bt_context *ctx, KeyT *ka, uint32_t *va, KeyT *kb, uint32_t *vb,
                         int64_t n, int begin_bit, int end_bit, bool identity_vals, bool *in_b)
{
    // 32-bit look-back words hold counts below 2^30 (BT_SORT_DBG & 16: always 64-bit)
    if (n < ((int64_t) 1 << 30) && !(sort_dbg() & 16))
        return radix_sort_pairs_w<KeyT, Tr, uint32_t>(ctx, ka, va, kb, vb, n, begin_bit, end_bit,
                                                      identity_vals, in_b);
    return radix_sort_pairs_w<KeyT, Tr, uint64_t>(ctx, ka, va, kb, vb, n, begin_bit, end_bit,
                                                  identity_vals, in_b);
}

template <class KeyT>
int radix_sort_pairs(bt_context *ctx, KeyT *ka, uint32_t *va, KeyT *kb, uint32_t *vb,
                     int64_t n, int begin_bit, int end_bit, bool identity_vals, bool *in_b)
{
    return radix_sort_pairs_cfg<KeyT, Cfg0>(ctx, ka, va, kb, vb, n, begin_bit, end_bit, identity_vals, in_b);
}

template int radix_sort_pairs<uint64_t>(bt_context *, uint64_t *, uint32_t *, uint64_t *,
                                        uint32_t *, int64_t, int, int, bool, bool *);
template int radix_sort_pairs<uint32_t>(bt_context *, uint32_t *, uint32_t *, uint32_t *,
                                        uint32_t *, int64_t, int, int, bool, bool *);

}  // namespace bt

// ---- C API -------------------------------------------------------------------

template <class KeyT>
static int sort_api(bt_context *ctx, KeyT *keys_in, uint32_t *vals_in, KeyT *keys_out,
                    uint32_t *vals_out, int64_t n, int begin_bit, int end_bit)
{
    if (!ctx || n < 0 || (n > 0 && (!keys_in || !vals_in || !keys_out || !vals_out))) {
        bt::set_error("bt_radix_sort: NULL argument or negative n");
        return BT_ERR_INVALID;
    }
    if (n == 0) return BT_OK;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_CHECK(bt::zero_begin(ctx));          // (resets the status word too)
    bool in_b = false;
    BT_CHECK(bt::radix_sort_pairs<KeyT>(ctx, keys_in, vals_in, keys_out, vals_out, n,
                                        begin_bit, end_bit, false, &in_b));
    if (!in_b && n > 0) {
        BT_HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, (size_t) n * sizeof(KeyT),
                                    hipMemcpyDeviceToDevice, ctx->stream));
        BT_HIP_CHECK(hipMemcpyAsync(vals_out, vals_in, (size_t) n * 4,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    }
    return bt::check_status(ctx);
}

extern "C" {

int bt_radix_sort_u64_u32(bt_context *ctx, uint64_t *keys_in, uint32_t *vals_in,
                          uint64_t *keys_out, uint32_t *vals_out, int64_t n,
                          int begin_bit, int end_bit)
{
    bt::CallScope bt_call_scope_(ctx);
    return sort_api<uint64_t>(ctx, keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit);
}

int bt_radix_sort_u32_u32(bt_context *ctx, uint32_t *keys_in, uint32_t *vals_in,
                          uint32_t *keys_out, uint32_t *vals_out, int64_t n,
                          int begin_bit, int end_bit)
{
    bt::CallScope bt_call_scope_(ctx);
    return sort_api<uint32_t>(ctx, keys_in, vals_in, keys_out, vals_out, n, begin_bit, end_bit);
}

int bt_radix_sort_u64_keys(bt_context *ctx, uint64_t *keys_in, uint64_t *keys_out, int64_t n,
                           int begin_bit, int end_bit)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || n < 0 || (n > 0 && (!keys_in || !keys_out))) {
        bt::set_error("bt_radix_sort_u64_keys: NULL argument or negative n");
        return BT_ERR_INVALID;
    }
    if (n == 0) return BT_OK;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_CHECK(bt::zero_begin(ctx));
    bool in_b = false;
    BT_CHECK(bt::radix_sort_keys(ctx, keys_in, keys_out, n, begin_bit, end_bit, &in_b));
    if (!in_b)
        BT_HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, (size_t) n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    return bt::check_status(ctx);
}

int bt_get_sort_stats(bt_context *ctx, bt_sort_stats *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !out) return BT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    if (!ctx->sort_ev_pending) return BT_OK;
    hipEvent_t *ev = (hipEvent_t *) ctx->sort_ev;
    BT_HIP_CHECK(hipEventSynchronize(ev[3]));
    float hist_ms = 0.f, first_ms = 0.f, rest_ms = 0.f;
    BT_HIP_CHECK(hipEventElapsedTime(&hist_ms, ev[0], ev[1]));
    BT_HIP_CHECK(hipEventElapsedTime(&first_ms, ev[1], ev[2]));
    BT_HIP_CHECK(hipEventElapsedTime(&rest_ms, ev[2], ev[3]));
    const int np = ctx->sort_ev_passes;
    out->n = ctx->sort_ev_n;
    out->passes = np;
    out->hist_ms = hist_ms;
    out->total_ms = hist_ms + first_ms + rest_ms;
    out->pass_ms_avg = np > 0 ? (first_ms + rest_ms) / np : 0.f;
    out->first_pass_ms = first_ms;
    out->first_pass_identity = ctx->sort_ev_first_identity;
    out->bytes_per_element_per_pass = 2 * ctx->sort_ev_key_bytes;
    out->digit_bits = ctx->sort_ev_digit_bits;
    if (ctx->sort_ev_first_identity) {
        out->full_passes = np - 1;
        out->full_pass_ms_avg = np > 1 ? rest_ms / (np - 1) : 0.f;
    } else {
        out->full_passes = np;
        out->full_pass_ms_avg = out->pass_ms_avg;
    }
    return BT_OK;
}

}  // extern "C"
