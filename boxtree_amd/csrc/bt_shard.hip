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
    const T *rootbox;       // device {min[3], max[3], ...}: takes the place of bmin / bmax
    int64_t n;
    int level;
};

// same float expression as the key kernel (tbk:374-376) at level `level`
template <class T, int D>
__device__ __forceinline__ uint32_t morton_cell_of(const T (&x)[D], const T (&gmin)[D], const T (&gext)[D], int level)
{
    uint32_t cell = 0;
    const uint32_t top = (1u << level) - 1u;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        uint32_t v = (uint32_t) (((x[ax] - gmin[ax]) / gext[ax]) * (T) (1u << level));
        v = v > top ? top : v;
        for (int b = 0; b < level; ++b)
            cell |= ((v >> b) & 1u) << (D * b + (D - 1 - ax));       // x most significant
    }
    return cell;
}

template <class T, int D>
__global__ __launch_bounds__(256) void morton_cells_kernel(CellArgs<T, D> a, uint32_t *cells,
                                                           int32_t *hist)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    T gmin[D], gext[D], x[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        gmin[ax] = a.rootbox ? a.rootbox[ax] : a.bmin[ax];
        gext[ax] = (a.rootbox ? a.rootbox[3 + ax] : a.bmax[ax]) - gmin[ax];
        x[ax] = a.x[ax][i];
    }
    const uint32_t cell = morton_cell_of<T, D>(x, gmin, gext, a.level);
    cells[i] = cell;
    atomicAdd(&hist[cell], 1);
}

// same, with the histogram privatised in LDS (up to 2^15 cells = 128 KiB, one
// workgroup per CU): 1e8 global atomics on 32 K addresses cost ~5 ms otherwise.  One
// workgroup per CU is 16 waves: four particles per lane and trip keep enough loads in flight
// (one per trip: 0.86 ms at 10^8 points, 3.2 TB/s).
constexpr int CELLS_LDS_MAX = 1 << 15;
constexpr int CELLS_UNROLL = 4;

template <class T, int D>
__global__ __launch_bounds__(1024) void morton_cells_lds_kernel(CellArgs<T, D> a, uint32_t *cells,
                                                                int32_t *hist, int ncells)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];
    for (int c = threadIdx.x; c < ncells; c += 1024) s_hist[c] = 0;
    T gmin[D], gext[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        gmin[ax] = a.rootbox ? a.rootbox[ax] : a.bmin[ax];
        gext[ax] = (a.rootbox ? a.rootbox[3 + ax] : a.bmax[ax]) - gmin[ax];
    }
    __syncthreads();
    const int64_t stride = (int64_t) gridDim.x * 1024 * CELLS_UNROLL;
    for (int64_t i0 = (int64_t) blockIdx.x * 1024 * CELLS_UNROLL + threadIdx.x; i0 < a.n; i0 += stride) {
        T x[CELLS_UNROLL][D];
#pragma unroll
        for (int u = 0; u < CELLS_UNROLL; ++u) {
            const int64_t i = i0 + (int64_t) u * 1024;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) x[u][ax] = i < a.n ? a.x[ax][i] : gmin[ax];
        }
#pragma unroll
        for (int u = 0; u < CELLS_UNROLL; ++u) {
            const int64_t i = i0 + (int64_t) u * 1024;
            if (i < a.n) {
                const uint32_t cell = morton_cell_of<T, D>(x[u], gmin, gext, a.level);
                cells[i] = cell;
                atomicAdd(&s_hist[cell], 1u);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncells; c += 1024) {
        const uint32_t v = s_hist[c];
        if (v) atomicAdd(&hist[c], (int32_t) v);
    }
}

// Refine weights per cell (int64: a cell's weight is the sum the reference saturates at
// INT_MAX, tbk:270-274 -- exact here, compared against the limit as a 64-bit number): 64-bit
// counters privatised in LDS, the cells in slices of 2^14 so that a slice's counters fit
// (blockIdx.y = slice; every slice reads all cells and weights: 8 bytes per particle each).
// (128 KiB of static LDS here, up to ~92 KiB of dynamic LDS in pp_scatter_kernel with many owners
// and 5-value records: sized for the 160 KiB of a gfx950 CU, the only target of this library --
// csrc/Makefile builds --offload-arch=gfx950 and nothing else)
constexpr int WH_SLICE = 1 << 14;
static_assert(WH_SLICE * 8 <= 160 * 1024, "weight_hist_kernel: the slice must fit the 160 KiB of LDS of a gfx950 CU");

__global__ __launch_bounds__(1024) void weight_hist_kernel(const uint32_t *cells, const int32_t *weights,
        int64_t n, int ncells, unsigned long long *whist)
{
    __shared__ unsigned long long s_w[WH_SLICE];
    const uint32_t lo = blockIdx.y * (uint32_t) WH_SLICE;
    const uint32_t cnt = min((uint32_t) WH_SLICE, (uint32_t) ncells - lo);
    for (uint32_t c = threadIdx.x; c < cnt; c += 1024) s_w[c] = 0ull;
    __syncthreads();
    const int64_t stride = (int64_t) gridDim.x * 1024;
    for (int64_t i = (int64_t) blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        const uint32_t c = cells[i] - lo;
        if (c < cnt) atomicAdd(&s_w[c], (unsigned long long) (weights ? weights[i] : 1));
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < cnt; c += 1024) {
        const unsigned long long v = s_w[c];
        if (v) atomicAdd(&whist[lo + c], v);
    }
}

// a 32-bit weight in the low bits of a record value (the partition moves values of one width)
template <class U>
__global__ __launch_bounds__(256) void widen_weights_kernel(int64_t n, const int32_t *w, U *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (U) (uint32_t) (w ? w[i] : 1);
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
               const double *bmax, const void *d_rootbox, int level, uint32_t *cells, int32_t *hist,
               bool wait)
{
    CellArgs<T, D> a{};
    for (int ax = 0; ax < D; ++ax) {
        a.x[ax] = (const T *) coords[ax];
        a.bmin[ax] = bmin ? (T) bmin[ax] : (T) 0;
        a.bmax[ax] = bmax ? (T) bmax[ax] : (T) 0;
    }
    a.rootbox = (const T *) d_rootbox;
    a.n = n;
    a.level = level;
    const int ncells = 1 << (D * level);
    if (n > 0 && ncells <= CELLS_LDS_MAX) {
        const unsigned blocks = (unsigned) std::min<int64_t>(div_up(n, 1024 * CELLS_UNROLL), ctx->num_cus);
        morton_cells_lds_kernel<T, D><<<blocks, 1024, (size_t) ncells * 4, ctx->stream>>>(
            a, cells, hist, ncells);
    } else if (n > 0) {
        morton_cells_kernel<T, D><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(a, cells, hist);
    }
    BT_HIP_CHECK(hipGetLastError());
    if (wait) BT_CHECK(bt::sync_stream(ctx));
    return BT_OK;
}

template <class T>
int cells_dims(bt_context *ctx, int dims, const void *const *coords, int64_t n, const double *bmin,
               const double *bmax, const void *d_rootbox, int level, uint32_t *cells, int32_t *hist, bool wait)
{
    switch (dims) {
    case 1: return cells_impl<T, 1>(ctx, coords, n, bmin, bmax, d_rootbox, level, cells, hist, wait);
    case 2: return cells_impl<T, 2>(ctx, coords, n, bmin, bmax, d_rootbox, level, cells, hist, wait);
    default: return cells_impl<T, 3>(ctx, coords, n, bmin, bmax, d_rootbox, level, cells, hist, wait);
    }
}

// ---- stable partition of the particles by owner rank, payload carried along -------------
//
// The send buffer of the exchange, made in one sweep over the coordinates: a wave owns
// PP_WAVE_ITEMS consecutive particles and counts them per owner (pp_count_kernel), a scan
// over (owner, wave) turns the counts into record offsets, and pp_scatter_kernel moves every
// particle's coordinates, interleaved, to its place -- the segment a rank keeps straight into
// the receive buffer.  The scatter issues all of a wave's loads first (cells, owners,
// coordinates: PP_ROWS x D values per lane in flight), ranks the rows in order into a staging
// area in LDS where the wave's particles of one owner are contiguous, and copies each owner's
// run out with 16-byte stores: the records of a wave leave as a few contiguous pieces
// instead of D strided 8-byte stores per particle (2.1 -> ~1 ms at 10^8 points, one to eight
// owners).  Reads are sequential, writes go to `nranks` advancing runs.  (A permutation by
// owner followed by a gather reads the coordinate arrays in `nranks` interleaved strided
// passes: with 8 ranks every 64-byte line is fetched 8 times.)
constexpr int PP_MAX_RANKS = BT_MGPU_MAX_RANKS;
constexpr int PP_ROWS = 8;                      // particles per lane
constexpr int PP_WAVE_ITEMS = 64 * PP_ROWS;
constexpr int PP_WAVES = 4;                     // waves per workgroup, each on its own

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

constexpr int PP_SMALL = 8;     // up to this many owners are ranked with ballots alone
constexpr int PP_TABLE_MAX = 1 << 15;   // cells whose owners fit the count kernel's LDS, a byte each

// counts[owner * nwaves + tile], and the owner of every particle as a byte for the scatter.
// A wave takes tiles in a grid-stride loop with the owner table in LDS (ncells > 0): looked up
// in memory, 10^8 random 4-byte reads of a 128-KiB table move a 128-byte line each through the
// L2s -- 0.3 ms per pass at 10^8 points, in this kernel and again in the scatter.
template <bool SMALL>
__global__ __launch_bounds__(64 * PP_WAVES) void pp_count_kernel(const uint32_t *cells, int64_t n,
        const int32_t *owner_of_cell, int ncells, int nranks, int bits, int64_t nwaves, int32_t *counts,
        uint8_t *owner_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_owner[];      // [ncells]
    __shared__ int32_t s_cnt[PP_WAVES][SMALL ? 1 : PP_MAX_RANKS];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < ncells; c += 64 * PP_WAVES) s_owner[c] = (uint8_t) owner_of_cell[c];
    __syncthreads();
    const int64_t stride = (int64_t) gridDim.x * PP_WAVES;
    for (int64_t wave = (int64_t) blockIdx.x * PP_WAVES + w; wave < nwaves; wave += stride) {
        const int64_t base = wave * PP_WAVE_ITEMS + lane;
        uint32_t d[PP_ROWS];
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) {
            const int64_t i = base + (int64_t) j * 64;
            d[j] = i < n ? cells[i] : 0u;
        }
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) {
            const int64_t i = base + (int64_t) j * 64;
            // (lanes past the end form a group of their own)
            d[j] = i < n ? (ncells > 0 ? (uint32_t) s_owner[d[j]] : (uint32_t) owner_of_cell[d[j]])
                         : (uint32_t) nranks;
            if (i < n) owner_out[i] = (uint8_t) d[j];
        }
        if constexpr (SMALL) {
            int32_t mine = 0;
            for (int r = 0; r < nranks; ++r) {
                int32_t c = 0;
#pragma unroll
                for (int j = 0; j < PP_ROWS; ++j) c += (int32_t) __popcll(__ballot(d[j] == (uint32_t) r));
                if (lane == r) mine = c;
            }
            if (lane < nranks) counts[(int64_t) lane * nwaves + wave] = mine;
        } else {
            for (int r = lane; r < nranks; r += 64) s_cnt[w][r] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < PP_ROWS; ++j) {
                const bool in = d[j] < (uint32_t) nranks;
                const uint64_t mask = pp_match(d[j], bits + 1);
                const bool leader = (mask & ((1ull << lane) - 1ull)) == 0ull;
                if (in && leader) s_cnt[w][d[j]] += (int32_t) __popcll(mask);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            for (int r = lane; r < nranks; r += 64) counts[(int64_t) r * nwaves + wave] = s_cnt[w][r];
            __builtin_amdgcn_wave_barrier();
        }
    }
}

struct PpScan {
    const int32_t *c;
    __device__ int64_t operator()(int64_t i) const { return (int64_t) c[i]; }
};

template <class U, int D>
struct PpArgs {
    const U *in[D];
    const uint8_t *owners;          // owner of every particle (pp_count_kernel)
    const int32_t *offsets;         // [nranks * nwaves + 1] exclusive scan of the counts
    int64_t n, nwaves;
    int nranks, bits, self_rank;
    int64_t self_delta;             // receive offset of the own segment minus its send offset
    const int64_t *self_offsets;    // or, on the device: {send offset, receive offset}
    U *send, *recv;
};

// The tables of one tile (a wave's PP_WAVE_ITEMS particles), from the scanned counts: s_base[r] =
// first slot of owner r's run in the wave's owner-major order, s_run[r] = its length, s_goff[r]
// = the run's first record in the send layout.  Returns the number of particles of the tile.
__device__ __forceinline__ int32_t pp_tile_tables(const int32_t *offsets, int64_t nwaves, int64_t wave, int nranks,
                                                  int lane, int32_t *s_base, int32_t *s_run, int32_t *s_goff)
{
    int32_t carry = 0;
    for (int r0 = 0; r0 < nranks; r0 += 64) {
        const int r = r0 + lane;
        int32_t g = 0, c = 0;
        if (r < nranks) {
            const int64_t at = (int64_t) r * nwaves + wave;
            g = offsets[at];
            c = offsets[at + 1] - g;
        }
        int32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (r < nranks) { s_base[r] = carry + incl - c; s_run[r] = c; s_goff[r] = g; }
        carry += __shfl(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return carry;
}

// slot[j] = place of row j's particle in the wave's owner-major, stable order (-1 past the end
// of the input).  SMALL (at most PP_SMALL owners): a row is ranked by one ballot per owner, no
// LDS traffic and no barriers; otherwise s_run (an owner's count) is counted down, rows last to
// first.  The SAME function ranks the payload sweep of the exchange and every later
// bt_mgpu_route over its plan: the send order is a pure function of (owners, offsets).
template <bool SMALL>
__device__ __forceinline__ void pp_rank_rows(const uint32_t (&d)[PP_ROWS], int nranks, int bits, int lane,
                                             const int32_t *s_base, int32_t *s_run, int32_t (&slot)[PP_ROWS])
{
    const uint64_t lt = (1ull << lane) - 1ull;
    if constexpr (SMALL) {
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) slot[j] = -1;
        for (int r = 0; r < nranks; ++r) {
            int32_t at = s_base[r];                     // (uniform)
#pragma unroll
            for (int j = 0; j < PP_ROWS; ++j) {
                const bool is = d[j] == (uint32_t) r;
                const uint64_t m = __ballot(is);
                if (is) slot[j] = at + (int32_t) __popcll(m & lt);
                at += (int32_t) __popcll(m);
            }
        }
    } else {
#pragma unroll
        for (int j = PP_ROWS - 1; j >= 0; --j) {
            const bool in = d[j] < (uint32_t) nranks;
            const uint64_t mask = pp_match(d[j], bits + 1);
            const uint64_t above = mask & ~(lt | (1ull << lane));
            int32_t left = 0, sb = 0;
            if (in) { left = s_run[d[j]]; sb = s_base[d[j]]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (in && above == 0ull) s_run[d[j]] = left - (int32_t) __popcll(mask);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            slot[j] = in ? sb + left - 1 - (int32_t) __popcll(above) : -1;
        }
    }
}

struct __attribute__((packed, aligned(4))) PpWords4 { uint32_t x, y, z, w; };

// dynamic LDS: per wave PP_WAVE_ITEMS records, then per wave three tables of `nr_pad` words
// (first record of an owner's run in the staging area, records ranked so far, global offset).
// One tile per wave (a grid-stride loop was measured: waves that start together load together
// and store together, 1.20 against 1.02 ms).  SMALL (at most PP_SMALL owners): a row is ranked
// by one ballot per owner, no LDS traffic and no barriers.
template <class U, int D, bool SMALL>
__global__ __launch_bounds__(64 * PP_WAVES) void pp_scatter_kernel(PpArgs<U, D> a, int nr_pad)
{
    constexpr int RW = D * (int) sizeof(U) / 4;             // 32-bit words per record
    extern __shared__ __attribute__((aligned(16))) uint32_t pp_lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t *stage = pp_lds + (size_t) w * (PP_WAVE_ITEMS * RW);
    int32_t *s_base = reinterpret_cast<int32_t *>(pp_lds + (size_t) PP_WAVES * (PP_WAVE_ITEMS * RW)) + (size_t) w * 3 * nr_pad;
    int32_t *s_run = s_base + nr_pad;
    int32_t *s_goff = s_run + nr_pad;
    const int64_t self_delta = a.self_offsets ? a.self_offsets[1] - a.self_offsets[0] : a.self_delta;
    const int64_t wave = (int64_t) blockIdx.x * PP_WAVES + w;
    if (wave < a.nwaves) {
        // every load of the tile first
        const int64_t base = wave * PP_WAVE_ITEMS + lane;
        uint32_t d[PP_ROWS];
        U v[PP_ROWS][D];
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) {
            const int64_t i = base + (int64_t) j * 64;
            d[j] = i < a.n ? (uint32_t) a.owners[i] : (uint32_t) a.nranks;
        }
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) {
            const int64_t i = base + (int64_t) j * 64;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) v[j][ax] = i < a.n ? a.in[ax][i] : (U) 0;
        }

        // the tile's counts per owner -> where an owner's run starts in the staging area; rows
        // in order: rank within the owner's run
        const int32_t carry = pp_tile_tables(a.offsets, a.nwaves, wave, a.nranks, lane, s_base, s_run, s_goff);
        int32_t slot[PP_ROWS];
        pp_rank_rows<SMALL>(d, a.nranks, a.bits, lane, s_base, s_run, slot);
#pragma unroll
        for (int j = 0; j < PP_ROWS; ++j) {
            if (slot[j] >= 0) {
                uint32_t *dst = stage + (size_t) slot[j] * RW;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    if constexpr (sizeof(U) == 8) {
                        dst[2 * ax] = (uint32_t) v[j][ax];
                        dst[2 * ax + 1] = (uint32_t) ((uint64_t) v[j][ax] >> 32);
                    } else {
                        dst[ax] = (uint32_t) v[j][ax];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // an owner's run leaves in one piece
        for (int r = 0; r < a.nranks; ++r) {
            const int32_t b0 = s_base[r];
            const int32_t cnt = (r + 1 < a.nranks ? s_base[r + 1] : carry) - b0;
            if (cnt == 0) continue;
            const uint32_t *src = stage + (size_t) b0 * RW;
            const int64_t rec = (int64_t) s_goff[r];
            uint32_t *dst = r == a.self_rank ? reinterpret_cast<uint32_t *>(a.recv) + (rec + self_delta) * RW
                                             : reinterpret_cast<uint32_t *>(a.send) + rec * RW;
            const int len = cnt * RW;
            for (int k = lane * 4; k < len; k += 256) {
                if (k + 4 <= len) {
                    *reinterpret_cast<PpWords4 *>(dst + k) = PpWords4{src[k], src[k + 1], src[k + 2], src[k + 3]};
                } else {
                    for (int q = k; q < len; ++q) dst[q] = src[q];
                }
            }
        }
    }
}

template <class U, int D>
int partition_pack_impl(bt_context *ctx, const void *const *in, const uint32_t *cells, int64_t n,
                        const int32_t *owner_of_cell, int ncells, int nranks, int self_rank,
                        int64_t self_send_offset, int64_t self_recv_offset, const int64_t *d_self_offsets,
                        void *send, void *recv, bool wait, Buf<uint8_t> *keep_owners = nullptr,
                        Buf<int32_t> *keep_offsets = nullptr)
{
    const int64_t nwaves = div_up(n, PP_WAVE_ITEMS);
    int bits = 0;
    while ((1 << bits) < nranks) ++bits;
    Buf<int32_t> counts, offsets;
    BT_CHECK(counts.alloc(ctx->pool, (int64_t) nranks * nwaves));
    BT_CHECK(offsets.alloc(ctx->pool, (int64_t) nranks * nwaves + 1));
    Buf<uint8_t> owners;
    BT_CHECK(owners.alloc(ctx->pool, n));
    const bool small = nranks <= PP_SMALL;
    if (ncells > PP_TABLE_MAX) ncells = 0;      // (looked up in memory)
    const unsigned cblocks = (unsigned) std::min<int64_t>(div_up(nwaves, PP_WAVES), (int64_t) ctx->num_cus * 4);
    if (small)
        pp_count_kernel<true><<<cblocks, 64 * PP_WAVES, (size_t) ncells, ctx->stream>>>(
            cells, n, owner_of_cell, ncells, nranks, bits, nwaves, counts.get(), owners.get());
    else
        pp_count_kernel<false><<<cblocks, 64 * PP_WAVES, (size_t) ncells, ctx->stream>>>(
            cells, n, owner_of_cell, ncells, nranks, bits, nwaves, counts.get(), owners.get());
    BT_CHECK((device_exclusive_scan<int64_t, int32_t>(ctx, PpScan{counts.get()}, (int64_t) nranks * nwaves,
                                                      offsets.get(), (int64_t *) nullptr, true)));
    PpArgs<U, D> a{};
    for (int ax = 0; ax < D; ++ax) a.in[ax] = (const U *) in[ax];
    a.owners = owners.get(); a.offsets = offsets.get();
    a.n = n; a.nwaves = nwaves; a.nranks = nranks; a.bits = bits; a.self_rank = self_rank;
    a.self_delta = self_recv_offset - self_send_offset;
    a.self_offsets = d_self_offsets;
    a.send = (U *) send; a.recv = (U *) recv;
    constexpr int RW = D * (int) sizeof(U) / 4;
    const int nr_pad = (nranks + 3) & ~3;
    const size_t lds = (size_t) PP_WAVES * ((size_t) PP_WAVE_ITEMS * RW + (size_t) 3 * nr_pad) * 4;
    if (lds > (size_t) 160 * 1024) {
        set_error("partition: %zu bytes of LDS for %d owners and %d-byte records exceed the 160 KiB of a gfx950 CU",
                  lds, nranks, RW * 4);
        return BT_ERR_UNSUPPORTED;
    }
    const unsigned sblocks = (unsigned) div_up(nwaves, PP_WAVES);
    if (small) pp_scatter_kernel<U, D, true><<<sblocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    else pp_scatter_kernel<U, D, false><<<sblocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    BT_HIP_CHECK(hipGetLastError());
    // the send plan (one byte per particle + the scanned tile counts): bt_mgpu_route re-ranks
    // any per-particle array with it
    if (keep_owners) *keep_owners = std::move(owners);
    if (keep_offsets) *keep_offsets = std::move(offsets);
    return wait ? bt::finish_call(ctx) : BT_OK;
}

// ---- any per-particle array over the plan of an exchange ------------------------------------
//
// FORWARD: lay[position of particle i in the send layout] = in[i] (in == nullptr: iota_base + i)
// -- the own segment straight to its place in the owner-order output (self != nullptr).
// REVERSE: out[i] = lay[position of particle i] (own segment read from the owner-order input).
// The position is recomputed from (owners, offsets) by the ranking the payload sweep used; 4-
// or 8-byte elements go directly (an owner's run of a tile is contiguous: lanes of one owner
// write neighbouring addresses).
template <class U>
struct PpRouteArgs {
    const U *in;
    U *out;
    U *lay;                         // send layout [n] (FORWARD: written, REVERSE: read)
    U *self;                        // owner-order array of this rank, or nullptr (the own segment
                                    // travels through the send layout like any other)
    const uint8_t *owners;
    const int32_t *offsets;
    int64_t n, nwaves, self_delta, iota_base;
    int nranks, bits, self_rank;
};

template <class U, bool SMALL, bool REVERSE>
__global__ __launch_bounds__(64 * PP_WAVES) void pp_route_kernel(PpRouteArgs<U> a, int nr_pad)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t pp_lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int32_t *s_base = reinterpret_cast<int32_t *>(pp_lds) + (size_t) w * 3 * nr_pad;
    int32_t *s_run = s_base + nr_pad;
    int32_t *s_goff = s_run + nr_pad;
    const int64_t wave = (int64_t) blockIdx.x * PP_WAVES + w;
    if (wave >= a.nwaves) return;
    const int64_t base = wave * PP_WAVE_ITEMS + lane;
    uint32_t d[PP_ROWS];
#pragma unroll
    for (int j = 0; j < PP_ROWS; ++j) {
        const int64_t i = base + (int64_t) j * 64;
        d[j] = i < a.n ? (uint32_t) a.owners[i] : (uint32_t) a.nranks;
    }
    (void) pp_tile_tables(a.offsets, a.nwaves, wave, a.nranks, lane, s_base, s_run, s_goff);
    int32_t slot[PP_ROWS];
    pp_rank_rows<SMALL>(d, a.nranks, a.bits, lane, s_base, s_run, slot);
#pragma unroll
    for (int j = 0; j < PP_ROWS; ++j) {
        const int64_t i = base + (int64_t) j * 64;
        if (slot[j] < 0) continue;
        const int r = (int) d[j];
        const int64_t pos = (int64_t) s_goff[r] + (slot[j] - s_base[r]);
        U *where = (r == a.self_rank && a.self) ? a.self + (pos + a.self_delta) : a.lay + pos;
        if constexpr (REVERSE) a.out[i] = *where;
        else *where = a.in ? a.in[i] : (U) (a.iota_base + i);
    }
}

template <class U>
int route_impl(bt_context *ctx, bool reverse, const void *in, int64_t iota_base, void *lay, void *self, void *out,
               const uint8_t *owners, const int32_t *offsets, int64_t n, int nranks, int self_rank,
               int64_t self_delta)
{
    const int64_t nwaves = div_up(n, PP_WAVE_ITEMS);
    int bits = 0;
    while ((1 << bits) < nranks) ++bits;
    PpRouteArgs<U> a{};
    a.in = (const U *) in; a.out = (U *) out; a.lay = (U *) lay; a.self = (U *) self;
    a.owners = owners; a.offsets = offsets;
    a.n = n; a.nwaves = nwaves; a.self_delta = self_delta; a.iota_base = iota_base;
    a.nranks = nranks; a.bits = bits; a.self_rank = self_rank;
    const int nr_pad = (nranks + 3) & ~3;
    const size_t lds = (size_t) PP_WAVES * (size_t) 3 * nr_pad * 4;
    const unsigned blocks = (unsigned) div_up(nwaves, PP_WAVES);
    const bool small = nranks <= PP_SMALL;
    if (small && !reverse) pp_route_kernel<U, true, false><<<blocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    else if (small) pp_route_kernel<U, true, true><<<blocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    else if (!reverse) pp_route_kernel<U, false, false><<<blocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    else pp_route_kernel<U, false, true><<<blocks, 64 * PP_WAVES, lds, ctx->stream>>>(a, nr_pad);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

}  // namespace

namespace bt {

int route_device(bt_context *ctx, int elem_size, bool reverse, const void *in, int64_t iota_base, void *lay,
                 void *self, void *out, const uint8_t *owners, const int32_t *offsets, int64_t n, int nranks,
                 int self_rank, int64_t self_delta)
{
    if (n == 0) return BT_OK;
    return elem_size == 8
        ? route_impl<uint64_t>(ctx, reverse, in, iota_base, lay, self, out, owners, offsets, n, nranks, self_rank, self_delta)
        : route_impl<uint32_t>(ctx, reverse, in, iota_base, lay, self, out, owners, offsets, n, nranks, self_rank, self_delta);
}

int morton_cells_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords, int64_t n,
                        const void *d_rootbox, int level, uint32_t *cells_out, int32_t *hist_inout)
{
    return coord_kind == BT_F64
        ? cells_dims<double>(ctx, dims, coords, n, nullptr, nullptr, d_rootbox, level, cells_out, hist_inout, false)
        : cells_dims<float>(ctx, dims, coords, n, nullptr, nullptr, d_rootbox, level, cells_out, hist_inout, false);
}

int weight_hist_device(bt_context *ctx, const uint32_t *cells, const int32_t *weights, int64_t n,
                       int ncells, int64_t *whist)
{
    if (n == 0) return BT_OK;
    const unsigned slices = (unsigned) div_up(ncells, WH_SLICE);
    const unsigned blocks = (unsigned) std::min<int64_t>(div_up(n, 1024 * 8), ctx->num_cus);
    weight_hist_kernel<<<dim3(blocks, slices), 1024, 0, ctx->stream>>>(cells, weights, n, ncells,
                                                                     (unsigned long long *) whist);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

int widen_weights_device(bt_context *ctx, const int32_t *weights, int64_t n, int elem_size, void *out)
{
    if (n == 0) return BT_OK;
    if (elem_size == 8) widen_weights_kernel<uint64_t><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(n, weights, (uint64_t *) out);
    else widen_weights_kernel<uint32_t><<<(unsigned) div_up(n, 256), 256, 0, ctx->stream>>>(n, weights, (uint32_t *) out);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

int partition_pack_device(bt_context *ctx, int dims, int elem_size, const void *const *in,
                          const uint32_t *cells, int64_t n, const int32_t *owner_of_cell, int ncells, int nranks,
                          int self_rank, const int64_t *d_self_offsets, void *send, void *recv,
                          Buf<uint8_t> *keep_owners, Buf<int32_t> *keep_offsets)
{
    if (keep_owners) keep_owners->reset();
    if (keep_offsets) keep_offsets->reset();
    if (n == 0) return BT_OK;
    // (dims counts the values of a record: coordinates, and a radius and a refine weight that
    // travel with them)
#define BT_PP(U, D) partition_pack_impl<U, D>(ctx, in, cells, n, owner_of_cell, ncells, nranks, self_rank, 0, 0, \
                                              d_self_offsets, send, recv, false, keep_owners, keep_offsets)
    if (elem_size == 8)
        return dims == 1 ? BT_PP(uint64_t, 1) : dims == 2 ? BT_PP(uint64_t, 2) : dims == 3 ? BT_PP(uint64_t, 3)
             : dims == 4 ? BT_PP(uint64_t, 4) : BT_PP(uint64_t, 5);
    return dims == 1 ? BT_PP(uint32_t, 1) : dims == 2 ? BT_PP(uint32_t, 2) : dims == 3 ? BT_PP(uint32_t, 3)
         : dims == 4 ? BT_PP(uint32_t, 4) : BT_PP(uint32_t, 5);
#undef BT_PP
}

}  // namespace bt

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
    return coord_kind == BT_F64
        ? cells_dims<double>(ctx, dims, coords, n, bbox_min, bbox_max, nullptr, level, cells_out, hist_inout, true)
        : cells_dims<float>(ctx, dims, coords, n, bbox_min, bbox_max, nullptr, level, cells_out, hist_inout, true);
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
#define BT_PP(U, D) partition_pack_impl<U, D>(ctx, in, cells, n, owner_of_cell, 0, nranks, self_rank, \
                                              self_send_offset, self_recv_offset, nullptr, send, recv, true)
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
    // (dims counts the values of a record: up to BT_MAX_DIMS coordinates and a radius)
    if (!ctx || n < 0 || dims < 1 || dims > BT_MAX_DIMS + 1 || (elem_size != 4 && elem_size != 8)
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
          : dims == 3 ? unpack_impl<uint64_t, 3>(ctx, in, n, out)
                      : unpack_impl<uint64_t, 4>(ctx, in, n, out);
    } else {
        s = dims == 1 ? unpack_impl<uint32_t, 1>(ctx, in, n, out)
          : dims == 2 ? unpack_impl<uint32_t, 2>(ctx, in, n, out)
          : dims == 3 ? unpack_impl<uint32_t, 3>(ctx, in, n, out)
                      : unpack_impl<uint32_t, 4>(ctx, in, n, out);
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

// level starts of a level-major box set, for kernels that handle every level in one launch
struct LevelStarts {
    int32_t nlevels;
    int32_t start[BT_MAX_LEVELS + 2];
};

__device__ __forceinline__ int level_of_box(const LevelStarts &ls, int32_t b)
{
    int lo = 0, hi = ls.nlevels - 1;          // largest l with start[l] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ls.start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// One thread per box, all levels in one launch: the parent by path lookup in the level above
// (the box then enters itself into its parent's child row: rows are cleared beforehand, a slot
// nobody claims stays 0), and the centre by the builder's own chain of roundings from the root
// (child centre = parent centre +/- root_extent / 2^(1+level), tree_build_kernels.py:698-705),
// walked down the digits of the path -- the values a level-by-level sweep produces.  A box
// without parent, or a level that is not ascending by path, raises the device status (code 70).
template <class T, int D>
struct LetLinkArgs {
    LevelStarts ls;
    int64_t aligned;
    const uint64_t *paths;
    int32_t *parent_ids, *child_ids;
    T *centers;
    T root_center[D];
    T root_extent;
    DeviceStatus *status;
};

template <class T, int D>
__global__ __launch_bounds__(256) void let_link_kernel(LetLinkArgs<T, D> a)
{
    constexpr int C = 1 << D;
    const int32_t nboxes = a.ls.start[a.ls.nlevels];
    const int32_t b = (int32_t) (blockIdx.x * 256 + threadIdx.x);
    if (b >= nboxes) return;
    const int lev = level_of_box(a.ls, b);
    const uint64_t p = a.paths[b];
    if (lev == 0) {
        a.parent_ids[b] = 0;
    } else {
        const int64_t par = find_path(a.paths, a.ls.start[lev - 1], a.ls.start[lev], p >> D);
        a.parent_ids[b] = par < 0 ? 0 : (int32_t) par;
        if (par < 0) atomicExch(&a.status->internal, 70);
        else a.child_ids[(int64_t) (p & (uint64_t) (C - 1)) * a.aligned + par] = b;
        if (b > a.ls.start[lev] && !(a.paths[b - 1] < p)) atomicExch(&a.status->internal, 70);
    }
    T c[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) c[ax] = a.root_center[ax];
    for (int l = 1; l <= lev; ++l) {
        const int m = (int) ((p >> (D * (lev - l))) & (uint64_t) (C - 1));
        const T radius = (a.root_extent * 1 / (T) (1ull << (1 + l)));
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            const bool has_bit = (m >> (D - 1 - ax)) & 1;
            c[ax] = has_bit ? c[ax] + radius : c[ax] - radius;
        }
    }
#pragma unroll
    for (int ax = 0; ax < D; ++ax) a.centers[(int64_t) ax * a.aligned + b] = c[ax];
}

template <class T, int D>
int let_build_impl(bt_context *ctx, int nlevels, const int32_t *level_starts, const uint64_t *paths,
                   int64_t aligned, const double *bbox_min, const double *bbox_max,
                   double root_extent, int32_t *parent_ids, int32_t *child_ids, T *centers)
{
    constexpr int C = 1 << D;
    const int32_t nboxes = level_starts[nlevels];
    BT_HIP_CHECK(hipMemsetAsync(child_ids, 0, (size_t) C * (size_t) aligned * 4, ctx->stream));
    if (nboxes <= 0) return BT_OK;
    LetLinkArgs<T, D> a{};
    a.ls.nlevels = nlevels;
    for (int l = 0; l <= nlevels; ++l) a.ls.start[l] = level_starts[l];
    a.aligned = aligned; a.paths = paths; a.parent_ids = parent_ids; a.child_ids = child_ids;
    a.centers = centers;
    // root centre: tree_build.py:585-590
    for (int ax = 0; ax < D; ++ax) {
        const T mn = (T) bbox_min[ax], mx = (T) bbox_max[ax];
        a.root_center[ax] = mn + (mx - mn) / 2;
    }
    a.root_extent = (T) root_extent;
    a.status = ctx->d_status;
    let_link_kernel<T, D><<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(a);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

template <class T, int D>
int box_paths_impl(bt_context *ctx, int64_t nboxes, int64_t aligned_nboxes, const void *box_centers,
                   const uint8_t *box_levels, const double *bbox_min, double root_extent, uint64_t *paths)
{
    const double m0 = bbox_min[0], m1 = D > 1 ? bbox_min[1] : 0, m2 = D > 2 ? bbox_min[2] : 0;
    box_paths_kernel<T, D><<<(unsigned) div_up(nboxes, 256), 256, 0, ctx->stream>>>(
        nboxes, aligned_nboxes, (const T *) box_centers, box_levels, m0, m1, m2, root_extent, paths);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

}  // namespace

namespace bt {

int box_paths_device(bt_context *ctx, int dims, int coord_kind, int64_t nboxes, int64_t aligned_nboxes,
                     const void *box_centers, const uint8_t *box_levels, const double *bbox_min,
                     double root_extent, uint64_t *paths)
{
    if (nboxes == 0) return BT_OK;
#define BP(T, D) return box_paths_impl<T, D>(ctx, nboxes, aligned_nboxes, box_centers, box_levels, bbox_min, root_extent, paths)
    if (coord_kind == BT_F64) { if (dims == 1) BP(double, 1); else if (dims == 2) BP(double, 2); else BP(double, 3); }
    else { if (dims == 1) BP(float, 1); else if (dims == 2) BP(float, 2); else BP(float, 3); }
#undef BP
}

int let_link_device(bt_context *ctx, int dims, int coord_kind, int nlevels, const int32_t *level_start_box_nrs,
                    const uint64_t *paths, int64_t aligned_nboxes, const double *bbox_min,
                    const double *bbox_max, double root_extent, int32_t *box_parent_ids,
                    int32_t *box_child_ids, void *box_centers)
{
#define LB(T, D) return let_build_impl<T, D>(ctx, nlevels, level_start_box_nrs, paths,            \
        aligned_nboxes, bbox_min, bbox_max, root_extent, box_parent_ids, box_child_ids,           \
        (T *) box_centers)
    if (coord_kind == BT_F64) { if (dims == 1) LB(double, 1); else if (dims == 2) LB(double, 2); else LB(double, 3); }
    else { if (dims == 1) LB(float, 1); else if (dims == 2) LB(float, 2); else LB(float, 3); }
#undef LB
}

}  // namespace bt

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
    BT_CHECK(bt::box_paths_device(ctx, dims, coord_kind, nboxes, aligned_nboxes, box_centers, box_levels,
                                  bbox_min, root_extent, paths));
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
    BT_CHECK(reset_status(ctx));
    BT_CHECK(bt::let_link_device(ctx, dims, coord_kind, nlevels, level_start_box_nrs, paths, aligned_nboxes,
                                 bbox_min, bbox_max, root_extent, box_parent_ids, box_child_ids, box_centers));
    // a box without parent among the boxes of the level above: device status code 70
    return bt::finish_call(ctx);
}

}  // extern "C"
