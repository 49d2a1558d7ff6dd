// Tree build for gfx950: bounding box, Morton keys, sort-first box construction,
// within-box order fix-up, source/target split, coordinate gather, flags and
// box extents.  Produces arrays bit-identical to the reference's level loop
// (boxtree/tree_build.py:145-1878, boxtree/tree_build_kernels.py) -- see
// DESIGN.md for the equivalence argument.
//
// Sort-first formulation.  Per particle one 64-bit composite key
//     K = (Kt << CAPBITS) | cap
// Kt  = Morton path of the particle (d bits per level, level 1 most significant,
//       x most significant inside a digit: tbk:441-445), truncated (zeroed)
//       below level `cap`,
// cap = deepest level the particle may descend to: one less than the first
//       level whose box it sticks out of (tbk:388-428); = L without extents.
// A stable LSD radix sort by K starting from user order places every box's
// particles contiguously, a box's own ("non-child") particles before its
// children (tbk:163), children in Morton order.  Boxes are then carved out of
// the sorted key array level by level with binary searches (work ~ #boxes, not
// #particles x #levels as in the reference).
#include "bt_common.hpp"
#include "bt_prims.hpp"
#include "bt_sort.hpp"
#include "bt_geom.hpp"

#include <algorithm>
#include <functional>
#include <climits>
#include <cstdlib>
#include <cmath>

using namespace bt;

namespace {

constexpr int CAPBITS_EXT = 5;

template <class T> struct CoordTraits;
template <> struct CoordTraits<float> { static constexpr float maxval = 3.402823466e+38f; };
template <> struct CoordTraits<double> { static constexpr double maxval = 1.7976931348623158e+308; };

// ---------------------------------------------------------------------------
// bounding box (bounding_box.py:54-122)
// ---------------------------------------------------------------------------

constexpr int BBOX_THREADS = 256;

template <class T> struct BboxAxes { const T *x[3]; };

template <class T>
__device__ __forceinline__ void bbox_block(const T *__restrict__ x, const T *__restrict__ radii,
                                           int64_t n, T *partial /* [2*gridDim.x] */);

template <class T>
__global__ __launch_bounds__(BBOX_THREADS) void bbox_kernel(const T *__restrict__ x,
        const T *__restrict__ radii, int64_t n, T *partial /* [2*gridDim.x] */)
{
    bbox_block<T>(x, radii, n, partial);
}

// all axes in one launch: blockIdx.y = axis, partial[axis][2*gridDim.x]
template <class T>
__global__ __launch_bounds__(BBOX_THREADS) void bbox_axes_kernel(BboxAxes<T> ax, int64_t n, T *partial)
{
    bbox_block<T>(ax.x[blockIdx.y], (const T *) nullptr, n, partial + (int64_t) 2 * gridDim.x * blockIdx.y);
}

template <class T>
__device__ __forceinline__ void bbox_block(const T *__restrict__ x, const T *__restrict__ radii,
                                           int64_t n, T *partial /* [2*gridDim.x] */)
{
    __shared__ T s_mn[BBOX_THREADS / 64], s_mx[BBOX_THREADS / 64];
    T mn = CoordTraits<T>::maxval, mx = -CoordTraits<T>::maxval;
    const int64_t stride = (int64_t) gridDim.x * BBOX_THREADS;
    for (int64_t i = (int64_t) blockIdx.x * BBOX_THREADS + threadIdx.x; i < n; i += stride) {
        const T r = radii ? radii[i] : (T) 0;
        const T c = x[i];
        const T lo = c - r, hi = c + r;
        mn = (lo < mn) ? lo : mn;
        mx = (hi > mx) ? hi : mx;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T omn = __shfl_xor(mn, off, 64), omx = __shfl_xor(mx, off, 64);
        mn = (omn < mn) ? omn : mn;
        mx = (omx > mx) ? omx : mx;
    }
    if (lane_id() == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BBOX_THREADS / 64; ++w) {
            mn = (s_mn[w] < mn) ? s_mn[w] : mn;
            mx = (s_mx[w] > mx) ? s_mx[w] : mx;
        }
        partial[2 * blockIdx.x] = mn;
        partial[2 * blockIdx.x + 1] = mx;
    }
}

// Root box on the device (tree_build.py:456-476 in the coordinate type): partial[k][ax]
// are the per-workgroup (min, max) pairs of bbox_kernel for sources (k = 0) and targets
// (k = 1); out = {min[3], max[3], root_extent}.
template <class T, int D>
__global__ __launch_bounds__(256) void root_box_kernel(const T *partial, int nblk_src, int nblk_tgt,
        T one_plus_stretch, T *out)
{
    __shared__ T s_mn[256], s_mx[256];
    T lo[D], hi[D];
    for (int ax = 0; ax < D; ++ax) {
        T mn = CoordTraits<T>::maxval, mx = -CoordTraits<T>::maxval;
        const T *ps = partial + (int64_t) 2 * nblk_src * ax;
        for (int b = threadIdx.x; b < nblk_src; b += 256) {
            mn = (ps[2 * b] < mn) ? ps[2 * b] : mn;
            mx = (ps[2 * b + 1] > mx) ? ps[2 * b + 1] : mx;
        }
        const T *pt = partial + (int64_t) 2 * nblk_src * D + (int64_t) 2 * nblk_tgt * ax;
        for (int b = threadIdx.x; b < nblk_tgt; b += 256) {
            mn = (pt[2 * b] < mn) ? pt[2 * b] : mn;
            mx = (pt[2 * b + 1] > mx) ? pt[2 * b + 1] : mx;
        }
        s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int) threadIdx.x < off) {
                const T a = s_mn[threadIdx.x + off], c = s_mx[threadIdx.x + off];
                if (a < s_mn[threadIdx.x]) s_mn[threadIdx.x] = a;
                if (c > s_mx[threadIdx.x]) s_mx[threadIdx.x] = c;
            }
            __syncthreads();
        }
        lo[ax] = s_mn[0]; hi[ax] = s_mx[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        T widest = hi[0] - lo[0];
        for (int ax = 1; ax < D; ++ax) { const T w = hi[ax] - lo[ax]; widest = (w > widest) ? w : widest; }
        const T extent = widest * one_plus_stretch;
        for (int ax = 0; ax < 3; ++ax) {
            out[ax] = ax < D ? lo[ax] : (T) 0;
            out[3 + ax] = ax < D ? lo[ax] + extent : (T) 0;
        }
        out[6] = extent;
        out[7] = (T) 0;
    }
}

template <class T>
int bbox_impl(bt_context *ctx, int dims, const void *const *coords, const void *radii,
              int64_t n, double *out_min, double *out_max)
{
    for (int d = 0; d < dims; ++d) {
        out_min[d] = (double) CoordTraits<T>::maxval;
        out_max[d] = -(double) CoordTraits<T>::maxval;
    }
    if (n == 0) return BT_OK;
    int64_t blocks = std::min<int64_t>(div_up(n, BBOX_THREADS * 8), (int64_t) ctx->num_cus * 8);
    Buf<T> partial;
    BT_CHECK(partial.alloc(ctx->pool, 2 * blocks * dims));
    for (int d = 0; d < dims; ++d)
        bbox_kernel<T><<<(unsigned) blocks, BBOX_THREADS, 0, ctx->stream>>>(
            (const T *) coords[d], (const T *) radii, n, partial.get() + 2 * blocks * d);
    BT_HIP_CHECK(hipGetLastError());
    std::vector<T> h((size_t) (2 * blocks * dims));
    BT_CHECK(bt::d2h(ctx, h.data(), partial.get(), h.size() * sizeof(T)));
    BT_CHECK(bt::sync_stream(ctx));
    for (int d = 0; d < dims; ++d) {
        T mn = CoordTraits<T>::maxval, mx = -CoordTraits<T>::maxval;
        for (int64_t b = 0; b < blocks; ++b) {
            T a = h[(size_t) (2 * blocks * d + 2 * b)], c = h[(size_t) (2 * blocks * d + 2 * b + 1)];
            mn = (a < mn) ? a : mn;
            mx = (c > mx) ? c : mx;
        }
        out_min[d] = (double) mn;
        out_max[d] = (double) mx;
    }
    return BT_OK;
}

// (min, -max) per axis of the per-workgroup pairs, folded into mm[0..D) / mm[D..2D) with MIN:
// the form an all-reduce(MIN) over ranks takes (bt_mgpu.hip)
template <class T>
__global__ __launch_bounds__(256) void bbox_fold_kernel(const T *partial, int nblk, int D, double *mm)
{
    __shared__ T s_mn[256], s_mx[256];
    for (int ax = 0; ax < D; ++ax) {
        T mn = CoordTraits<T>::maxval, mx = -CoordTraits<T>::maxval;
        const T *ps = partial + (int64_t) 2 * nblk * ax;
        for (int b = threadIdx.x; b < nblk; b += 256) {
            mn = (ps[2 * b] < mn) ? ps[2 * b] : mn;
            mx = (ps[2 * b + 1] > mx) ? ps[2 * b + 1] : mx;
        }
        s_mn[threadIdx.x] = mn; s_mx[threadIdx.x] = mx;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int) threadIdx.x < off) {
                const T a = s_mn[threadIdx.x + off], c = s_mx[threadIdx.x + off];
                if (a < s_mn[threadIdx.x]) s_mn[threadIdx.x] = a;
                if (c > s_mx[threadIdx.x]) s_mx[threadIdx.x] = c;
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const double lo = (double) s_mn[0], nhi = -(double) s_mx[0];
            if (lo < mm[ax]) mm[ax] = lo;
            if (nhi < mm[D + ax]) mm[D + ax] = nhi;
        }
        __syncthreads();
    }
}

template <class T>
int bbox_fold_impl(bt_context *ctx, int dims, const void *const *coords, const void *radii, int64_t n,
                   double *d_mm)
{
    if (n == 0) return BT_OK;
    const int64_t blocks = std::min<int64_t>(div_up(n, BBOX_THREADS * 8), (int64_t) ctx->num_cus * 8);
    Buf<T> partial;
    BT_CHECK(partial.alloc(ctx->pool, 2 * blocks * dims));
    if (radii) {
        // bounding_box.py:54-122 with radii: min(x - r), max(x + r)
        for (int ax = 0; ax < dims; ++ax)
            bbox_kernel<T><<<(unsigned) blocks, BBOX_THREADS, 0, ctx->stream>>>(
                (const T *) coords[ax], (const T *) radii, n, partial.get() + 2 * blocks * ax);
    } else {
        BboxAxes<T> axs{};
        for (int ax = 0; ax < dims; ++ax) axs.x[ax] = (const T *) coords[ax];
        bbox_axes_kernel<T><<<dim3((unsigned) blocks, dims), BBOX_THREADS, 0, ctx->stream>>>(axs, n, partial.get());
    }
    bbox_fold_kernel<T><<<1, 256, 0, ctx->stream>>>(partial.get(), (int) blocks, dims, d_mm);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

// ---------------------------------------------------------------------------
// Morton keys (tbk:308-470)
// ---------------------------------------------------------------------------

template <int D> __device__ __forceinline__ uint64_t spread_bits(uint32_t v);
template <> __device__ __forceinline__ uint64_t spread_bits<1>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint64_t spread_bits<2>(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
template <> __device__ __forceinline__ uint64_t spread_bits<3>(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

template <class T, int D>
struct KeygenArgs {
    const T *src[D];
    const T *tgt[D];
    const T *src_radii;
    const T *tgt_radii;
    int64_t nsources, n;
    int64_t src_stride, tgt_stride;     // elements between consecutive points (1 = dense)
    T bbox_min[D], bbox_max[D];
    const T *rootbox;        // device {min[3], max[3], extent}: overrides bbox_min/max (or null)
    T stick_out_factor;
    int L;          // levels in the key
    int norm;       // BT_NORM_*
    int point_skip_levels;   // levels 1..this cannot stop a particle of radius 0
    // packed keys (point particles): key = (path >> pack_drop) << pack_idbits | srcntgt id
    int pack_idbits, pack_drop;
};

// Records of the interleaved coordinate copy: 3-D points are padded to four
// coordinates so that a record never straddles a 32-byte DRAM sector (f64) -- the
// tree-order gather is bound by the number of sectors it touches.
template <int D>
struct PackStride { static constexpr int value = D == 3 ? 4 : D; };

// The sort key of one particle: (Kt << capbits) | cap with extents, Kt without (see the head of
// DESIGN.md section 3).  x: coordinates, radius: its extent radius (EXT only).
template <class T, int D, bool EXT>
__device__ __forceinline__ uint64_t particle_key(const KeygenArgs<T, D> &a, const T (&x)[D], T radius)
{
    T gmin[D], gext[D];
    uint32_t v[D];
    const int L = a.L;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        gmin[ax] = a.rootbox ? a.rootbox[ax] : a.bbox_min[ax];                    // tbk:358
        gext[ax] = (a.rootbox ? a.rootbox[3 + ax] : a.bbox_max[ax]) - gmin[ax];   // tbk:359
        // tbk:374-376 evaluated at the deepest level; scaling by 2^k is exact,
        // so (v >> (L-l)) is the reference's level-l value.
        v[ax] = (uint32_t) (((x[ax] - gmin[ax]) / gext[ax]) * (T) (1u << L));
    }
    int cap = L;
    if (EXT) {
        const T one_half = ((T) 1) / 2;
        const T brf = (T) ((1. + (double) a.stick_out_factor) * (double) one_half);   // tbk:342-346
        // A point (radius 0) lies inside every box of its own path, at least
        // stick_out_factor/2 box sizes away from the stick-out limit.  As long as that
        // margin exceeds the rounding of the expressions below by a wide factor (the
        // host works out down to which level it does, from the bounding box, the
        // factor and the precision of T) the test cannot fire, and the walk starts
        // below those levels: all of them in double precision, the first ~14 in
        // single.  Deeper, where a box is smaller than the spacing of T around the
        // point, upstream's test does fire for points, and so does this one.
        const int lfirst = (radius == (T) 0) ? a.point_skip_levels + 1 : 1;
        for (int l = lfirst; l <= L; ++l) {
            const T size_factor = ((T) 1) / ((T) (1u << l));    // tbk:328-329
            bool stop = false;
            T center[D];
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const uint32_t bits = v[ax] >> (L - l);
                center[ax] = gmin[ax] + gext[ax] * ((T) bits + one_half) * size_factor;  // tbk:380-384
            }
            if (a.norm == BT_NORM_LINF) {
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const T sor = brf * gext[ax] * size_factor;                // tbk:390-393
                    stop = stop || (x[ax] + radius >= center[ax] + sor);       // tbk:396-399
                    stop = stop || (x[ax] - radius < center[ax] - sor);        // tbk:400-403
                }
            } else {
                const T sor = brf * gext[0] * size_factor;                     // tbk:408-411
                T sumsq = (T) 0;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const T t = (x[ax] - center[ax]) * (x[ax] - center[ax]);
                    sumsq = (ax == 0) ? t : sumsq + t;
                }
                const T dist = sqrt(sumsq) + radius;                           // tbk:413-419
                stop = stop || (dist * dist >= D * sor * sor);                 // tbk:422-428
            }
            if (stop) { cap = l - 1; break; }
        }
    }

    uint64_t kt = 0;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const uint32_t m = (L >= 32) ? v[ax] : (v[ax] & ((1u << L) - 1u));
        kt |= spread_bits<D>(m) << (D - 1 - ax);                              // tbk:441-445
    }
    if (EXT) {
        const int drop = D * (L - cap);
        if (drop > 0) kt = (drop >= 64) ? 0 : (kt >> drop) << drop;
        return (kt << CAPBITS_EXT) | (uint64_t) cap;
    }
    return kt;
}

// packed word of a key: the path bits of the first pack_levels levels, the cap (extents), the id
template <int D, bool EXT>
__device__ __forceinline__ uint64_t pack_key(uint64_t key, int pack_drop, int pack_idbits, uint64_t id)
{
    constexpr int CB = EXT ? CAPBITS_EXT : 0;
    const uint64_t top = ((key >> CB) >> pack_drop) << CB | (key & (((uint64_t) 1 << CB) - 1));
    return (top << pack_idbits) | id;
}

template <class T, int D, bool EXT>
__global__ __launch_bounds__(256) void keygen_kernel(KeygenArgs<T, D> a, uint64_t *__restrict__ keys,
                                                     T *__restrict__ packed /* [n][PackStride<D>] */)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const bool is_src = i < a.nsources;
    const int64_t j = is_src ? i : i - a.nsources;

    T x[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax)
        x[ax] = is_src ? a.src[ax][j * a.src_stride] : a.tgt[ax][j * a.tgt_stride];
    // interleaved copy: the tree-order gather later needs ONE random access per
    // particle instead of one per axis
    {
        constexpr int PS = PackStride<D>::value;
        T rec[PS];
#pragma unroll
        for (int ax = 0; ax < PS; ++ax) rec[ax] = ax < D ? x[ax] : (T) 0;
#pragma unroll
        for (int ax = 0; ax < PS; ++ax) packed[i * PS + ax] = rec[ax];
    }
    T radius = (T) 0;
    if (EXT) {
        if (is_src) { if (a.src_radii) radius = a.src_radii[j]; }
        else        { if (a.tgt_radii) radius = a.tgt_radii[j]; }
    }
    const uint64_t key = particle_key<T, D, EXT>(a, x, radius);
    keys[i] = a.pack_idbits > 0 ? pack_key<D, EXT>(key, a.pack_drop, a.pack_idbits, (uint64_t) i) : key;
}

// ---------------------------------------------------------------------------
// Packed keys.  A build of N < 2^idbits point particles (no extents, unit weights,
// kind "adaptive") sorts ONE word per particle: the Morton path of the levels the tree
// is expected to reach (Lk of them) shifted over the particle's srcntgt id.  A stable
// LSD sort on the path bits alone (bt::radix_sort_keys: 16 bytes per particle and pass
// instead of 24) leaves the ids of equal paths ascending -- the pair sort's order.  The
// level kernels read the array as a key of Lk levels with idbits "cap" bits below the
// path; the fix-up unpacks the ids while it orders the leaves.  A tree deeper than Lk
// levels goes back to full keys (rekey_full_kernel + the pair sort) at that point.
// ---------------------------------------------------------------------------

// ids (and, with keys_out, the full keys) of the particles in their current order
template <class T, int D, bool EXT>
__global__ __launch_bounds__(256) void rekey_full_kernel(int64_t n, const uint64_t *pk, uint64_t id_mask,
        KeygenArgs<T, D> a, const T *__restrict__ packed, uint32_t *ids_out, uint64_t *keys_out)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t id = (uint32_t) (pk[p] & id_mask);
    ids_out[p] = id;
    if (!keys_out) return;
    constexpr int PS = PackStride<D>::value;
    T x[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) x[ax] = packed[(int64_t) id * PS + ax];
    T radius = (T) 0;
    if (EXT) {
        if ((int64_t) id < a.nsources) { if (a.src_radii) radius = a.src_radii[id]; }
        else                           { if (a.tgt_radii) radius = a.tgt_radii[id - a.nsources]; }
    }
    keys_out[p] = particle_key<T, D, EXT>(a, x, radius);
}

// ---------------------------------------------------------------------------
// Depth probe: cells of level k of (a sample of) the particles.  The densest cell and
// the ratio of occupied cells between levels k and k-1 (the dimension of the set the
// points lie on, at that scale) give the depth estimate that decides how many key bits
// are sorted up front -- N alone has to assume a surface.
// ---------------------------------------------------------------------------

template <class T, int D>
__global__ __launch_bounds__(256) void depth_probe_kernel(KeygenArgs<T, D> a, int64_t nsample, int64_t stride,
                                                          int k, uint32_t *hist)
{
    const int64_t sidx = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (sidx >= nsample) return;
    const int64_t i = sidx * stride;
    const bool is_src = i < a.nsources;
    const int64_t j = is_src ? i : i - a.nsources;
    uint64_t cell = 0;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const T x = is_src ? a.src[ax][j * a.src_stride] : a.tgt[ax][j * a.tgt_stride];
        const T gmin = a.rootbox ? a.rootbox[ax] : a.bbox_min[ax];
        const T gext = (a.rootbox ? a.rootbox[3 + ax] : a.bbox_max[ax]) - gmin;
        uint32_t v = (uint32_t) (((x - gmin) / gext) * (T) (1u << k));
        v = v < (1u << k) ? v : (1u << k) - 1u;
        cell |= spread_bits<D>(v) << (D - 1 - ax);
    }
    atomicAdd(&hist[cell], 1u);
}

__global__ __launch_bounds__(256) void prefix_to_hist_kernel(const int64_t *prefix, int64_t ncells, uint32_t *hist)
{
    const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (c >= ncells) return;
    const int64_t w = prefix[c + 1] - prefix[c];
    hist[c] = w > 0x7fffffff ? 0x7fffffffu : (uint32_t) w;
}

// out[0] = largest cell count, out[1] = occupied cells, out[2] = occupied parent cells
__global__ __launch_bounds__(256) void depth_probe_reduce_kernel(const uint32_t *hist, int64_t ncells, int C,
                                                                 uint32_t *out)
{
    uint32_t cmax = 0, occ = 0, occp = 0;
    for (int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x; g * C < ncells; g += (int64_t) gridDim.x * 256) {
        bool any = false;
        for (int c = 0; c < C; ++c) {
            const uint32_t h = hist[g * C + c];
            cmax = h > cmax ? h : cmax;
            occ += h ? 1u : 0u;
            any = any || h;
        }
        occp += any ? 1u : 0u;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(cmax, off, 64);
        cmax = o > cmax ? o : cmax;
        occ += __shfl_xor(occ, off, 64);
        occp += __shfl_xor(occp, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&out[0], cmax);
        atomicAdd(&out[1], occ);
        atomicAdd(&out[2], occp);
    }
}

__global__ __launch_bounds__(256) void unpack_ids_kernel(int64_t n, const uint64_t *pk, uint64_t id_mask,
                                                         uint32_t *ids)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (p < n) ids[p] = (uint32_t) (pk[p] & id_mask);
}

// ---------------------------------------------------------------------------
// Continuation keys.  The 64-bit key addresses L1 levels (21 in 3D); the reference
// is defined down to level 31 (`1U << (1 + level)`, tbk:329, 374-376).  Boxes of
// level L1 that still have to split get their particles re-keyed for the levels
// L1+1 .. 31 and re-sorted inside the box:
//     K2 = (segment << (D*L2 + capbits)) | (Kt2 << capbits) | cap2,
// segment = rank of the box among the re-keyed ones (keeps boxes apart in one
// global stable sort), Kt2 = Morton path below level L1, cap2 = (deepest level the
// particle may descend to) - L1.  The per-axis cell index at level 31 is the same
// expression as keygen_kernel's, evaluated with 2^31.
// ---------------------------------------------------------------------------

constexpr int KEY_AXIS_BITS = 31;

template <class T, int D>
struct Keygen2Args {
    const T *packed;            // [n][PackStride<D>] interleaved coordinates (srcntgt order)
    const T *src_radii, *tgt_radii;
    int64_t nsources;
    T bbox_min[D], bbox_max[D];
    T stick_out_factor;
    int L1, L2, capbits, norm, point_skip_levels;
    const uint32_t *ids;        // tree order -> srcntgt id
    const int32_t *box_start;
    const int32_t *seg_box;     // [nseg] re-keyed boxes, ascending
    const int32_t *seg_start;   // [nseg + 1] offsets into the compact arrays
    int32_t nseg;
    int64_t m;                  // particles to re-key
};

template <class T, int D, bool EXT>
__global__ __launch_bounds__(256) void keygen2_kernel(Keygen2Args<T, D> a, uint64_t *keys2,
                                                      uint32_t *ids2, int32_t *positions)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= a.m) return;
    int lo = 0, hi = a.nseg;                    // last segment with seg_start <= j
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t) a.seg_start[mid] <= j) lo = mid; else hi = mid;
    }
    const int seg = lo;
    const int32_t p = a.box_start[a.seg_box[seg]] + (int32_t) (j - a.seg_start[seg]);
    const uint32_t id = a.ids[p];
    constexpr int PS = PackStride<D>::value;
    T x[D], gmin[D], gext[D];
    uint32_t v[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        x[ax] = a.packed[(int64_t) id * PS + ax];
        gmin[ax] = a.bbox_min[ax];
        gext[ax] = a.bbox_max[ax] - gmin[ax];
        v[ax] = (uint32_t) (((x[ax] - gmin[ax]) / gext[ax]) * (T) (1u << KEY_AXIS_BITS));
    }
    const int Ltot = a.L1 + a.L2;               // == KEY_AXIS_BITS
    int cap = Ltot;
    if (EXT) {
        T radius = (T) 0;
        if ((int64_t) id < a.nsources) { if (a.src_radii) radius = a.src_radii[id]; }
        else                           { if (a.tgt_radii) radius = a.tgt_radii[id - a.nsources]; }
        const T one_half = ((T) 1) / 2;
        const T brf = (T) ((1. + (double) a.stick_out_factor) * (double) one_half);   // tbk:342-346
        int lfirst = a.L1 + 1;
        if (radius == (T) 0 && a.point_skip_levels + 1 > lfirst) lfirst = a.point_skip_levels + 1;
        for (int l = lfirst; l <= Ltot; ++l) {
            const T size_factor = ((T) 1) / ((T) (1u << l));    // tbk:328-329
            bool stop = false;
            T center[D];
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const uint32_t bits = v[ax] >> (Ltot - l);
                center[ax] = gmin[ax] + gext[ax] * ((T) bits + one_half) * size_factor;  // tbk:380-384
            }
            if (a.norm == BT_NORM_LINF) {
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const T sor = brf * gext[ax] * size_factor;                // tbk:390-393
                    stop = stop || (x[ax] + radius >= center[ax] + sor);       // tbk:396-399
                    stop = stop || (x[ax] - radius < center[ax] - sor);        // tbk:400-403
                }
            } else {
                const T sor = brf * gext[0] * size_factor;                     // tbk:408-411
                T sumsq = (T) 0;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const T t = (x[ax] - center[ax]) * (x[ax] - center[ax]);
                    sumsq = (ax == 0) ? t : sumsq + t;
                }
                const T dist = sqrt(sumsq) + radius;                           // tbk:413-419
                stop = stop || (dist * dist >= D * sor * sor);                 // tbk:422-428
            }
            if (stop) { cap = l - 1; break; }
        }
    }
    uint64_t kt = 0;
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const uint32_t mlow = v[ax] & ((1u << a.L2) - 1u);
        kt |= spread_bits<D>(mlow) << (D - 1 - ax);                               // tbk:441-445
    }
    const int cap2 = cap - a.L1;
    if (EXT) {
        const int drop = D * (a.L2 - cap2);
        if (drop > 0) kt = (drop >= 64) ? 0 : (kt >> drop) << drop;
    }
    const int keybits2 = D * a.L2 + a.capbits;
    keys2[j] = ((uint64_t) seg << keybits2) | (kt << a.capbits) | (EXT ? (uint64_t) cap2 : 0ull);
    ids2[j] = id;
    positions[j] = p;
}

__global__ __launch_bounds__(256) void scatter_rekeyed_kernel(int64_t m, const int32_t *positions,
        const uint64_t *keys2, const uint32_t *ids2, uint64_t *keys_full, uint32_t *ids)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const int32_t p = positions[j];
    keys_full[p] = keys2[j];
    ids[p] = ids2[j];
}

// boxes of one level that may have to split below the key's reach
struct RekeyPred {
    const int32_t *box_count;
    const int64_t *wprefix;
    const int32_t *box_start;
    int32_t b0, max_weight;
    int adaptive;
    // level-restricted builds (boxes in creation order, any leaf may be split later): every
    // non-empty box of level `want_level`
    const uint8_t *levels;
    int want_level;
    __device__ bool cand(int64_t i) const
    {
        const int32_t b = b0 + (int32_t) i;
        const int32_t n = box_count[b];
        if (n <= 0) return false;
        if (levels) return (int) levels[b] == want_level;
        if (!adaptive) return true;
        int64_t w = n;
        if (wprefix) { const int32_t s = box_start[b]; w = wprefix[s + n] - wprefix[s]; }
        return w > (int64_t) max_weight;
    }
};
struct RekeyFlag {
    RekeyPred p;
    __device__ int32_t operator()(int64_t i) const { return p.cand(i) ? 1 : 0; }
};
struct RekeyCount {
    RekeyPred p;
    __device__ int32_t operator()(int64_t i) const { return p.cand(i) ? p.box_count[p.b0 + (int32_t) i] : 0; }
};

__global__ __launch_bounds__(256) void rekey_segments_kernel(int32_t nb, RekeyPred pr,
        const int32_t *seg_rank, const int32_t *seg_off, uint8_t *cand, int32_t *seg_box,
        int32_t *seg_start)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i > nb) return;
    if (i == nb) { seg_start[seg_rank[nb]] = seg_off[nb]; return; }
    const bool c = pr.cand(i);
    cand[i] = c ? 1 : 0;
    if (c) { seg_box[seg_rank[i]] = pr.b0 + i; seg_start[seg_rank[i]] = seg_off[i]; }
}

// ---------------------------------------------------------------------------
// box construction from the sorted keys
// ---------------------------------------------------------------------------

__device__ __forceinline__ int lower_bound_key(const uint64_t *k, int lo, int hi, uint64_t v)
{
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (k[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int upper_bound_key(const uint64_t *k, int lo, int hi, uint64_t v)
{
    while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (k[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct LevelFlags {
    int32_t total_new;      // written by the scan
    int32_t have_oversize;  // tbk:600-610
    int32_t need_more;      // a box at the key's deepest level must split: continuation
    int32_t pad;
};

struct BuildArgs {
    const uint64_t *keys;
    const int64_t *wprefix;     // [N+1] or null (unit weights)
    int32_t *box_start, *box_count, *box_parent, *box_nonchild, *box_child;
    uint8_t *box_level, *box_haschild;
    int32_t *bounds;            // [nprev][C+1]
    int32_t *nnew;              // [nprev]
    const int32_t *offsets;     // [nprev] exclusive scan of nnew
    LevelFlags *flags;
    DeviceStatus *status;
    int32_t max_weight;
    int level;                  // level being built
    int L, capbits;
    int idshift;                // packed keys: the key proper is keys[i] >> idshift (the id rides below)
    int b0, nprev;              // boxes of level-1: [b0, b0+nprev)
    int new_level_start;
    int adaptive;
    int keep_empty;             // skip_prune: empty children become boxes too
    const int32_t *parent_list; // level restriction: force-split these boxes (any level)
    int top_level;              // sharded builds: levels above it use global counts
    const int64_t *top_prefix;  // [C^top_level + 1] or null
    const int64_t *cell_starts; // THIS key array's cell starts at level cell_level (cell_starts_kernel),
    int cell_level;             // or null: child ranges down to that level are looked up, not searched
    const uint64_t *keys2;      // count_children_kernel (level-restricted builds): the key of the
    int L1k, L2k;               // levels L1k+1 .. L1k+L2k, for parents of level >= L1k
    const int64_t *top_arrive;  // particles with extents: per top box (levels 0..top_level, index
    const int64_t *top_stay;    // (C^level - 1) / (C - 1) + path) the global arrivals / stuck ones
    int loff;                   // the key addresses levels loff+1 .. loff+L (continuation keys)
    int can_continue;           // a continuation key exists below level loff+L
    const uint8_t *cand;        // continuation: which boxes of level loff were re-keyed
};

// index of the top box with Morton path `path` at `level` in the tables over levels 0..top_level
template <int D>
__device__ __forceinline__ int64_t top_box_index(uint64_t path, int level)
{
    constexpr uint64_t C = 1u << D;
    uint64_t off = 0, pw = 1;
    for (int l = 0; l < level; ++l) { off += pw; pw *= C; }
    return (int64_t) (off + path);
}

// global particle count of the box with Morton path `path` at `level` <= top_level: everything
// that arrives in it
template <int D>
__device__ __forceinline__ int32_t top_weight(const BuildArgs &a, uint64_t path, int level)
{
    int64_t w;
    if (a.top_arrive) {
        w = a.top_arrive[top_box_index<D>(path, level)];
    } else {
        const int sh = D * (a.top_level - level);
        w = a.top_prefix[(path + 1) << sh] - a.top_prefix[path << sh];
    }
    return (w > (int64_t) INT_MAX) ? INT_MAX : (int32_t) w;
}

// ... and the part of it bound for the box's children (tbk:569-573): with extents, what does not
// stick out of them
template <int D>
__device__ __forceinline__ int32_t top_descend_weight(const BuildArgs &a, uint64_t path, int level)
{
    if (!a.top_arrive) return top_weight<D>(a, path, level);
    const int64_t i = top_box_index<D>(path, level);
    const int64_t w = a.top_arrive[i] - a.top_stay[i];
    return (w > (int64_t) INT_MAX) ? INT_MAX : (int32_t) w;
}

__device__ __forceinline__ int32_t range_weight(const BuildArgs &a, int lo, int hi)
{
    if (!a.wprefix) return hi - lo;
    const int64_t w = a.wprefix[hi] - a.wprefix[lo];
    return (w > (int64_t) INT_MAX) ? INT_MAX : (int32_t) w;     // my_add_sat, tbk:270-274
}

// a new box has no children yet: its row of the child table in as few stores as its
// alignment allows (2^d ints at a multiple of 2^d ints from a 256-byte-aligned base)
template <int C>
__device__ __forceinline__ void zero_child_row(int32_t *row)
{
    if constexpr (C == 8) {
        int4 *r = reinterpret_cast<int4 *>(row);
        r[0] = make_int4(0, 0, 0, 0); r[1] = make_int4(0, 0, 0, 0);
    } else if constexpr (C == 4) {
        *reinterpret_cast<int4 *>(row) = make_int4(0, 0, 0, 0);
    } else {
        *reinterpret_cast<int2 *>(row) = make_int2(0, 0);
    }
}

// one thread per (box of level-1, child morton number)
template <int D, bool EXT>
__global__ __launch_bounds__(256) void count_children_kernel(BuildArgs a)
{
    constexpr int C = 1 << D;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bl = t / C, m = t % C;
    const bool active = bl < a.nprev;
    const bool forced = a.parent_list != nullptr;
    const int b = forced ? a.parent_list[active ? bl : 0] : a.b0 + (active ? bl : 0);
    const int l = forced ? (int) a.box_level[b] + 1 : a.level;

    int lo = 0, e = 0, s = 0;
    uint64_t prefix = 0;
    // a parent below the reach of the first key is searched in the continuation key
    const bool deep = a.keys2 != nullptr && l - 1 >= a.L1k;
    const uint64_t *keys = deep ? a.keys2 : a.keys;
    const int KL = deep ? a.L2k : a.L;
    const int lr = l - (deep ? a.L1k : a.loff);          // level relative to the key
    // continuation: boxes of level loff that were not re-keyed have nothing to split
    const bool skipped = a.cand != nullptr && !forced && active && !a.cand[bl];
    if (active) {
        s = a.box_start[b];
        e = s + a.box_count[b];
        if (lr - 1 < KL && e > s && !skipped) {
            const int pshift = a.capbits + D * (KL - (lr - 1));
            prefix = (pshift >= 64) ? 0 : (keys[s] >> pshift);
            const int cshift = a.capbits + D * (KL - lr);
            if (m == 0) {
                if (EXT) {
                    // own (stuck) particles: Kt == prefix000.., cap == lr-1
                    const uint64_t stuck = ((prefix << D) << cshift) | (uint64_t) (lr - 1);
                    lo = upper_bound_key(keys, s, e, stuck);
                } else {
                    lo = s;
                }
            } else {
                const uint64_t ck = ((prefix << D) | (uint64_t) m) << cshift;
                lo = lower_bound_key(keys, s, e, ck);
            }
        } else {
            lo = (m == 0) ? s : e;
        }
    }
    // children boundaries across the C lanes of the group
    int hi = __shfl_down(lo, 1, C);
    if (m == C - 1) hi = e;
    const int first = __shfl(lo, 0, C);     // start of the child-bound range
    if (!active) return;

    int32_t W = range_weight(a, first, e);                       // tbk:569-573
    const bool top = a.top_prefix && a.loff == 0 && l - 1 < a.top_level && e > s;
    if (top) W = top_descend_weight<D>(a, prefix, l - 1);
    bool split;
    if (forced) split = true;                                    // tbk:593-595
    else if (a.adaptive) split = W > a.max_weight;               // tbk:577-591
    else split = true;
    if (skipped) split = false;
    if (lr - 1 >= KL) {
        if (split && e > s && (a.adaptive || W > a.max_weight)) {
            // deeper than this key reaches: re-key the box's particles for the levels
            // below (tree_build_impl), or give up where the coordinate bits end
            if (m == 0) {
                if (a.can_continue && !deep)
                    __hip_atomic_store(&a.flags->need_more, 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                else
                    atomicExch(&a.status->max_levels, 1);
            }
        }
        split = false;
    }
    // empty boxes exist only with skip_prune; "non-adaptive" splits them like any
    // other box of the level (tbk:593-597), "adaptive" never does (weight 0)
    if (e == s && (a.adaptive || !a.keep_empty) && !forced) split = false;

    const int cnt = hi - lo;
    const bool nonempty = split && (cnt > 0 || a.keep_empty);
    const uint64_t bal = __ballot(nonempty);
    const int gshift = (threadIdx.x & 63) / C * C;
    const uint32_t gmask = (uint32_t) ((bal >> gshift) & ((1ull << C) - 1));
    const int32_t Wc = top ? top_weight<D>(a, (prefix << D) | (uint64_t) m, l)
                           : range_weight(a, lo, hi);
    if (split && cnt > 0 && Wc > a.max_weight) {                 // tbk:600-610
        // one (idempotent) store per wave at most, and none once the flag is up:
        // millions of same-address atomics serialise in L2
        if (__hip_atomic_load(&a.flags->have_oversize, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_AGENT) == 0)
            __hip_atomic_store(&a.flags->have_oversize, 1, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }

    a.bounds[(int64_t) bl * (C + 1) + m] = lo;
    if (m == 0) {
        a.bounds[(int64_t) bl * (C + 1) + C] = e;
        a.nnew[bl] = split ? __popc(gmask) : 0;
        a.box_haschild[b] = split ? 1 : 0;
        a.box_nonchild[b] = split ? (first - s) : 0;
    }
}

template <class T, int D>
__global__ __launch_bounds__(256) void write_children_kernel(BuildArgs a, T *centers /* [cap][D] */,
                                                            T root_extent)
{
    constexpr int C = 1 << D;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bl = t / C, m = t % C;
    const bool active = bl < a.nprev;
    const bool forced = a.parent_list != nullptr;
    const int b = forced ? a.parent_list[active ? bl : 0] : a.b0 + (active ? bl : 0);
    const int level = forced ? (int) a.box_level[b] + 1 : a.level;
    const bool split = active && a.box_haschild[b];
    int lo = 0, hi = 0;
    if (split) {
        lo = a.bounds[(int64_t) bl * (C + 1) + m];
        hi = a.bounds[(int64_t) bl * (C + 1) + m + 1];
    }
    const bool nonempty = split && (hi > lo || a.keep_empty);
    const uint64_t bal = __ballot(nonempty);
    if (!active) return;
    const int gshift = (threadIdx.x & 63) / C * C;
    const uint32_t gmask = (uint32_t) ((bal >> gshift) & ((1ull << C) - 1));
    const int rank = __popc(gmask & ((1u << m) - 1u));

    int32_t child_id = 0;
    if (nonempty) {
        child_id = a.new_level_start + a.offsets[bl] + rank;     // tbk:667 (after pruning)
        // an empty child (skip_prune) keeps start 0: tbk:680-695 only sets the start
        // "if the new box has particles to begin with"
        a.box_start[child_id] = hi > lo ? lo : 0;
        a.box_count[child_id] = hi - lo;
        a.box_parent[child_id] = b;
        a.box_level[child_id] = (uint8_t) level;
        a.box_haschild[child_id] = 0;
        a.box_nonchild[child_id] = 0;
        // tbk:698-705: centre = parent centre +/- root_extent / 2^(1+level)
        const T radius = (root_extent * 1 / (T) (1ull << (1 + level)));
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            const bool has_bit = (m >> (D - 1 - ax)) & 1;
            const T pc = centers[(int64_t) b * D + ax];
            centers[(int64_t) child_id * D + ax] = has_bit ? pc + radius : pc - radius;
        }
        zero_child_row<C>(a.box_child + (int64_t) child_id * C);
    }
    a.box_child[(int64_t) b * C + m] = child_id;
}

// ---------------------------------------------------------------------------
// One level in ONE launch, no host round trip: count_children + scan + write_children
// fused.  A workgroup takes tiles of 256 / C parent boxes in ticket order; a tile
// counts its children (boundaries stay in registers), learns the number of children
// created before it by decoupled look-back over the preceding tiles (the chained scan
// of bt_prims.hpp with a second field: "some child is still overfull") and writes its
// children at their final numbers.  The last tile records where the next level starts
// and whether any child must split again; the launch of the next level reads that
// from device memory, so the host can queue several levels blindly and look at the
// state once (tree_build.py:762-1060 spends a dozen host round trips per level here).
// ---------------------------------------------------------------------------

struct LoopState {
    int32_t level_start[BT_MAX_LEVELS + 3];
    int32_t oversize[BT_MAX_LEVELS + 3];    // level l was created with an overfull child
    int32_t done;                           // no further level is needed
    int32_t last_level;                     // deepest level created
    int32_t overflow;                       // box arrays too small: nothing was written
    int32_t need_more;                      // continuation keys needed
};

// descriptor: generation:16 | flag:2 | overfull tiles:14 (saturating) | children:32
__device__ __forceinline__ uint64_t sl_pack(uint32_t gen, uint32_t flag, uint32_t os, uint32_t sum)
{
    if (os > 0x3fffu) os = 0x3fffu;
    return ((uint64_t) (gen & 0xffffu) << 48) | ((uint64_t) flag << 46) | ((uint64_t) os << 32) | sum;
}

// sub-tiles per ticket: the look-back chain advances by some 40 tiles per microsecond
// (bt_prims.hpp), so a tile must hold enough parents -- 256 in 3D -- for the chain to
// keep up with the counting (32-parent tiles: 1.5 ms for the 10^6 parents of one level)
constexpr int SL_SUB = 8;

template <class T, int D, bool EXT>
__global__ __launch_bounds__(256) void split_level_kernel(BuildArgs a, LoopState *ls,
        T *centers /* [cap][D] */, T root_extent, int32_t box_cap, int first_level,
        uint64_t *desc, uint32_t gen, uint32_t *ticket, const T *rootbox)
{
    constexpr int C = 1 << D;
    constexpr int PPS = 256 / C;            // parents per sub-tile
    if (rootbox) root_extent = rootbox[6];  // root box computed on the device
    constexpr int PPT = PPS * SL_SUB;       // parents per tile
    __shared__ uint32_t s_tile;
    __shared__ int32_t s_scan[256 / 64 + 1];
    __shared__ int32_t s_excl;
    __shared__ int32_t s_os;

    if (ls->done || ls->overflow) return;
    const int level = a.level;
    const int b0 = ls->level_start[level - 1];
    const int nprev = ls->level_start[level] - b0;
    const int new_level_start = ls->level_start[level];
    // every workgroup takes the same decision from what earlier launches wrote
    if (level > first_level && (nprev == 0 || !ls->oversize[level - 1])) {   // tree_build.py:1228-1230
        if (blockIdx.x == 0 && threadIdx.x == 0) ls->done = 1;
        return;
    }
    const int ntiles = (nprev + PPT - 1) / PPT;
    const int lane = threadIdx.x & 63;
    const int m = threadIdx.x % C;
    const int gshift = lane / C * C;
    const int lr = level - a.loff;          // level relative to the key

    while (true) {
        __syncthreads();
        if (threadIdx.x == 0) { s_tile = atomicAdd(ticket, 1u); s_os = 0; }
        __syncthreads();
        const int tile = (int) s_tile;
        if (tile >= ntiles) break;

        // ---- children boundaries of SL_SUB sub-tiles (count_children_kernel) -------------
        // The binary searches of the sub-tiles advance in lockstep: every step issues
        // SL_SUB independent loads (one search at a time is a chain of ~17 dependent
        // loads per thread, and the kernel holds few waves per SIMD).
        int lo_[SL_SUB], hi_[SL_SUB], first_[SL_SUB], s_[SL_SUB], off_[SL_SUB];
        uint32_t bits_[SL_SUB];             // 1: split, 2: nonempty, 8: active, rank << 4
        int e_[SL_SUB];
        uint64_t pre_[SL_SUB], tgt_[SL_SUB];
        uint32_t act_[SL_SUB];              // 1: active, 2: skipped
        {
            int shi[SL_SUB];
#pragma unroll
            for (int k = 0; k < SL_SUB; ++k) {
                const int bl = tile * PPT + k * PPS + threadIdx.x / C;
                const bool active = bl < nprev;
                const int b = b0 + (active ? bl : 0);
                int lo = 0, hi = 0, e = 0, s = 0;       // search range [lo, hi)
                uint64_t prefix = 0, tgt = 0;
                const bool skipped = a.cand != nullptr && active && !a.cand[bl];
                if (active) {
                    s = a.box_start[b];
                    e = s + a.box_count[b];
                    // without extents the split decision needs no child boundary (all of
                    // the box's particles are child-bound): a box that stays a leaf -- most
                    // boxes of the deepest levels -- is not searched at all
                    bool will_split = true;
                    if (!EXT && a.adaptive && !(a.top_prefix && a.loff == 0 && level - 1 < a.top_level))
                        will_split = range_weight(a, s, e) > a.max_weight;
                    if (lr - 1 < a.L && e > s && !skipped && will_split) {
                        const int pshift = a.capbits + D * (a.L - (lr - 1));
                        prefix = (pshift >= 64) ? 0 : ((a.keys[s] >> a.idshift) >> pshift);
                        const int cshift = a.capbits + D * (a.L - lr);
                        // first position whose key is > tgt (upper bound); a lower bound
                        // of ck is the upper bound of ck - 1
                        if (m == 0) {
                            if (EXT) {
                                // own (stuck) particles: Kt == prefix000.., cap == lr-1
                                tgt = ((prefix << D) << cshift) | (uint64_t) (lr - 1);
                                lo = s; hi = e;
                            } else {
                                lo = hi = s;
                            }
                        } else if (!EXT && a.cell_starts && a.loff == 0 && level <= a.cell_level) {
                            // the child's first particle = the first particle of its first
                            // level-cell_level cell
                            const int sh = D * (a.cell_level - level);
                            lo = hi = (int) a.cell_starts[((prefix << D) | (uint64_t) m) << sh];
                        } else {
                            const uint64_t ck = ((prefix << D) | (uint64_t) m) << cshift;
                            if (ck == 0) { lo = hi = s; }
                            else { tgt = ck - 1; lo = s; hi = e; }
                        }
                    } else {
                        lo = hi = (m == 0) ? s : e;
                    }
                }
                lo_[k] = lo; shi[k] = hi; s_[k] = s; e_[k] = e; pre_[k] = prefix; tgt_[k] = tgt;
                act_[k] = (active ? 1u : 0u) | (skipped ? 2u : 0u);
            }
            bool any = true;
            while (any) {
                any = false;
#pragma unroll
                for (int k = 0; k < SL_SUB; ++k) {
                    if (lo_[k] < shi[k]) {
                        const int mid = lo_[k] + ((shi[k] - lo_[k]) >> 1);
                        if ((a.keys[mid] >> a.idshift) <= tgt_[k]) lo_[k] = mid + 1; else shi[k] = mid;
                        any = true;
                    }
                }
            }
        }
        int32_t run = 0;                    // children of the tile's earlier sub-tiles
        bool any_os = false;
#pragma unroll
        for (int k = 0; k < SL_SUB; ++k) {
            const bool active = act_[k] & 1u, skipped = act_[k] & 2u;
            const int lo = lo_[k], e = e_[k], s = s_[k];
            const uint64_t prefix = pre_[k];
            int hi = __shfl_down(lo, 1, C);
            if (m == C - 1) hi = e;
            const int first = __shfl(lo, 0, C);     // start of the child-bound range

            bool split = false, oversize = false;
            int cnt = 0;
            if (active) {
                int32_t W = range_weight(a, first, e);                       // tbk:569-573
                const bool top = a.top_prefix && a.loff == 0 && level - 1 < a.top_level && e > s;
                if (top) W = top_descend_weight<D>(a, prefix, level - 1);
                split = a.adaptive ? W > a.max_weight : true;                // tbk:577-597
                if (skipped) split = false;
                if (lr - 1 >= a.L) {
                    if (split && e > s && (a.adaptive || W > a.max_weight) && m == 0) {
                        if (a.can_continue)
                            __hip_atomic_store(&ls->need_more, 1, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                        else
                            atomicExch(&a.status->max_levels, 1);
                    }
                    split = false;
                }
                if (e == s && (a.adaptive || !a.keep_empty)) split = false;
                cnt = hi - lo;
                const int32_t Wc = top ? top_weight<D>(a, (prefix << D) | (uint64_t) m, level)
                                       : range_weight(a, lo, hi);
                oversize = split && cnt > 0 && Wc > a.max_weight;            // tbk:600-610
            }
            const bool nonempty = split && (cnt > 0 || a.keep_empty);
            const uint64_t bal = __ballot(nonempty);
            const uint32_t gmask = (uint32_t) ((bal >> gshift) & ((1ull << C) - 1));
            const int rank = __popc(gmask & ((1u << m) - 1u));
            const int nnew = (active && split && m == 0) ? __popc(gmask) : 0;
            any_os = any_os || (__ballot(oversize) != 0ull);

            int32_t sub_total = 0;
            const int32_t in_sub = block_exclusive_scan<int32_t, 256>(nnew, s_scan, &sub_total);
            hi_[k] = hi; first_[k] = first;
            off_[k] = run + __shfl(in_sub, 0, C);       // the group's m == 0 lane
            bits_[k] = (active ? 8u : 0u) | (split ? 1u : 0u) | (nonempty ? 2u : 0u) | ((uint32_t) rank << 4);
            run += sub_total;
        }
        const int32_t tile_total = run;
        if (any_os && lane == 0) s_os = 1;              // (benign race: same value)
        __syncthreads();
        const uint32_t tile_os = (uint32_t) s_os;

        // ---- look-back over the preceding tiles (wave 0) ---------------------------------
        if (threadIdx.x < 64) {
            uint32_t ex_sum = 0, ex_os = 0;
            if (tile > 0) {
                if (lane == 0)
                    __hip_atomic_store(desc + tile, sl_pack(gen, SP_AGG, tile_os, (uint32_t) tile_total),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int64_t look = (int64_t) tile - 1;
                uint32_t spins = 0;
                while (true) {
                    const int64_t idx = look - lane;
                    uint64_t word = sl_pack(gen, SP_PREFIX, 0, 0);       // before tile 0
                    if (idx >= 0)
                        word = __hip_atomic_load(desc + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t flag = (uint32_t) (word >> 46) & 3u;
                    const bool valid = (uint32_t) (word >> 48) == (gen & 0xffffu) && flag != 0;
                    const uint64_t is_prefix = __ballot(valid && flag == SP_PREFIX);
                    const uint64_t invalid = __ballot(!valid);
                    const int first_prefix = is_prefix ? __builtin_ctzll(is_prefix) : 64;
                    const int first_invalid = invalid ? __builtin_ctzll(invalid) : 64;
                    if (first_invalid <= first_prefix && first_invalid < 64) {
                        if (++spins > SP_SPIN_LIMIT) {
                            if (lane == 0) atomicExch(&a.status->internal, 78);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        continue;
                    }
                    const bool use = lane <= first_prefix;
                    ex_sum += wave_reduce_sum(use ? (uint32_t) word : 0u);
                    ex_os += wave_reduce_sum(use ? (uint32_t) (word >> 32) & 0x3fffu : 0u);
                    if (first_prefix < 64) break;
                    look -= 64;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(desc + tile,
                                   sl_pack(gen, SP_PREFIX, ex_os + tile_os, ex_sum + (uint32_t) tile_total),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_excl = (int32_t) ex_sum;
                if (tile == ntiles - 1) {
                    // the level is complete as far as the counts go
                    const int64_t end = (int64_t) new_level_start + ex_sum + (uint32_t) tile_total;
                    if (end > (int64_t) box_cap) {
                        ls->overflow = level;
                        ls->level_start[level + 1] = new_level_start;
                    } else {
                        ls->level_start[level + 1] = (int32_t) end;
                        ls->oversize[level] = (ex_os + tile_os) > 0 ? 1 : 0;
                        if (end > new_level_start) ls->last_level = level;
                    }
                }
            }
        }
        __syncthreads();
        const int32_t tile_excl = s_excl;

        // ---- children (write_children_kernel) -------------------------------------------------
        // tbk:698-705: centre = parent centre +/- root_extent / 2^(1+level)
        const T radius = (root_extent * 1 / (T) (1ull << (1 + level)));
#pragma unroll
        for (int k = 0; k < SL_SUB; ++k) {
            if (!(bits_[k] & 8u)) continue;
            const int bl = tile * PPT + k * PPS + threadIdx.x / C;
            const int b = b0 + bl;
            const bool split = bits_[k] & 1u, nonempty = bits_[k] & 2u;
            const int rank = (int) (bits_[k] >> 4);
            int32_t child_id = 0;
            const int64_t cid = (int64_t) new_level_start + tile_excl + off_[k] + rank;
            if (nonempty && cid < (int64_t) box_cap) {
                child_id = (int32_t) cid;                                  // tbk:667 (after pruning)
                // an empty child (skip_prune) keeps start 0: tbk:680-695 only sets the start
                // "if the new box has particles to begin with"
                a.box_start[child_id] = hi_[k] > lo_[k] ? lo_[k] : 0;
                a.box_count[child_id] = hi_[k] - lo_[k];
                a.box_parent[child_id] = b;
                a.box_level[child_id] = (uint8_t) level;
                a.box_haschild[child_id] = 0;
                a.box_nonchild[child_id] = 0;
#pragma unroll
                for (int ax = 0; ax < D; ++ax) {
                    const bool has_bit = (m >> (D - 1 - ax)) & 1;
                    const T pc = centers[(int64_t) b * D + ax];
                    centers[(int64_t) child_id * D + ax] = has_bit ? pc + radius : pc - radius;
                }
                zero_child_row<C>(a.box_child + (int64_t) child_id * C);
            }
            a.box_child[(int64_t) b * C + m] = child_id;
            if (m == 0) {
                a.box_haschild[b] = split ? 1 : 0;
                a.box_nonchild[b] = split ? (first_[k] - s_[k]) : 0;
            }
        }
    }
}

// first particle of every level-k Morton cell in the sorted keys (lower bounds; out[ncells]
// = n): the child ranges of the top k levels are read from here instead of being found by a
// binary search over all n keys per child -- 20 to 27 dependent loads each, which is what
// the launches of the small top levels spend their time on
__global__ __launch_bounds__(256) void cell_starts_kernel(const uint64_t *keys, int64_t n, int shift,
        int64_t ncells, int64_t *out)
{
    const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (c > ncells) return;
    if (c == ncells) { out[c] = n; return; }
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t) (keys[mid] >> shift) < c) lo = mid + 1; else hi = mid;
    }
    out[c] = lo;
}

template <class T, int D>
__global__ void init_root_kernel(BuildArgs a, LoopState *ls, T *centers, int32_t n, T bbox_min0,
        T bbox_min1, T bbox_min2, T bbox_max0, T bbox_max1, T bbox_max2, uint32_t *tickets,
        const T *rootbox)
{
    constexpr int C = 1 << D;
    const int t = threadIdx.x;
    if (t == 0) {
        // root box: tree_build.py:585-618
        a.box_start[0] = 0; a.box_count[0] = n; a.box_parent[0] = 0;
        a.box_level[0] = 0; a.box_haschild[0] = 0; a.box_nonchild[0] = 0;
        T mn[3] = {bbox_min0, bbox_min1, bbox_min2}, mx[3] = {bbox_max0, bbox_max1, bbox_max2};
        if (rootbox)
            for (int ax = 0; ax < 3; ++ax) { mn[ax] = rootbox[ax]; mx[ax] = rootbox[3 + ax]; }
        for (int ax = 0; ax < D; ++ax) centers[ax] = mn[ax] + (mx[ax] - mn[ax]) / 2;
        for (int m = 0; m < C; ++m) a.box_child[m] = 0;
        ls->level_start[0] = 0; ls->level_start[1] = 1;
        ls->done = 0; ls->last_level = 0; ls->overflow = 0; ls->need_more = 0;
    }
    if (t < BT_MAX_LEVELS + 3) { ls->oversize[t] = 0; tickets[t] = 0; }
    if (t >= 2 && t < BT_MAX_LEVELS + 3) ls->level_start[t] = 1;
}

// after a capacity overflow at level `from`: forget what the launches from that level on did
__global__ void reset_loop_kernel(LoopState *ls, uint32_t *tickets, int from)
{
    const int t = threadIdx.x;
    if (t == 0) { ls->done = 0; ls->overflow = 0; }
    if (t >= from && t < BT_MAX_LEVELS + 3) { ls->oversize[t] = 0; tickets[t] = 0; }
    if (t > from && t < BT_MAX_LEVELS + 3) ls->level_start[t] = ls->level_start[from];
}

struct ScanNnew {
    const int32_t *nnew;
    __device__ int32_t operator()(int64_t i) const { return nnew[i]; }
};

struct GatherWeight {
    const int32_t *w;
    const uint32_t *ids;
    __device__ int64_t operator()(int64_t i) const { return (int64_t) w[ids[i]]; }
};

// ---------------------------------------------------------------------------
// within-box order fix-up: key = start of the owning segment, scattered to
// user order; a stable sort of (key, 0..N-1) then yields the reference's
// "ascending user id inside every box" order (tbk:776-790 is stable).
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void segment_key_kernel(int nboxes, const int32_t *box_start,
        const int32_t *box_count, const int32_t *box_nonchild, const uint8_t *box_haschild,
        const uint32_t *ids, uint32_t *fix_key /* [N], user order */)
{
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4;   // 16 lanes per box
    const int l16 = threadIdx.x & 15;
    if (g >= nboxes) return;
    const int s = box_start[g];
    const int n = box_haschild[g] ? box_nonchild[g] : box_count[g];
    for (int p = s + l16; p < s + n; p += 16) fix_key[ids[p]] = (uint32_t) s;
}

// Fast fix-up: a leaf's particles are a short contiguous run, so sort the ids of
// every run in place -- one wave per box for runs <= 64 (bitonic network on
// shuffles), one workgroup for runs <= SEG_BLOCK_MAX; anything longer (zero
// refine weights, many stuck particles) falls back to the global sort above.
constexpr int SEG_BLOCK_MAX = 4096;

struct SegSortFlags {
    int32_t n_large;      // runs in (64, SEG_BLOCK_MAX]
    int32_t has_huge;     // some run > SEG_BLOCK_MAX
    int32_t n_copy;       // packed keys with extents: runs > 64 whose ids a second kernel unpacks
};

// Half a wave per box, up to two ids per lane: the kernel is bound by the latency of
// its dependent global loads (count, start, ids), so two boxes per wave keep twice as
// many of them in flight; the 64-element network's j=32 stage is register-local.
// (The exchanges go through __shfl_xor, i.e. ds_bpermute.  DPP moves for the distances
// inside a row of 16 lanes -- 1 LDS round trip instead of 15 -- were measured 13 % SLOWER
// at 10^8 points: the kernel is bound by vector-ALU issue, and the shuffles run on the LDS
// pipeline beside it.)
// PACKED: the ids are the low bits of the packed keys (above); every leaf's ids are written
// to `ids`, ordered or not, and a leaf of one particle is copied.
// (Two boxes per half-wave, their loads issued together: the kernel holds all the waves a
// SIMD takes and still spends its time waiting for the chain box record -> ids; with the
// loads of two boxes in flight per lane a wave covers its latency twice as well.)
constexpr int SEG_BOXES_PER_HALF_WAVE = 2;

template <bool PACKED>
__global__ __launch_bounds__(256) void segment_sort_wave_kernel(int nboxes, const int32_t *box_start,
        const int32_t *box_count, const uint8_t *box_haschild, uint32_t *ids,
        int32_t *large_list, SegSortFlags *flags, const uint64_t *pk, uint64_t id_mask)
{
    constexpr int NB = SEG_BOXES_PER_HALF_WAVE;
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 5;
    const int lane = threadIdx.x & 31;
    int n[NB], s[NB];
    bool act[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int b = g * NB + q;
        const bool in = b < nboxes;
        const uint8_t has_children = in ? box_haschild[b] : 1;
        n[q] = in ? box_count[b] : 0;
        s[q] = in ? box_start[b] : 0;
        // a split box's own particles share one key: the stable sort left them in id order
        // (packed keys: point particles, a split box has none of its own)
        act[q] = in && !has_children && n[q] > (PACKED ? 0 : 1);
        if (act[q] && n[q] > 64) {
            if (lane == 0) {
                if (n[q] <= SEG_BLOCK_MAX) large_list[atomicAdd(&flags->n_large, 1)] = b;
                else atomicExch(&flags->has_huge, 1);
            }
            act[q] = false;
        }
    }
    uint32_t v0[NB], v1[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        v0[q] = v1[q] = 0xFFFFFFFFu;
        if (act[q]) {
            if (PACKED) {
                if (lane < n[q]) v0[q] = (uint32_t) (pk[s[q] + lane] & id_mask);
                if (lane + 32 < n[q]) v1[q] = (uint32_t) (pk[s[q] + lane + 32] & id_mask);
            } else {
                if (lane < n[q]) v0[q] = ids[s[q] + lane];
                if (lane + 32 < n[q]) v1[q] = ids[s[q] + lane + 32];
            }
        }
    }
    // element index of v0 is `lane`, of v1 `lane + 32`
    auto cmpx = [&](uint32_t v, int idx, int k, int j) {
        const uint32_t o = __shfl_xor(v, j, 32);
        const bool up = (idx & k) == 0;
        const bool lower = (idx & j) == 0;
        const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
        return (lower == up) ? mn : mx;
    };
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        if (!act[q]) continue;                  // (uniform over the half-wave)
        uint32_t a0 = v0[q], a1 = v1[q];
        const int nq = n[q];
#pragma unroll
        for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                a0 = cmpx(a0, lane, k, j);
                if (nq > 32) a1 = cmpx(a1, lane + 32, k, j);
            }
        }
        if (nq > 32) {
            // k = 64: both halves ascending overall; j = 32 pairs a0 with a1 of the same lane
            {
                const uint32_t mn = a0 < a1 ? a0 : a1, mx = a0 < a1 ? a1 : a0;
                a0 = mn; a1 = mx;
            }
#pragma unroll
            for (int j = 16; j > 0; j >>= 1) {
                a0 = cmpx(a0, lane, 64, j);
                a1 = cmpx(a1, lane + 32, 64, j);
            }
        }
        if (lane < nq) ids[s[q] + lane] = a0;
        if (lane + 32 < nq) ids[s[q] + lane + 32] = a1;
    }
}

// Packed keys with extents (path | cap | id): runs of any length occur -- the particles stuck
// in a split box, which keep their order, and leaves --, and the global fix-up route needs
// every id in the id array.  This kernel unpacks every run of at most 64 ids, ordering the
// leaves among them; longer runs go on a list for segment_unpack_block_kernel (and, leaves,
// on the lists of the ordering kernels, which then work on the id array as always).
__global__ __launch_bounds__(256) void segment_unpack_sort_wave_kernel(int nboxes, const int32_t *box_start,
        const int32_t *box_count, const int32_t *box_nonchild, const uint8_t *box_haschild, uint32_t *ids,
        int32_t *large_list, int32_t *copy_list, SegSortFlags *flags, const uint64_t *pk, uint64_t id_mask)
{
    constexpr int NB = SEG_BOXES_PER_HALF_WAVE;
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 5;
    const int lane = threadIdx.x & 31;
    int n[NB], s[NB];
    bool act[NB], leaf[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const int b = g * NB + q;
        const bool in = b < nboxes;
        const uint8_t has_children = in ? box_haschild[b] : 1;
        leaf[q] = !has_children;
        n[q] = in ? (has_children ? box_nonchild[b] : box_count[b]) : 0;
        s[q] = in ? box_start[b] : 0;
        act[q] = in && n[q] > 0;
        if (act[q] && n[q] > 64) {
            if (lane == 0) {
                copy_list[atomicAdd(&flags->n_copy, 1)] = b;
                if (leaf[q]) {
                    if (n[q] <= SEG_BLOCK_MAX) large_list[atomicAdd(&flags->n_large, 1)] = b;
                    else atomicExch(&flags->has_huge, 1);
                }
            }
            act[q] = false;
        }
    }
    uint32_t v0[NB], v1[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        v0[q] = v1[q] = 0xFFFFFFFFu;
        if (act[q]) {
            if (lane < n[q]) v0[q] = (uint32_t) (pk[s[q] + lane] & id_mask);
            if (lane + 32 < n[q]) v1[q] = (uint32_t) (pk[s[q] + lane + 32] & id_mask);
        }
    }
    auto cmpx = [&](uint32_t v, int idx, int k, int j) {
        const uint32_t o = __shfl_xor(v, j, 32);
        const bool up = (idx & k) == 0;
        const bool lower = (idx & j) == 0;
        const uint32_t mn = v < o ? v : o, mx = v < o ? o : v;
        return (lower == up) ? mn : mx;
    };
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        if (!act[q]) continue;                  // (uniform over the half-wave)
        uint32_t a0 = v0[q], a1 = v1[q];
        const int nq = n[q];
        if (leaf[q] && nq > 1) {
#pragma unroll
            for (int k = 2; k <= 32; k <<= 1) {
#pragma unroll
                for (int j = k >> 1; j > 0; j >>= 1) {
                    a0 = cmpx(a0, lane, k, j);
                    if (nq > 32) a1 = cmpx(a1, lane + 32, k, j);
                }
            }
            if (nq > 32) {
                {
                    const uint32_t mn = a0 < a1 ? a0 : a1, mx = a0 < a1 ? a1 : a0;
                    a0 = mn; a1 = mx;
                }
#pragma unroll
                for (int j = 16; j > 0; j >>= 1) {
                    a0 = cmpx(a0, lane, 64, j);
                    a1 = cmpx(a1, lane + 32, 64, j);
                }
            }
        }
        if (lane < nq) ids[s[q] + lane] = a0;
        if (lane + 32 < nq) ids[s[q] + lane + 32] = a1;
    }
}

__global__ __launch_bounds__(256) void segment_unpack_block_kernel(const int32_t *copy_list,
        const SegSortFlags *flags, const int32_t *box_start, const int32_t *box_count,
        const int32_t *box_nonchild, const uint8_t *box_haschild, uint32_t *ids, const uint64_t *pk,
        uint64_t id_mask)
{
    const int nc = flags->n_copy;
    for (int r = blockIdx.x; r < nc; r += gridDim.x) {
        const int b = copy_list[r];
        const int s = box_start[b], n = box_haschild[b] ? box_nonchild[b] : box_count[b];
        for (int i = threadIdx.x; i < n; i += 256) ids[s + i] = (uint32_t) (pk[s + i] & id_mask);
    }
}

// The number of listed runs is read on the device (flags->n_large): the launch needs no
// host round trip; a run longer than the workgroup sort on a build that cannot have one
// is reported through the status word.
template <bool PACKED>
__global__ __launch_bounds__(256) void segment_sort_block_kernel(const int32_t *large_list,
        const SegSortFlags *flags, int huge_is_error, DeviceStatus *status,
        const int32_t *box_start, const int32_t *box_count, uint32_t *ids,
        const uint64_t *pk, uint64_t id_mask)
{
    __shared__ uint32_t s_v[SEG_BLOCK_MAX];
    if (huge_is_error && flags->has_huge && blockIdx.x == 0 && threadIdx.x == 0)
        atomicExch(&status->internal, 41);
    const int nl = flags->n_large;
    for (int r = blockIdx.x; r < nl; r += gridDim.x) {
        const int b = large_list[r];
        const int s = box_start[b], n = box_count[b];
        int m = 128;
        while (m < n) m <<= 1;
        for (int i = threadIdx.x; i < m; i += 256)
            s_v[i] = (i < n) ? (PACKED ? (uint32_t) (pk[s + i] & id_mask) : ids[s + i]) : 0xFFFFFFFFu;
        __syncthreads();
        for (int k = 2; k <= m; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < m; i += 256) {
                    const int l = i ^ j;
                    if (l > i) {
                        const uint32_t a = s_v[i], c = s_v[l];
                        const bool up = (i & k) == 0;
                        if ((a > c) == up) { s_v[i] = c; s_v[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < n; i += 256) ids[s + i] = s_v[i];
        __syncthreads();
    }
}

// (Leaves in one pass -- ids ordered, written out, coordinates gathered and the leaf's bounding box
// reduced while they are in registers, a half-wave per leaf -- measured 4.5 ms at 10^8 sphere points
// against 0.9 + 2.6 + 0.4 ms for the separate id sort, gather and leaf extents: the random 32-byte
// gathers keep 40 % of a half-wave per ~25-particle leaf busy, a thread per particle all of them.
// Dropped with its switch in round 6.)

// ---------------------------------------------------------------------------
// sources / targets (tbk:1013-1164, 1770-1782; tools.py:81-109)
// ---------------------------------------------------------------------------

struct IsSource {
    const uint32_t *ids;
    uint32_t nsources;
    __device__ int32_t operator()(int64_t i) const { return ids[i] < nsources ? 1 : 0; }
};

// Sources before a tree-order position, without an N-sized prefix array: one bit per position
// ("is a source") in 64-bit words and the number of sources before every word.  S(p) is two
// loads and a population count; a kernel that walks the positions in order gets the word
// from a ballot of its own flags and loads one count per wave.
struct SrcPrefix {
    const uint64_t *bits;       // [ceil(N / 64)]
    const int32_t *before;      // [ceil(N / 64) + 1] sources before the word
    __device__ __forceinline__ int32_t operator()(int64_t p) const
    {
        const int64_t w = p >> 6;
        const int r = (int) (p & 63);
        if (r == 0) return before[w];
        return before[w] + __popcll(bits[w] & ((1ull << r) - 1ull));
    }
};

// one wave per 64 positions: the word of flags and its population count
__global__ __launch_bounds__(256) void source_bits_kernel(int64_t n, const uint32_t *ids, uint32_t nsources,
                                                          uint64_t *bits, int32_t *counts)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const bool src = p < n && ids[p] < nsources;
    const uint64_t bal = __ballot(src);
    if ((threadIdx.x & 63) == 0 && (p >> 6) < ((n + 63) >> 6)) {
        bits[p >> 6] = bal;
        counts[p >> 6] = __popcll(bal);
    }
}

struct ScanCounts {
    const int32_t *c;
    __device__ int32_t operator()(int64_t i) const { return c[i]; }
};

__global__ __launch_bounds__(256) void split_ids_kernel(int64_t n, const uint32_t *ids,
        const int32_t *words_before, uint32_t nsources, int32_t *user_source_ids,
        int32_t *srcntgt_target_ids, int32_t *sorted_target_ids)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const bool in = p < n;
    const uint32_t id = in ? ids[p] : 0xffffffffu;
    // (a wave covers exactly one 64-position word)
    const uint64_t bal = __ballot(in && id < nsources);
    if (!in) return;
    const int lane = threadIdx.x & 63;
    const int32_t source_nr = words_before[p >> 6] + __popcll(bal & ((1ull << lane) - 1ull));
    if (id < nsources) {
        user_source_ids[source_nr] = (int32_t) id;
    } else {
        const int32_t target_nr = (int32_t) p - source_nr;
        srcntgt_target_ids[target_nr] = (int32_t) id;
        sorted_target_ids[id - nsources] = target_nr;
    }
}

// (A fused ids + 3-axis gather kernel was measured 1.5x SLOWER than the separate
// passes at 1e8 points: four concurrent random streams over 2.8 GB thrash the
// TLB; one random stream per pass does not.)
__global__ __launch_bounds__(256) void same_ids_kernel(int64_t n, const uint32_t *ids,
        int32_t *user_source_ids, int32_t *sorted_target_ids)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t id = ids[p];
    user_source_ids[p] = (int32_t) id;
    if ((int64_t) id < n) sorted_target_ids[id] = (int32_t) p;   // reverse_index_array, tools.py:81-109
}

// inverse permutation from (id, position) pairs that one radix pass has grouped by
// the top byte of the id: the writes of neighbouring pairs fall into one window of
// n/256 ids (2 MB at 10^8), and with the workgroups of an XCD kept on one
// contiguous stretch of pairs that window stays in the XCD's L2 until its lines
// are complete.
__global__ __launch_bounds__(256) void scatter_inverse_kernel(int64_t n, const uint32_t *ids,
        const uint32_t *positions, int32_t *sorted_target_ids)
{
    const int64_t nblocks = gridDim.x;
    const int64_t per_xcd = (nblocks + 7) / 8;
    const int64_t logical = (int64_t) (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    const int64_t j = logical * 256 + threadIdx.x;
    if (j >= n) return;
    // (the pairs stream through once: read past the L2's replacement order so that they do
    // not push out the window the writes are being merged in)
    constexpr bool nt = true;
    const uint32_t id = nt ? __builtin_nontemporal_load(ids + j) : ids[j];
    const uint32_t pos = nt ? __builtin_nontemporal_load(positions + j) : positions[j];
    if ((int64_t) id < n) sorted_target_ids[id] = (int32_t) pos;
}

__global__ __launch_bounds__(256) void copy_ids_kernel(int64_t n, const uint32_t *ids,
                                                       int32_t *user_source_ids)
{
    const int64_t p = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (p < n) user_source_ids[p] = (int32_t) ids[p];
}

template <class T, int D>
struct GatherOut { T *out[D]; };

template <class T, int D>
__global__ __launch_bounds__(256) void gather_packed_kernel(int64_t n, const int32_t *from_ids,
        const T *__restrict__ packed, GatherOut<T, D> g)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t id = __builtin_nontemporal_load(&from_ids[i]);   // srcntgt numbering
    constexpr int PS = PackStride<D>::value;
    T c[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) c[ax] = packed[id * PS + ax];    // tbk:1170-1186
#pragma unroll
    for (int ax = 0; ax < D; ++ax) __builtin_nontemporal_store(c[ax], &g.out[ax][i]);
}

template <class T>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const int32_t *from_ids,
        int32_t id_offset, const T *__restrict__ in, T *__restrict__ out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = in[from_ids[i] - id_offset];      // tbk:1170-1186
}

// ---------------------------------------------------------------------------
// per-box outputs: counts, flags (tbk:1192-1305), repack (tree_build.py:1636-1664)
// ---------------------------------------------------------------------------

struct BoxInfoArgs {
    int nboxes;
    int64_t aligned;
    int C, D;
    int sat, have_extent;
    const int32_t *box_start, *box_count, *box_parent, *box_nonchild, *box_child;
    const uint8_t *box_level, *box_haschild;
    SrcPrefix src_prefix;          // (sat: unused)
    int32_t *o_src_starts, *o_src_nonchild, *o_src_cumul;
    int32_t *o_tgt_starts, *o_tgt_nonchild, *o_tgt_cumul;
    int32_t *o_parent, *o_child;
    uint8_t *o_levels, *o_flags;
    const void *centers;           // [nboxes][D], T = float (csize 4) or double (8)
    void *o_centers;               // [D][aligned]
    int csize;
    void *o_extents[4];            // [D][aligned] arrays whose padding columns are zeroed (or null)
    int32_t *o_level_starts;       // [n_level_starts] or null
    int32_t *o_sizes;              // [nboxes] boxes per subtree: 1 here, box_extent_kernel adds
                                   // the children's (or null)
    int n_level_starts;
    int32_t level_starts[BT_MAX_LEVELS + 1];
};

// One thread per box of the padded range [0, aligned): exported per-box arrays, centres in
// the reference's [d][aligned] layout, zeros in the padding columns of the two 2-D arrays,
// and (first workgroup) the level starts, which travel as kernel arguments.
__global__ __launch_bounds__(256) void box_info_kernel(BoxInfoArgs a)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && a.o_level_starts) {
        int32_t v = 0;
#pragma unroll
        for (int i = 0; i <= BT_MAX_LEVELS; ++i) v = ((int) threadIdx.x == i) ? a.level_starts[i] : v;
        if ((int) threadIdx.x < a.n_level_starts) a.o_level_starts[threadIdx.x] = v;
    }
    if (b >= a.nboxes) {
        if (b < a.aligned) {
            for (int m = 0; m < a.C; ++m) a.o_child[(int64_t) m * a.aligned + b] = 0;
            for (int ax = 0; ax < a.D; ++ax) {
                if (a.csize == 8) ((double *) a.o_centers)[(int64_t) ax * a.aligned + b] = 0.0;
                else ((float *) a.o_centers)[(int64_t) ax * a.aligned + b] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!a.o_extents[k]) continue;
                    if (a.csize == 8) ((double *) a.o_extents[k])[(int64_t) ax * a.aligned + b] = 0.0;
                    else ((float *) a.o_extents[k])[(int64_t) ax * a.aligned + b] = 0.f;
                }
            }
        }
        return;
    }
    for (int ax = 0; ax < a.D; ++ax) {
        if (a.csize == 8)
            ((double *) a.o_centers)[(int64_t) ax * a.aligned + b] =
                ((const double *) a.centers)[(int64_t) b * a.D + ax];
        else
            ((float *) a.o_centers)[(int64_t) ax * a.aligned + b] =
                ((const float *) a.centers)[(int64_t) b * a.D + ax];
    }
    const int s = a.box_start[b], cnt = a.box_count[b];
    const int haschild = a.box_haschild[b];
    const int n0 = a.box_nonchild[b];      // own particles of a split box (0 w/o extents)
    int32_t src_start, src_cumul, tgt_start, tgt_cumul, src_nc, tgt_nc;
    if (a.sat) {
        src_start = tgt_start = s;
        src_cumul = tgt_cumul = cnt;
        src_nc = tgt_nc = haschild ? n0 : cnt;
    } else {
        const int32_t S0 = a.src_prefix(s), S1 = a.src_prefix((int64_t) s + cnt);
        src_start = S0; tgt_start = s - S0;
        src_cumul = S1 - S0; tgt_cumul = cnt - src_cumul;
        if (haschild) {
            const int32_t Sn = a.src_prefix((int64_t) s + n0);
            src_nc = Sn - S0; tgt_nc = n0 - src_nc;
        } else {
            src_nc = src_cumul; tgt_nc = tgt_cumul;
        }
    }
    uint8_t flags = 0;
    if (haschild) {
        flags |= BT_BOX_HAS_SOURCE_CHILD_BOXES | BT_BOX_HAS_TARGET_CHILD_BOXES;   // tbk:1252-1256
        if (src_nc) flags |= BT_BOX_IS_SOURCE_BOX;
        if (tgt_nc) flags |= BT_BOX_IS_TARGET_BOX;
    } else {
        if (src_cumul) flags |= BT_BOX_IS_SOURCE_BOX;
        if (tgt_cumul) flags |= BT_BOX_IS_TARGET_BOX;
    }
    a.o_src_starts[b] = src_start; a.o_src_cumul[b] = src_cumul; a.o_src_nonchild[b] = src_nc;
    if (!a.sat) {
        a.o_tgt_starts[b] = tgt_start; a.o_tgt_cumul[b] = tgt_cumul; a.o_tgt_nonchild[b] = tgt_nc;
    }
    a.o_parent[b] = a.box_parent[b];
    a.o_levels[b] = a.box_level[b];
    a.o_flags[b] = flags;
    if (a.o_sizes) a.o_sizes[b] = 1;
    for (int m = 0; m < a.C; ++m)
        a.o_child[(int64_t) m * a.aligned + b] = a.box_child[(int64_t) b * a.C + m];
}

// value of the lane CTRL names within the same row of 16 lanes (0x100 + n: row_shl:n, lane
// i reads lane i + n); lanes without a source keep their own value
template <int CTRL>
__device__ __forceinline__ float dpp_row_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL,
                                                      0xf, 0xf, false));
}

template <int CTRL>
__device__ __forceinline__ double dpp_row_mov(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false));
}

// box extents: tbk:1311-1399, one 16-lane group per box of one level
template <class T, int D>
struct ExtentArgs {
    int b0, nb;
    int64_t aligned;
    const int32_t *starts, *counts_nonchild, *child /* [C][aligned] */;
    const T *centers /* [D][aligned] */;
    const T *part[D];
    const T *radii;          // null if disabled
    T *bmin, *bmax;          // [D][aligned]
    int leaves_done;         // leaf_gather_*_kernel wrote the extents of the leaves
    int32_t *sizes;          // [nboxes] boxes per subtree, 1 on entry (or null): the sweep is
                             // bottom-up over the child table anyway, so the traversal's
                             // depth-first ranks need no sweep of their own
};

template <class T, int D>
__global__ __launch_bounds__(256) void box_extent_kernel(ExtentArgs<T, D> a)
{
    constexpr int C = 1 << D;
    const int g = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const int l16 = threadIdx.x & 15;
    bool active = g < a.nb;
    const int b = a.b0 + (active ? g : 0);
    if (active && a.leaves_done) {
        // only boxes with children are left to do (whole 16-lane groups drop out)
        bool any_child = false;
        for (int m = 0; m < C; ++m) any_child = any_child || a.child[(int64_t) m * a.aligned + b] != 0;
        if (!any_child) return;
    }
    T mn[D], mx[D];
    int32_t below = 0;       // boxes under child l16
#pragma unroll
    for (int ax = 0; ax < D; ++ax) mn[ax] = mx[ax] = a.centers[(int64_t) ax * a.aligned + b];
    if (active) {
        // Two chains of dependent loads per box -- range -> coordinates, child -> its extents --
        // and a box is a few dozen particles: what bounds this kernel is how many loads a lane
        // has in flight, so both chains start together and the first 64 own particles (every
        // particle of an ordinary leaf) are fetched in one go before anything is reduced.
        const int s = a.starts[b], e = s + a.counts_nonchild[b];
        const int32_t ch = l16 < C ? a.child[(int64_t) l16 * a.aligned + b] : 0;
        constexpr int UNR = 4;
        T pc[UNR][D], pr[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int p = s + l16 + 16 * k;
            const bool in = p < e;
            pr[k] = (in && a.radii) ? a.radii[p] : (T) 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) pc[k][ax] = in ? a.part[ax][p] : mn[ax];
        }
        T cmn[D], cmx[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) { cmn[ax] = mn[ax]; cmx[ax] = mx[ax]; }
        if (ch != 0) {
            if (a.sizes) below = a.sizes[ch];
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                cmn[ax] = a.bmin[(int64_t) ax * a.aligned + ch];
                cmx[ax] = a.bmax[(int64_t) ax * a.aligned + ch];
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const T lo = pc[k][ax] - pr[k], hi = pc[k][ax] + pr[k];
                mn[ax] = (lo < mn[ax]) ? lo : mn[ax];
                mx[ax] = (hi > mx[ax]) ? hi : mx[ax];
            }
        }
        for (int p = s + l16 + 16 * UNR; p < e; p += 16) {
            const T r = a.radii ? a.radii[p] : (T) 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) {
                const T c = a.part[ax][p];
                const T lo = c - r, hi = c + r;
                mn[ax] = (lo < mn[ax]) ? lo : mn[ax];
                mx[ax] = (hi > mx[ax]) ? hi : mx[ax];
            }
        }
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            mn[ax] = (cmn[ax] < mn[ax]) ? cmn[ax] : mn[ax];
            mx[ax] = (cmx[ax] > mx[ax]) ? cmx[ax] : mx[ax];
        }
    }
    // reduction over the group's 16 lanes (one DPP row) into lane 0: row_shl:8/4/2/1 moves
    // instead of shuffles through LDS (48 ds_bpermute per lane in double precision)
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        T omn, omx;
        omn = dpp_row_mov<0x108>(mn[ax]); omx = dpp_row_mov<0x108>(mx[ax]);
        mn[ax] = (omn < mn[ax]) ? omn : mn[ax]; mx[ax] = (omx > mx[ax]) ? omx : mx[ax];
        omn = dpp_row_mov<0x104>(mn[ax]); omx = dpp_row_mov<0x104>(mx[ax]);
        mn[ax] = (omn < mn[ax]) ? omn : mn[ax]; mx[ax] = (omx > mx[ax]) ? omx : mx[ax];
        omn = dpp_row_mov<0x102>(mn[ax]); omx = dpp_row_mov<0x102>(mx[ax]);
        mn[ax] = (omn < mn[ax]) ? omn : mn[ax]; mx[ax] = (omx > mx[ax]) ? omx : mx[ax];
        omn = dpp_row_mov<0x101>(mn[ax]); omx = dpp_row_mov<0x101>(mx[ax]);
        mn[ax] = (omn < mn[ax]) ? omn : mn[ax]; mx[ax] = (omx > mx[ax]) ? omx : mx[ax];
    }
    if (a.sizes) {
        // lanes >= 2^d hold 0: the sum of the first 8 lanes lands in lane 0
        below += __builtin_amdgcn_update_dpp(0, below, 0x104, 0xf, 0xf, true);
        below += __builtin_amdgcn_update_dpp(0, below, 0x102, 0xf, 0xf, true);
        below += __builtin_amdgcn_update_dpp(0, below, 0x101, 0xf, 0xf, true);
    }
    if (active && l16 == 0) {
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            a.bmin[(int64_t) ax * a.aligned + b] = mn[ax];
            a.bmax[(int64_t) ax * a.aligned + b] = mx[ax];
        }
        if (a.sizes) a.sizes[b] = 1 + below;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

struct TreeState {
    bt_tree_params p{};
    int D = 0, C = 0, L = 0, capbits = 0;
    bool f64 = true, sat = true, have_extent = false;
    int64_t N = 0, nsources = 0, ntargets = 0;
    int64_t nboxes = 0, cap = 0;
    std::vector<int32_t> level_start;

    Buf<uint64_t> keys_a, keys_b;
    Buf<uint32_t> ids_a, ids_b;
    const uint64_t *pk = nullptr;      // packed keys (path << idbits | id) still holding the ids:
    uint64_t pk_mask = 0;              // the fix-up unpacks them into `ids`
    uint32_t *ids = nullptr;           // final tree order -> user srcntgt id
    uint32_t *ids_other = nullptr;     // the other id buffer (scratch of the global fix-up)
    bool fixup_done = false;           // ids are in the reference's within-box order
    // the fix-up is started by the build and finished by the export (fixup_launch /
    // fixup_finish): the caller allocates its arrays in between, while the kernel runs
    Buf<unsigned char> rootbox;        // device {min[3], max[3], extent, 0} of the coordinate
                                       // type when the library computes the root box
    unsigned char h_rootbox[64] = {0};
    bool fixup_pending = false;
    Buf<int32_t> fix_large_list;
    Buf<SegSortFlags> fix_flags;
    SegSortFlags h_fix{};
    Buf<int64_t> wprefix;
    Buf<uint64_t> src_bits;            // separate targets only: "is a source" per tree-order
    Buf<int32_t> src_before;           // position, and the sources before every 64-bit word
    Buf<int32_t> srcntgt_target_ids;   // [ntargets]

    Buf<int32_t> box_start, box_count, box_parent, box_nonchild, box_child;
    Buf<uint8_t> box_level, box_haschild;
    Buf<unsigned char> centers;        // [cap][D] of coord type
    Buf<unsigned char> packed;         // [N][D] interleaved input coordinates (srcntgt order)

    std::vector<std::pair<const char *, hipEvent_t>> events;
    bool built = false;
};

void bt_free_tree_state(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    if (ctx->tree) {
        // a read queued into this state (fixup_launch) must not outlive it
        if (ctx->tree->fixup_pending) (void) bt::sync_stream(ctx);
        for (auto &e : ctx->tree->events) (void) hipEventDestroy(e.second);
        delete ctx->tree;
        ctx->tree = nullptr;
    }
}

namespace {

int mark(bt_context *ctx, TreeState *st, const char *name)
{
    host_trace(name);
    if (!ctx->stage_timing) return BT_OK;
    hipEvent_t e;
    BT_HIP_CHECK(hipEventCreate(&e));
    BT_HIP_CHECK(hipEventRecord(e, ctx->stream));
    st->events.push_back({name, e});
    return BT_OK;
}

template <class U>
int grow(bt_context *ctx, Buf<U> &buf, int64_t old_n, int64_t new_n, bool zero)
{
    Buf<U> nb;
    BT_CHECK(nb.alloc(ctx->pool, new_n));
    if (zero) BT_HIP_CHECK(hipMemsetAsync(nb.get(), 0, (size_t) new_n * sizeof(U), ctx->stream));
    if (old_n > 0 && buf.get())
        BT_HIP_CHECK(hipMemcpyAsync(nb.get(), buf.get(), (size_t) old_n * sizeof(U),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    buf.swap(nb);
    return BT_OK;
}

int ensure_box_capacity(bt_context *ctx, TreeState *st, int64_t need, size_t coord_size)
{
    if (need <= st->cap) return BT_OK;
    int64_t nc = std::max<int64_t>(st->cap * 2, 1024);
    while (nc < need) nc *= 2;
    const int64_t old = st->nboxes;
    BT_CHECK(grow(ctx, st->box_start, old, nc, false));
    BT_CHECK(grow(ctx, st->box_count, old, nc, false));
    BT_CHECK(grow(ctx, st->box_parent, old, nc, false));
    // (every field of a box is written when the box is created: no zero fill)
    BT_CHECK(grow(ctx, st->box_nonchild, old, nc, false));
    BT_CHECK(grow(ctx, st->box_child, old * st->C, nc * st->C, false));
    BT_CHECK(grow(ctx, st->box_level, old, nc, false));
    BT_CHECK(grow(ctx, st->box_haschild, old, nc, false));
    BT_CHECK(grow(ctx, st->centers, old * st->D * (int64_t) coord_size,
                  nc * st->D * (int64_t) coord_size, false));
    st->cap = nc;
    return BT_OK;
}


// ---------------------------------------------------------------------------
// kind = "adaptive-level-restricted" (tree_build.py:606-611, 1125-1224;
// tree_build_kernels.py:825-972).  The reference interleaves three things per
// level-loop trip: the regular split that creates the new level, the split of
// the boxes flagged in the previous trip ("force split", children appended to
// the END of their level), and an upward pass that flags leaves with a
// neighbouring leaf two levels deeper.  Box numbers therefore depend on the trip
// in which a box was created.  Here boxes are kept in creation order while the
// loop runs (no renumbering); the final numbers are the stable sort of the
// creation order by level, which is exactly what the reference's
// order-preserving renumberings and the final gap/empty-box removal produce.
// Children of every split are all created (empty ones matter for the balance)
// and empty boxes are dropped at the end unless skip_prune.
// ---------------------------------------------------------------------------

// LEVEL_RESTRICT_TPL (tbk:825-913): one thread per box; only leaves of
// `upper_level` do anything.
template <class T, int D>
__global__ __launch_bounds__(WALK_THREADS) void lr_pass_kernel(int32_t nboxes, int upper_level,
        T root_extent, const uint8_t *box_level, const uint8_t *box_haschild,
        const int32_t *box_child, const T *centers, int32_t *force_split, int32_t *have_split)
{
    constexpr int C = 1 << D;
    const int32_t box_id = blockIdx.x * WALK_THREADS + threadIdx.x;
    if (box_id >= nboxes) return;
    if ((int) box_level[box_id] != upper_level || box_haschild[box_id]) return;
    T bc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) bc[d] = centers[(int64_t) box_id * D + d];
    Walk w(s_walk_lds + threadIdx.x);
    w.init(0);
    while (w.go) {
        const int32_t child = box_child[(int64_t) w.parent * C + w.mnr];
        if (child) {
            const int child_level = w.size + 1;
            bool is_adjacent = false;
            if (child != box_id) {
                T cc[D];
#pragma unroll
                for (int d = 0; d < D; ++d) cc[d] = centers[(int64_t) child * D + d];
                is_adjacent = adj<T, D>(root_extent, cc, child_level, bc, upper_level);
            }
            if (is_adjacent) {
                if (box_haschild[child]) {
                    if (child_level <= 1 + upper_level) { w.push(child); continue; }
                } else if (child_level == 2 + upper_level
                           || (child_level == 1 + upper_level && force_split[child])) {
                    force_split[box_id] = 1;
                    __hip_atomic_store(have_split, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        w.template advance<C>();
    }
}

// Morton path of the boxes [b0, b1) created by the last split (raw numbering):
// parent's path, then the child slot
template <int D>
__global__ __launch_bounds__(256) void lr_paths_kernel(int32_t b0, int32_t b1,
        const int32_t *box_parent, const int32_t *box_child, uint64_t *paths)
{
    constexpr int C = 1 << D;
    const int32_t b = b0 + blockIdx.x * 256 + threadIdx.x;
    if (b >= b1) return;
    const int32_t p = box_parent[b];
    int slot = 0;
#pragma unroll
    for (int m = 0; m < C; ++m)
        if (box_child[(int64_t) p * C + m] == b) slot = m;
    paths[b] = (paths[p] << D) | (uint64_t) slot;
}

// The level-restriction pass turned around: instead of every leaf of `upper_level`
// walking the tree from the root for a finer neighbour (lr_pass_kernel), every
// SOURCE -- a leaf two levels finer, or a leaf one level finer that is already
// flagged -- looks up the <= 3^d boxes of `upper_level` around it by descending along
// their Morton paths, and flags those that are leaves and pass the reference's own
// float adjacency test.  Touching boxes are exactly the integer neighbours, so the
// flagged set is the same (tbk:880-900).
template <class T, int D>
__global__ __launch_bounds__(256) void lr_mark_kernel(int32_t nboxes, int src_level,
        int upper_level, int require_flag, T root_extent, const uint8_t *box_level,
        const uint8_t *box_haschild, const int32_t *box_child, const T *centers,
        const uint64_t *paths, int32_t *force_split, int32_t *have_split)
{
    constexpr int C = 1 << D;
    const int32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nboxes) return;
    if ((int) box_level[s] != src_level || box_haschild[s]) return;
    if (require_flag && !force_split[s]) return;
    const uint64_t path = paths[s];
    int64_t cell[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        int64_t v = 0;
        for (int bit = 0; bit < src_level; ++bit)
            v |= (int64_t) ((path >> (D * bit + (D - 1 - ax))) & 1) << bit;
        cell[ax] = v >> (src_level - upper_level);          // ancestor cell at upper_level
    }
    T sc[D];
#pragma unroll
    for (int ax = 0; ax < D; ++ax) sc[ax] = centers[(int64_t) s * D + ax];
    const int64_t ncell = (int64_t) 1 << upper_level;
    constexpr int NOFF = D == 1 ? 3 : D == 2 ? 9 : 27;
    for (int o = 0; o < NOFF; ++o) {
        int64_t nb[D];
        bool inside = true;
        int oo = o;
#pragma unroll
        for (int ax = 0; ax < D; ++ax) {
            nb[ax] = cell[ax] + (oo % 3) - 1;
            oo /= 3;
            inside = inside && nb[ax] >= 0 && nb[ax] < ncell;
        }
        if (!inside) continue;
        // descend from the root along the neighbour cell's digits
        int32_t cur = 0;
        int lev = 0;
        for (; lev < upper_level; ++lev) {
            int digit = 0;
#pragma unroll
            for (int ax = 0; ax < D; ++ax)
                digit |= (int) ((nb[ax] >> (upper_level - 1 - lev)) & 1) << (D - 1 - ax);
            const int32_t ch = box_child[(int64_t) cur * C + digit];
            if (!ch) break;
            cur = ch;
        }
        if (lev != upper_level || box_haschild[cur] || cur == s) continue;
        T tc[D];
#pragma unroll
        for (int ax = 0; ax < D; ++ax) tc[ax] = centers[(int64_t) cur * D + ax];
        if (adj<T, D>(root_extent, sc, src_level, tc, upper_level)) {
            force_split[cur] = 1;
            __hip_atomic_store(have_split, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

struct NonZeroI32 {
    const int32_t *f;
    __device__ int32_t operator()(int64_t i) const { return f[i] != 0; }
};

__global__ __launch_bounds__(256) void lr_compact_flagged_kernel(int32_t n, const int32_t *flag,
        const int32_t *pos, const uint8_t *box_level, uint32_t *levels_out, uint32_t *ids_out)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag[i]) return;
    levels_out[pos[i]] = box_level[i];
    ids_out[pos[i]] = (uint32_t) i;
}

__global__ __launch_bounds__(256) void lr_mark_haschild_kernel(int32_t n, const uint32_t *list,
                                                               uint8_t *box_haschild)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) box_haschild[list[i]] = 1;
}

__global__ __launch_bounds__(256) void lr_levels_kernel(int32_t n, const uint8_t *box_level,
                                                        uint32_t *keys)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = box_level[i];
}

struct LrKeep {
    const uint32_t *order;
    const int32_t *box_count;
    int keep_empty;
    __device__ int32_t operator()(int64_t i) const
    {
        return (keep_empty || box_count[order[i]] > 0) ? 1 : 0;
    }
};

__global__ __launch_bounds__(256) void lr_final_ids_kernel(int32_t n, LrKeep keep, const int32_t *pos,
        int32_t *final_of_raw)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t raw = keep.order[i];
    const bool k = keep(i) != 0;
    final_of_raw[raw] = k ? pos[i] : 0;     // pruned boxes map to 0 (tree_build.py:1337-1340)
}

// kept boxes per level: the boxes are sorted by level, so a level is a range of the
// sorted order and its count a difference of the keep-scan (one thread per level; a
// per-box atomicAdd on a dozen counters serialises: 35 ms for 3*10^6 boxes)
__global__ void lr_level_counts_kernel(int32_t n, const uint32_t *sorted_levels, const int32_t *pos,
                                       int32_t *level_counts)
{
    const uint32_t l = threadIdx.x;
    if (l >= BT_MAX_LEVELS) return;
    auto lower = [&](uint32_t key) {
        int32_t lo = 0, hi = n;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (sorted_levels[mid] < key) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    level_counts[l] = pos[lower(l + 1)] - pos[lower(l)];
}

template <class T, int D>
struct LrArrays {
    int32_t *start, *count, *parent, *nonchild, *child;
    uint8_t *level, *haschild;
    T *centers;
};

template <class T, int D>
__global__ __launch_bounds__(256) void lr_renumber_kernel(int32_t n, LrKeep keep, const int32_t *pos,
        const int32_t *final_of_raw, LrArrays<T, D> in, LrArrays<T, D> out)
{
    constexpr int C = 1 << D;
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep(i)) return;
    const uint32_t raw = keep.order[i];
    const int32_t f = pos[i];
    out.start[f] = in.start[raw];
    out.count[f] = in.count[raw];
    out.parent[f] = final_of_raw[in.parent[raw]];
    out.nonchild[f] = in.nonchild[raw];
    out.level[f] = in.level[raw];
    out.haschild[f] = in.haschild[raw];
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = in.child[(int64_t) raw * C + m];
        out.child[(int64_t) f * C + m] = c ? final_of_raw[c] : 0;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) out.centers[(int64_t) f * D + d] = in.centers[(int64_t) raw * D + d];
}

template <class T, int D>
int lr_build_boxes(bt_context *ctx, TreeState *st, const uint64_t *keys, const uint64_t *keys2, int L2,
                   const std::function<int(int64_t, int64_t)> &rekey_level)
{
    constexpr int C = 1 << D;
    const bt_tree_params &p = st->p;
    const bool EXT = st->have_extent;
    Buf<LevelFlags> d_flags;
    BT_CHECK(d_flags.alloc(ctx->pool, 1));
    Buf<int32_t> d_have;
    BT_CHECK(d_have.alloc(ctx->pool, 1));
    bool keys2_ready = false;

    auto make_args = [&](BuildArgs &a) {
        a = BuildArgs{};
        a.keys = keys;
        a.wprefix = st->wprefix.get();
        a.box_start = st->box_start.get(); a.box_count = st->box_count.get();
        a.box_parent = st->box_parent.get(); a.box_nonchild = st->box_nonchild.get();
        a.box_child = st->box_child.get();
        a.box_level = st->box_level.get(); a.box_haschild = st->box_haschild.get();
        a.flags = d_flags.get(); a.status = ctx->d_status;
        a.max_weight = p.max_leaf_refine_weight;
        a.L = st->L; a.capbits = st->capbits;
        a.adaptive = 1;
        a.keep_empty = 1;
        // below level L: the continuation key, once it has been made (rekey_level)
        a.keys2 = keys2_ready ? keys2 : nullptr;
        a.L1k = st->L; a.L2k = L2;
        a.can_continue = (L2 > 0 && !keys2_ready) ? 1 : 0;
    };

    // splits the boxes [b0, b0+n) of one level, or the listed boxes (forced)
    auto split = [&](int level, int64_t b0, int64_t n, const uint32_t *list, int *total_new,
                     int *oversize) -> int {
        Buf<int32_t> bounds, nnew, offsets;
        BT_CHECK(bounds.alloc(ctx->pool, n * (C + 1)));
        BT_CHECK(nnew.alloc(ctx->pool, n));
        BT_CHECK(offsets.alloc(ctx->pool, n));
        BT_HIP_CHECK(hipMemsetAsync(d_flags.get(), 0, sizeof(LevelFlags), ctx->stream));
        BuildArgs a;
        make_args(a);
        a.bounds = bounds.get(); a.nnew = nnew.get(); a.offsets = offsets.get();
        a.level = level; a.b0 = (int) b0; a.nprev = (int) n;
        a.parent_list = (const int32_t *) list;
        const unsigned blocks = (unsigned) div_up(n * C, 256);
        if (EXT) count_children_kernel<D, true><<<blocks, 256, 0, ctx->stream>>>(a);
        else count_children_kernel<D, false><<<blocks, 256, 0, ctx->stream>>>(a);
        ScanNnew sn{nnew.get()};
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, sn, n, offsets.get(),
                                                          &d_flags.get()->total_new)));
        LevelFlags hf;
        BT_CHECK(bt::d2h(ctx, &hf, d_flags.get(), sizeof(hf)));
        BT_CHECK(bt::sync_stream(ctx));
        if (hf.need_more && !keys2_ready) {
            // a box at the deepest level of the key has to split: continuation keys for the
            // boxes of that level, then the same boxes once more
            BT_CHECK(rekey_level(0, st->nboxes));
            keys2_ready = true;
            BT_HIP_CHECK(hipMemsetAsync(d_flags.get(), 0, sizeof(LevelFlags), ctx->stream));
            make_args(a);
            a.bounds = bounds.get(); a.nnew = nnew.get(); a.offsets = offsets.get();
            a.level = level; a.b0 = (int) b0; a.nprev = (int) n;
            a.parent_list = (const int32_t *) list;
            if (EXT) count_children_kernel<D, true><<<blocks, 256, 0, ctx->stream>>>(a);
            else count_children_kernel<D, false><<<blocks, 256, 0, ctx->stream>>>(a);
            BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, sn, n, offsets.get(),
                                                              &d_flags.get()->total_new)));
            BT_CHECK(bt::d2h(ctx, &hf, d_flags.get(), sizeof(hf)));
            BT_CHECK(bt::sync_stream(ctx));
        }
        *total_new = hf.total_new;
        *oversize = hf.have_oversize;
        if (hf.total_new == 0) return BT_OK;
        BT_CHECK(ensure_box_capacity(ctx, st, st->nboxes + hf.total_new, sizeof(T)));
        make_args(a);      // buffers may have moved
        a.bounds = bounds.get(); a.nnew = nnew.get(); a.offsets = offsets.get();
        a.level = level; a.b0 = (int) b0; a.nprev = (int) n;
        a.parent_list = (const int32_t *) list;
        a.new_level_start = (int) st->nboxes;
        write_children_kernel<T, D><<<blocks, 256, 0, ctx->stream>>>(a, (T *) st->centers.get(),
                                                                     (T) p.root_extent);
        BT_HIP_CHECK(hipGetLastError());
        const int64_t first_new = st->nboxes;
        st->nboxes += hf.total_new;
        // boxes of the key's deepest level that appear after the continuation keys were made
        // (children of leaves the restriction splits late) get theirs now
        if (keys2_ready && (list != nullptr || level == st->L))
            BT_CHECK(rekey_level(first_new, hf.total_new));
        return BT_OK;
    };

    Buf<uint64_t> box_path;
    int64_t path_cap = 0, paths_done = 1;
    auto update_paths = [&]() -> int {
        if (st->nboxes > path_cap) {
            Buf<uint64_t> nbuf;
            const int64_t ncap = std::max<int64_t>(st->cap, st->nboxes);
            BT_CHECK(nbuf.alloc(ctx->pool, ncap));
            if (path_cap > 0)
                BT_HIP_CHECK(hipMemcpyAsync(nbuf.get(), box_path.get(), (size_t) paths_done * 8,
                                            hipMemcpyDeviceToDevice, ctx->stream));
            else
                BT_HIP_CHECK(hipMemsetAsync(nbuf.get(), 0, 8, ctx->stream));   // root: path 0
            box_path.swap(nbuf);
            path_cap = ncap;
        }
        return BT_OK;
    };
    // paths of the boxes [from, nboxes): parents are older boxes, one launch per split
    auto extend_paths = [&](int64_t from) -> int {
        BT_CHECK(update_paths());
        if (st->nboxes > from)
            lr_paths_kernel<D><<<(unsigned) div_up(st->nboxes - from, 256), 256, 0, ctx->stream>>>(
                (int32_t) from, (int32_t) st->nboxes, st->box_parent.get(), st->box_child.get(),
                box_path.get());
        paths_done = st->nboxes;
        return BT_OK;
    };

    int level = 1;
    bool final_iteration = false;               // tree_build.py:695
    int64_t reg_b0 = 0, reg_n = 1;              // the regular boxes of level-1
    Buf<uint32_t> fl_levels, fl_ids, fl_levels_b, fl_ids_b;
    const uint32_t *flagged = nullptr;
    int64_t nflagged = 0;
    while (true) {
        int total_new = 0, oversize = 0;
        if (!final_iteration) {
            const int64_t first_new = st->nboxes;
            BT_CHECK(split(level, reg_b0, reg_n, nullptr, &total_new, &oversize));
            if (total_new == 0) {
                // tree_build.py:1016-1025: nothing new on this level (extents); the split
                // scan has already marked the flagged boxes as parents (tbk:596-598)
                if (nflagged > 0)
                    lr_mark_haschild_kernel<<<(unsigned) div_up(nflagged, 256), 256, 0, ctx->stream>>>(
                        (int32_t) nflagged, flagged, st->box_haschild.get());
                break;
            }
            reg_b0 = first_new;
            reg_n = total_new;
            BT_CHECK(extend_paths(first_new));
        }
        if (nflagged > 0) {
            int forced_new = 0, dummy = 0;
            const int64_t first_forced = st->nboxes;
            BT_CHECK(split(0, 0, nflagged, flagged, &forced_new, &dummy));
            BT_CHECK(extend_paths(first_forced));
            nflagged = 0;
        }
        if (final_iteration) break;             // :1127-1143

        // upward pass: :1145-1224
        const int32_t nb = (int32_t) st->nboxes;
        Buf<int32_t> force;
        BT_CHECK(force.alloc(ctx->pool, nb));
        BT_HIP_CHECK(hipMemsetAsync(force.get(), 0, (size_t) nb * 4, ctx->stream));
        bool did_upper_level_split = false;
        const size_t lds = (size_t) (level + 2) * WALK_THREADS * 4;
        for (int upper_level = level - 2; upper_level >= 1; --upper_level) {
            BT_HIP_CHECK(hipMemsetAsync(d_have.get(), 0, 4, ctx->stream));
            // (the look-up form descends along 64-bit Morton paths: trees that deep -- below
            // the first key -- take the reference's walk)
            if (D * level > 63) {
                lr_pass_kernel<T, D><<<(unsigned) div_up(nb, WALK_THREADS), WALK_THREADS, lds, ctx->stream>>>(
                    nb, upper_level, (T) p.root_extent, st->box_level.get(), st->box_haschild.get(),
                    st->box_child.get(), (const T *) st->centers.get(), force.get(), d_have.get());
            } else {
                // leaves two levels finer, then flagged leaves one level finer (flagged
                // by the previous, finer step of this pass)
                for (int k = 2; k >= 1; --k)
                    lr_mark_kernel<T, D><<<(unsigned) div_up(nb, 256), 256, 0, ctx->stream>>>(
                        nb, upper_level + k, upper_level, k == 1, (T) p.root_extent,
                        st->box_level.get(), st->box_haschild.get(), st->box_child.get(),
                        (const T *) st->centers.get(), box_path.get(), force.get(), d_have.get());
            }
            int32_t have = 0;
            BT_CHECK(bt::d2h(ctx, &have, d_have.get(), 4));
            BT_CHECK(bt::sync_stream(ctx));
            if (!have) break;                   // :1201-1202
            did_upper_level_split = true;
        }
        if (did_upper_level_split) {
            // flagged boxes ordered by (level, creation order): the order in which the
            // split scan hands out their children's numbers (tbk:514-640)
            Buf<int32_t> pos;
            BT_CHECK(pos.alloc(ctx->pool, (int64_t) nb + 1));
            BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, NonZeroI32{force.get()}, nb,
                                                              pos.get(), (int32_t *) nullptr, true)));
            int32_t nf = 0;
            BT_CHECK(bt::d2h(ctx, &nf, pos.get() + nb, 4));
            BT_CHECK(bt::sync_stream(ctx));
            BT_CHECK(fl_levels.alloc(ctx->pool, nf));
            BT_CHECK(fl_ids.alloc(ctx->pool, nf));
            BT_CHECK(fl_levels_b.alloc(ctx->pool, nf));
            BT_CHECK(fl_ids_b.alloc(ctx->pool, nf));
            lr_compact_flagged_kernel<<<(unsigned) div_up(nb, 256), 256, 0, ctx->stream>>>(
                nb, force.get(), pos.get(), st->box_level.get(), fl_levels.get(), fl_ids.get());
            bool in_b = false;
            BT_CHECK(radix_sort_pairs<uint32_t>(ctx, fl_levels.get(), fl_ids.get(), fl_levels_b.get(),
                                                fl_ids_b.get(), nf, 0, 8, false, &in_b));
            flagged = in_b ? fl_ids_b.get() : fl_ids.get();
            nflagged = nf;
        }
        if (!oversize && did_upper_level_split) {   // :1216-1224
            final_iteration = true;
            level += 1;
            continue;
        }
        if (!oversize) break;                   // :1228-1230
        level += 1;
        if (level > st->L + L2 + 1) break;      // defensive; max_levels flag is set on device
    }
    BT_CHECK(check_status(ctx));

    // ---- final numbers: creation order -> (level, creation order), empty boxes out ----
    const int32_t nraw = (int32_t) st->nboxes;
    Buf<uint32_t> ka, kb, va, vb;
    BT_CHECK(ka.alloc(ctx->pool, nraw));
    BT_CHECK(kb.alloc(ctx->pool, nraw));
    BT_CHECK(va.alloc(ctx->pool, nraw));
    BT_CHECK(vb.alloc(ctx->pool, nraw));
    lr_levels_kernel<<<(unsigned) div_up(nraw, 256), 256, 0, ctx->stream>>>(nraw, st->box_level.get(),
                                                                           ka.get());
    bool in_b = false;
    BT_CHECK(radix_sort_pairs<uint32_t>(ctx, ka.get(), va.get(), kb.get(), vb.get(), nraw, 0, 8, true,
                                        &in_b));
    const uint32_t *order = in_b ? vb.get() : va.get();
    LrKeep keep{order, st->box_count.get(), p.skip_prune ? 1 : 0};
    Buf<int32_t> pos, final_of_raw, level_counts;
    BT_CHECK(pos.alloc(ctx->pool, (int64_t) nraw + 1));
    BT_CHECK(final_of_raw.alloc(ctx->pool, nraw));
    BT_CHECK(level_counts.alloc(ctx->pool, BT_MAX_LEVELS));
    BT_HIP_CHECK(hipMemsetAsync(level_counts.get(), 0, BT_MAX_LEVELS * 4, ctx->stream));
    BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, keep, nraw, pos.get(), (int32_t *) nullptr,
                                                      true)));
    lr_final_ids_kernel<<<(unsigned) div_up(nraw, 256), 256, 0, ctx->stream>>>(
        nraw, keep, pos.get(), final_of_raw.get());
    lr_level_counts_kernel<<<1, BT_MAX_LEVELS, 0, ctx->stream>>>(nraw, in_b ? kb.get() : ka.get(),
                                                                pos.get(), level_counts.get());
    int32_t h_counts[BT_MAX_LEVELS];
    int32_t nfinal = 0;
    BT_CHECK(bt::d2h(ctx, h_counts, level_counts.get(), sizeof(h_counts)));
    BT_CHECK(bt::d2h(ctx, &nfinal, pos.get() + nraw, 4));
    BT_CHECK(bt::sync_stream(ctx));

    Buf<int32_t> n_start, n_count, n_parent, n_nonchild, n_child;
    Buf<uint8_t> n_level, n_haschild;
    Buf<unsigned char> n_centers;
    const int64_t cap = std::max<int64_t>(nfinal, 1);
    BT_CHECK(n_start.alloc(ctx->pool, cap));
    BT_CHECK(n_count.alloc(ctx->pool, cap));
    BT_CHECK(n_parent.alloc(ctx->pool, cap));
    BT_CHECK(n_nonchild.alloc(ctx->pool, cap));
    BT_CHECK(n_child.alloc(ctx->pool, cap * C));
    BT_CHECK(n_level.alloc(ctx->pool, cap));
    BT_CHECK(n_haschild.alloc(ctx->pool, cap));
    BT_CHECK(n_centers.alloc(ctx->pool, cap * D * (int64_t) sizeof(T)));
    LrArrays<T, D> in{st->box_start.get(), st->box_count.get(), st->box_parent.get(),
                      st->box_nonchild.get(), st->box_child.get(), st->box_level.get(),
                      st->box_haschild.get(), (T *) st->centers.get()};
    LrArrays<T, D> out{n_start.get(), n_count.get(), n_parent.get(), n_nonchild.get(), n_child.get(),
                       n_level.get(), n_haschild.get(), (T *) n_centers.get()};
    lr_renumber_kernel<T, D><<<(unsigned) div_up(nraw, 256), 256, 0, ctx->stream>>>(
        nraw, keep, pos.get(), final_of_raw.get(), in, out);
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::sync_stream(ctx));
    st->box_start.swap(n_start); st->box_count.swap(n_count); st->box_parent.swap(n_parent);
    st->box_nonchild.swap(n_nonchild); st->box_child.swap(n_child); st->box_level.swap(n_level);
    st->box_haschild.swap(n_haschild); st->centers.swap(n_centers);
    st->nboxes = nfinal;
    st->cap = cap;
    st->level_start = {0};
    for (int l = 0; l < BT_MAX_LEVELS && h_counts[l] > 0; ++l)
        st->level_start.push_back(st->level_start.back() + h_counts[l]);
    return BT_OK;
}

// Within-box order fix-up (see the header): every leaf's ids, and every box's run of own
// particles, ascending -- in two halves.  fixup_launch queues the half-wave sort of the
// leaves' ids, then the workgroup sort of the runs of 65..SEG_BLOCK_MAX ids (their number is
// read on the device).  Only a build that can have longer runs -- particle extents (the
// particles stuck in a split box form one run), refine weights (zero-weight particles),
// more than SEG_BLOCK_MAX particles per leaf, level restriction -- needs the host to look
// at the flags: fixup_finish does, at the start of the export (the caller allocates its
// arrays in between, while the kernel runs), and takes the global route if it must
// (segment_key_kernel scatters "start of my run" to user order and a 32-bit radix sort of
// (run start, 0..N-1) yields the permutation).
int fixup_launch(bt_context *ctx, TreeState *st)
{
    const int64_t N = st->N;
    st->fixup_pending = false;
    if (N <= 1) { st->fixup_done = true; return BT_OK; }
    const bt_tree_params &p = st->p;
    const bool can_be_huge = st->have_extent || p.refine_weights != nullptr
        || p.max_leaf_refine_weight > SEG_BLOCK_MAX || p.kind != BT_KIND_ADAPTIVE;
    BT_CHECK(st->fix_large_list.alloc(ctx->pool, N / 64 + 1));
    // (when the host will not look at the flags they can live in the call's zeroed block;
    // otherwise they must survive until the export's call)
    SegSortFlags *zf = can_be_huge ? nullptr : (SegSortFlags *) bt::zero_alloc(ctx, sizeof(SegSortFlags));
    if (zf) {
        st->fix_flags.set_external(zf, 1);
    } else {
        BT_CHECK(st->fix_flags.alloc(ctx->pool, 1));
        BT_HIP_CHECK(hipMemsetAsync(st->fix_flags.get(), 0, sizeof(SegSortFlags), ctx->stream));
    }
    const unsigned wgrid = (unsigned) div_up(div_up(st->nboxes, SEG_BOXES_PER_HALF_WAVE) * 32, 256);
    if (st->pk && can_be_huge) {
        // runs of any length (particles stuck in a split box): the host may have to take the
        // global route, which works on the id array -- every run is unpacked into it, the
        // short ones (and the leaves among them ordered) by the wave kernel, the rest by a
        // kernel over the list it makes
        if (!st->have_extent) {
            // (refine weights etc. never come with packed keys; belt and braces)
            unpack_ids_kernel<<<(unsigned) div_up(N, 256), 256, 0, ctx->stream>>>(N, st->pk, st->pk_mask, st->ids);
            st->pk = nullptr;
        } else {
            Buf<int32_t> copy_list;
            BT_CHECK(copy_list.alloc(ctx->pool, N / 64 + 1));
            segment_unpack_sort_wave_kernel<<<wgrid, 256, 0, ctx->stream>>>(
                (int) st->nboxes, st->box_start.get(), st->box_count.get(), st->box_nonchild.get(),
                st->box_haschild.get(), st->ids, st->fix_large_list.get(), copy_list.get(),
                st->fix_flags.get(), st->pk, st->pk_mask);
            segment_unpack_block_kernel<<<(unsigned) std::min<int64_t>(ctx->num_cus * 4, N / 64 + 1), 256, 0,
                                          ctx->stream>>>(
                copy_list.get(), st->fix_flags.get(), st->box_start.get(), st->box_count.get(),
                st->box_nonchild.get(), st->box_haschild.get(), st->ids, st->pk, st->pk_mask);
            BT_HIP_CHECK(hipGetLastError());
            st->pk = nullptr;
            BT_CHECK(bt::d2h(ctx, &st->h_fix, st->fix_flags.get(), sizeof(SegSortFlags), /*persistent=*/true));
            st->fixup_pending = true;
            return BT_OK;
        }
    }
    if (st->pk) {
        segment_sort_wave_kernel<true><<<wgrid, 256, 0, ctx->stream>>>(
            (int) st->nboxes, st->box_start.get(), st->box_count.get(), st->box_haschild.get(),
            st->ids, st->fix_large_list.get(), st->fix_flags.get(), st->pk, st->pk_mask);
    } else {
        segment_sort_wave_kernel<false><<<wgrid, 256, 0, ctx->stream>>>(
            (int) st->nboxes, st->box_start.get(), st->box_count.get(), st->box_haschild.get(),
            st->ids, st->fix_large_list.get(), st->fix_flags.get(), nullptr, 0);
    }
    if (!can_be_huge) {
        const unsigned grid = (unsigned) std::min<int64_t>(ctx->num_cus * 4, N / 64 + 1);
        if (st->pk)
            segment_sort_block_kernel<true><<<grid, 256, 0, ctx->stream>>>(
                st->fix_large_list.get(), st->fix_flags.get(), 1, ctx->d_status, st->box_start.get(),
                st->box_count.get(), st->ids, st->pk, st->pk_mask);
        else
            segment_sort_block_kernel<false><<<grid, 256, 0, ctx->stream>>>(
                st->fix_large_list.get(), st->fix_flags.get(), 1, ctx->d_status, st->box_start.get(),
                st->box_count.get(), st->ids, nullptr, 0);
        st->pk = nullptr;
        BT_HIP_CHECK(hipGetLastError());
        st->fix_large_list.reset();
        st->fix_flags.reset();
        st->fixup_done = true;
        return BT_OK;
    }
    BT_CHECK(bt::d2h(ctx, &st->h_fix, st->fix_flags.get(), sizeof(SegSortFlags), /*persistent=*/true));
    st->fixup_pending = true;
    return BT_OK;
}

int fixup_finish(bt_context *ctx, TreeState *st)
{
    if (!st->fixup_pending) return BT_OK;
    const int64_t N = st->N;
    BT_CHECK(bt::sync_stream(ctx));
    st->fixup_pending = false;
    ctx->n_host_syncs++;
    if (!st->h_fix.has_huge) {
        if (st->h_fix.n_large > 0)
            segment_sort_block_kernel<false><<<(unsigned) std::min<int64_t>(st->h_fix.n_large, ctx->num_cus * 8),
                                               256, 0, ctx->stream>>>(
                st->fix_large_list.get(), st->fix_flags.get(), 0, ctx->d_status, st->box_start.get(),
                st->box_count.get(), st->ids, nullptr, 0);
        BT_HIP_CHECK(hipGetLastError());
    } else {
        if (!st->ids_other) {          // (packed keys: the second id buffer was never needed)
            BT_CHECK(st->ids_b.alloc(ctx->pool, N));
            st->ids_other = st->ids_b.get();
        }
        uint32_t *ids = st->ids, *ids_other = st->ids_other;
        Buf<uint32_t> fk_a, fk_b;
        BT_CHECK(fk_a.alloc(ctx->pool, N));
        BT_CHECK(fk_b.alloc(ctx->pool, N));
        segment_key_kernel<<<(unsigned) div_up(st->nboxes * 16, 256), 256, 0, ctx->stream>>>(
            (int) st->nboxes, st->box_start.get(), st->box_count.get(), st->box_nonchild.get(),
            st->box_haschild.get(), ids, fk_a.get());
        int bits = 1;
        while (((int64_t) 1 << bits) < N) ++bits;
        bool in_b = false;
        // values: reuse the two id buffers; `ids` is consumed by segment_key_kernel
        // (stream order) before the sort overwrites it.
        uint32_t *va = ids_other, *vb = ids;
        BT_CHECK(radix_sort_pairs<uint32_t>(ctx, fk_a.get(), va, fk_b.get(), vb, N, 0, bits,
                                            true, &in_b));
        st->ids = in_b ? vb : va;
        st->ids_other = in_b ? va : vb;
        BT_CHECK(check_status(ctx));
    }
    st->fix_large_list.reset();
    st->fix_flags.reset();
    st->fixup_done = true;
    return BT_OK;
}

int run_fixup(bt_context *ctx, TreeState *st)
{
    BT_CHECK(fixup_launch(ctx, st));
    return fixup_finish(ctx, st);
}

template <class T, int D>
int tree_build_impl(bt_context *ctx, TreeState *st, bt_tree_sizes *out)
{
    constexpr int C = 1 << D;
    const bt_tree_params &p = st->p;
    const int64_t N = st->N;
    const bool EXT = st->have_extent;

    BT_CHECK(mark(ctx, st, "start"));
    int point_skip_raw = 0;

    // ---- root box on the device (compute_root_box) --------------------------------
    if (p.compute_root_box) {
        const int64_t ns = st->nsources, nt = st->sat ? 0 : st->ntargets;
        auto nblocks = [&](int64_t n) {
            return n > 0 ? std::min<int64_t>(div_up(n, BBOX_THREADS * 8), (int64_t) ctx->num_cus * 8) : 0;
        };
        const int64_t bs = nblocks(ns), bt_ = nblocks(nt);
        Buf<T> partial;
        BT_CHECK(partial.alloc(ctx->pool, 2 * (bs + bt_) * D + 2));
        if (bs > 0) {
            BboxAxes<T> axs{};
            for (int ax = 0; ax < D; ++ax) axs.x[ax] = (const T *) p.sources[ax];
            bbox_axes_kernel<T><<<dim3((unsigned) bs, D), BBOX_THREADS, 0, ctx->stream>>>(
                axs, ns, partial.get());
        }
        if (bt_ > 0) {
            BboxAxes<T> axs{};
            for (int ax = 0; ax < D; ++ax) axs.x[ax] = (const T *) p.targets[ax];
            bbox_axes_kernel<T><<<dim3((unsigned) bt_, D), BBOX_THREADS, 0, ctx->stream>>>(
                axs, nt, partial.get() + 2 * bs * D);
        }
        BT_CHECK(st->rootbox.alloc(ctx->pool, 8 * sizeof(T)));
        root_box_kernel<T, D><<<1, 256, 0, ctx->stream>>>(
            partial.get(), (int) bs, (int) bt_, (T) (1.0 + p.root_extent_stretch), (T *) st->rootbox.get());
        BT_HIP_CHECK(hipGetLastError());
        // the host copy arrives with the level loop's first wait
        BT_CHECK(bt::d2h(ctx, st->h_rootbox, st->rootbox.get(), 8 * sizeof(T), /*persistent=*/true));
    }

    // ---- key generator arguments (also the depth probe's and the re-keying kernel's) -------
    KeygenArgs<T, D> ka{};
    for (int ax = 0; ax < D; ++ax) {
        ka.src[ax] = (const T *) p.sources[ax];
        ka.tgt[ax] = (const T *) p.targets[ax];
        ka.bbox_min[ax] = (T) p.bbox_min[ax];
        ka.bbox_max[ax] = (T) p.bbox_max[ax];
    }
    ka.rootbox = (const T *) st->rootbox.get();
    ka.src_radii = (const T *) p.source_radii;
    ka.tgt_radii = (const T *) p.target_radii;
    ka.nsources = st->nsources;
    ka.n = N;
    ka.src_stride = p.source_stride > 0 ? p.source_stride : 1;
    ka.tgt_stride = p.target_stride > 0 ? p.target_stride : 1;
    ka.stick_out_factor = (T) p.stick_out_factor;
    ka.L = st->L;
    ka.norm = p.extent_norm;
    {
        // margin at level l: (stick_out_factor / 2) * extent * 2^-l; rounding of the
        // cell assignment, the centre and the limit: a few spacings of T at the
        // magnitude of the coordinates.  Require margin >= 128 spacings.
        double scale = std::fabs(p.root_extent);
        for (int ax = 0; ax < D; ++ax)
            scale = std::max(scale, std::max(std::fabs(p.bbox_min[ax]), std::fabs(p.bbox_max[ax])));
        const double eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
        const double ratio = p.stick_out_factor * p.root_extent / (256.0 * eps * scale);
        int skip = 0;
        // (extents only; with compute_root_box there are none and the root box is not
        // known here)
        if (EXT && ratio > 1.0) skip = (int) std::floor(std::log2(ratio));
        point_skip_raw = std::max(0, skip);
        ka.point_skip_levels = std::max(0, std::min(skip, st->L));
    }

    // ---- how deep will the tree get?  From N alone: points on a (D-1)-dimensional set
    // (surfaces are the deep case in practice) fill 2^(D-1) children per split, plus two
    // levels of slack.  Large builds look first (depth probe): the densest level-k cell of a
    // sample of the particles -- of the global histogram for a sharded build -- and the
    // dimension of the occupied cells; the smaller of the two estimates counts.  A tree
    // that turns out deeper still builds (the level loop sorts the remaining bits when it
    // gets there), it only costs those extra passes.
    int est_levels = st->L;
    // what packing depends on besides the depth
    int pk_idbits = 1;
    while (((int64_t) 1 << pk_idbits) < N) ++pk_idbits;
    const int lk_max = std::min(st->L, (64 - pk_idbits - st->capbits) / D);
    const char *e_lev = getenv("BT_PACKED_LEVELS");                // testing / tuning aids
    bool can_pack;
    {
        const char *e_off = getenv("BT_NO_PACKED_KEYS");
        can_pack = !(e_off && atoi(e_off)) && !p.refine_weights && p.kind == BT_KIND_ADAPTIVE
            && (EXT || p.max_leaf_refine_weight <= SEG_BLOCK_MAX) && N >= 2 && lk_max >= 1;
    }
    // levels the packed word holds for a depth estimate: the passes are whole digits, take
    // the levels they cover anyway
    auto packed_levels_for = [&](int est) {
        int lk = std::min(lk_max, std::max(est, 1));
        int rb = 8;
        const int np = bt::radix_sort_keys_plan(D * lk + st->capbits, &rb);
        lk = std::min(lk_max, std::max(lk, (np * rb - st->capbits) / D));
        if (e_lev && atoi(e_lev) > 0) lk = std::min(lk, atoi(e_lev));
        return lk;
    };
    bool packed = false;
    int pk_levels = 0;           // path levels in the packed word
    int pk_sorted = 0;           // ... of which the sort orders the first pk_sorted
    bool keys_queued = false;
    auto queue_keygen = [&]() -> int {
        BT_CHECK(st->keys_a.alloc(ctx->pool, N));
        BT_CHECK(st->keys_b.alloc(ctx->pool, N));
        BT_CHECK(st->ids_a.alloc(ctx->pool, N));
        if (!packed) BT_CHECK(st->ids_b.alloc(ctx->pool, N));
        keys_queued = true;
        if (N <= 0) return BT_OK;
        KeygenArgs<T, D> kg = ka;
        kg.pack_idbits = packed ? pk_idbits : 0;
        kg.pack_drop = packed ? D * (st->L - pk_levels) : 0;
        const unsigned blocks = (unsigned) div_up(N, 256);
        BT_CHECK(st->packed.alloc(ctx->pool, N * PackStride<D>::value * (int64_t) sizeof(T)));
        T *packed_coords = (T *) st->packed.get();
        if (EXT) keygen_kernel<T, D, true><<<blocks, 256, 0, ctx->stream>>>(kg, st->keys_a.get(), packed_coords);
        else keygen_kernel<T, D, false><<<blocks, 256, 0, ctx->stream>>>(kg, st->keys_a.get(), packed_coords);
        BT_HIP_CHECK(hipGetLastError());
        return BT_OK;
    };
    if (!p.refine_weights) {
        const double per_leaf = std::max(1.0, (double) N / std::max(1, p.max_leaf_refine_weight));
        const int fan = D > 1 ? D - 1 : 1;
        if (!p.top_cell_prefix) est_levels = (int) std::ceil(std::log2(per_leaf) / fan) + 2;
        est_levels = std::max(1, std::min(est_levels, st->L));
        const bool probe = p.kind != BT_KIND_ADAPTIVE_LEVEL_RESTRICTED
            && (p.top_cell_prefix || N >= ((int64_t) 1 << 20));
        if (probe) {
            int k = p.top_cell_prefix ? p.top_level : (D == 3 ? 4 : D == 2 ? 6 : 12);
            k = std::min(k, st->L);
            const int64_t ncells = (int64_t) 1 << (D * k);
            const int64_t nsample = p.top_cell_prefix ? 0 : std::min<int64_t>(N, (int64_t) 1 << 17);
            const int64_t stride = nsample > 0 ? N / nsample : 1;
            Buf<uint32_t> hist;
            BT_CHECK(hist.alloc(ctx->pool, ncells + 4));
            BT_HIP_CHECK(hipMemsetAsync(hist.get(), 0, (size_t) (ncells + 4) * 4, ctx->stream));
            if (p.top_cell_prefix)
                prefix_to_hist_kernel<<<(unsigned) div_up(ncells, 256), 256, 0, ctx->stream>>>(
                    p.top_cell_prefix, ncells, hist.get());
            else
                depth_probe_kernel<T, D><<<(unsigned) div_up(nsample, 256), 256, 0, ctx->stream>>>(
                    ka, nsample, stride, k, hist.get());
            depth_probe_reduce_kernel<<<(unsigned) std::min<int64_t>(div_up(ncells / C, 256), 256), 256, 0,
                                        ctx->stream>>>(hist.get(), ncells, k > 0 ? C : 1, hist.get() + ncells);
            BT_HIP_CHECK(hipGetLastError());
            uint32_t h_probe[3] = {0, 0, 0};
            BT_CHECK(bt::d2h(ctx, h_probe, hist.get() + ncells, sizeof(h_probe)));
            // If N alone already allows packing, the word's format does not depend on what
            // the probe finds -- only how many of its path bits get sorted does: the key
            // kernel is queued first and runs while the host waits for the probe.  (With
            // extents the cap sits between the path and the id, and sorting fewer path bits
            // would take two bit ranges: those builds wait for the probe first.)
            if (can_pack && !EXT && lk_max >= est_levels - 1) {
                packed = true;
                pk_levels = packed_levels_for(est_levels);
                BT_CHECK(queue_keygen());
            }
            BT_CHECK(bt::sync_stream(ctx));
            ctx->n_host_syncs++;
            // largest cell: the sample's count scaled up, with three standard deviations
            double cmax = (double) h_probe[0];
            if (nsample > 0) cmax = (cmax + 3.0 * std::sqrt(cmax)) * ((double) N / (double) (nsample));
            const double ratio = h_probe[2] > 0 ? (double) h_probe[1] / (double) h_probe[2] : 1.0;
            // dimension of the occupied set at this scale, never below a surface's
            int pfan = (int) std::floor(std::log2(std::max(1.0, ratio)) + 0.2);
            pfan = std::max(fan, std::min(D, pfan));
            const int est_probe = k + (int) std::ceil(std::log2(std::max(
                1.0, cmax / std::max(1, p.max_leaf_refine_weight))) / pfan) + 2;
            est_levels = std::min(est_levels, est_probe);
        }
        est_levels = std::max(1, std::min(est_levels, st->L));
    }
    // ---- packed keys (see rekey_full_kernel): path bits of Lk levels (and the cap) over the id --
    if (keys_queued) {
        pk_sorted = std::min(pk_levels, packed_levels_for(est_levels));
    } else {
        if (can_pack && lk_max >= est_levels - 1) {
            packed = true;
            pk_levels = pk_sorted = packed_levels_for(est_levels);
        }
        BT_CHECK(queue_keygen());
    }
    BT_CHECK(mark(ctx, st, "keygen"));

    // ---- sort ----------------------------------------------------------------
    const uint64_t *keys = st->keys_a.get();
    uint32_t *ids = st->ids_a.get();
    uint32_t *ids_other = st->ids_b.get();
    // Only the key bits of the levels the tree actually reaches need to be in
    // order: boxes of level l are cut by the top D*l bits, and the order inside a
    // leaf is fixed up by user id afterwards.  Five digit passes (40 bits: 13
    // levels in 3D, 20 in 2D) cover all but pathologically clustered inputs; if
    // the level loop gets deeper, the remaining bits are sorted then (below).
    // With extents the key ends in the stop level ("cap") of the particle, which decides
    // between particles of equal path: a box's own particles (path zeroed below the box,
    // smallest cap) must precede everything else in its range.  One digit pass over the
    // lowest key bits, then the passes over the top path bits, give the order (top path
    // bits, low path bits of the deepest level, cap): every particle that stops at a
    // level the sorted path bits reach has all-zero low path bits, so it still comes first
    // in its box, in cap order -- which is all the level kernels' searches rely on down
    // to that level.  6 passes instead of 8 at 10^8 + 10^7 particles.
    const int keybits = D * st->L + st->capbits;
    int sorted_high_bits = keybits;
    if (!p.refine_weights && keybits > 40 + (EXT ? 8 : 0)
            && p.kind != BT_KIND_ADAPTIVE_LEVEL_RESTRICTED) {
        // (depth estimate above)
        int bits = ((D * est_levels + 7) / 8) * 8;
        bits = std::max(40, bits);
        if (bits < keybits) sorted_high_bits = bits;
    }
    uint64_t *keys_cur = st->keys_a.get(), *keys_oth = st->keys_b.get();
    if (packed) {
        bool in_b = false;
        BT_CHECK(bt::radix_sort_keys(ctx, keys_cur, keys_oth, N,
                                     pk_idbits + (pk_sorted < pk_levels ? D * (pk_levels - pk_sorted) : 0),
                                     pk_idbits + st->capbits + D * pk_levels, &in_b));
        if (in_b) std::swap(keys_cur, keys_oth);
        keys = keys_cur;
        ids = st->ids_a.get(); ids_other = nullptr;       // filled by the fix-up
        sorted_high_bits = D * pk_sorted;
    } else if (N > 0) {
        uint64_t *ka = st->keys_a.get(), *kb = st->keys_b.get();
        uint32_t *ia = st->ids_a.get(), *ib = st->ids_b.get();
        bool identity = true;
        if (EXT && sorted_high_bits < keybits && keybits - sorted_high_bits > 8) {   // more than the one digit
            // the cap digit first (see above); if the two ranges touch, one sort does both
            bool in_b = false;
            BT_CHECK(radix_sort_pairs<uint64_t>(ctx, ka, ia, kb, ib, N, 0, st->capbits, true, &in_b));
            if (in_b) { std::swap(ka, kb); std::swap(ia, ib); }
            identity = false;
        } else if (EXT) {
            sorted_high_bits = keybits;
        }
        bool in_b = false;
        BT_CHECK(radix_sort_pairs<uint64_t>(ctx, ka, ia, kb, ib, N,
                                            keybits - sorted_high_bits, keybits, identity, &in_b));
        if (in_b) { std::swap(ka, kb); std::swap(ia, ib); }
        keys = ka; ids = ia; ids_other = ib;
        keys_cur = ka; keys_oth = kb;
    }
    BT_CHECK(mark(ctx, st, "sort"));

    // ---- refine-weight prefix sums (only with explicit weights) ---------------
    if (p.refine_weights && N > 0) {
        BT_CHECK(st->wprefix.alloc(ctx->pool, N + 1));
        GatherWeight gw{p.refine_weights, ids};
        BT_CHECK((device_exclusive_scan<int64_t, int64_t>(ctx, gw, N, st->wprefix.get(),
                                                          (int64_t *) nullptr, true)));
    }

    // ---- boxes, level by level -------------------------------------------------
    // Capacity up front (twice the number of boxes a tree with full leaves would have,
    // times the fan-out; measured trees need 2-3.5 N/mpb): the level kernels check it,
    // and an overflow grows the arrays and repeats the level.
    st->nboxes = 0;
    st->cap = 0;
    {
        const double per_leaf = (double) N / std::max(1, p.max_leaf_refine_weight);
        const int64_t guess = (int64_t) std::min(2.0 * C * per_leaf + 4096.0, 2.0e9);
        BT_CHECK(ensure_box_capacity(ctx, st, std::max<int64_t>(guess, 1024), sizeof(T)));
    }
    // ---- cell starts of the top levels (point particles, unit weights): see cell_starts_kernel.
    // A self-contained build also takes the counts of its top boxes from them; a sharded one has
    // the global tables for that (top_cell_prefix) and only looks child ranges up here -- searched,
    // its five shared levels took 0.3 ms of few threads' dependent loads at 10^8 points. ------
    Buf<int64_t> local_cells;
    int local_top_level = 0;
    {
        int k = D == 3 ? 5 : D == 2 ? 7 : 15;
        k = std::min(k, packed ? pk_sorted : st->L);
        if (!EXT && !p.refine_weights && N >= 4096 && k >= 2
                && p.kind != BT_KIND_ADAPTIVE_LEVEL_RESTRICTED) {
            const int64_t ncells = (int64_t) 1 << (D * k);
            BT_CHECK(local_cells.alloc(ctx->pool, ncells + 1));
            cell_starts_kernel<<<(unsigned) div_up(ncells + 1, 256), 256, 0, ctx->stream>>>(
                keys, N, packed ? pk_idbits + D * (pk_levels - k) : D * (st->L - k), ncells,
                local_cells.get());
            BT_HIP_CHECK(hipGetLastError());
            local_top_level = k;
        }
    }
    Buf<LoopState> d_ls;
    Buf<uint32_t> tickets;
    BT_CHECK(d_ls.alloc(ctx->pool, 1));
    BT_CHECK(tickets.alloc(ctx->pool, BT_MAX_LEVELS + 3));
    LoopState h_ls_value{};
    LoopState *h_ls = &h_ls_value;

    auto base_args = [&](BuildArgs &a) {
        a = BuildArgs{};
        a.wprefix = st->wprefix.get();
        a.box_start = st->box_start.get(); a.box_count = st->box_count.get();
        a.box_parent = st->box_parent.get(); a.box_nonchild = st->box_nonchild.get();
        a.box_child = st->box_child.get();
        a.box_level = st->box_level.get(); a.box_haschild = st->box_haschild.get();
        a.status = ctx->d_status;
        a.max_weight = p.max_leaf_refine_weight;
        a.capbits = st->capbits;
        a.idshift = packed ? pk_idbits : 0;
        a.adaptive = p.kind != BT_KIND_NON_ADAPTIVE;
        a.top_level = p.top_level;
        a.top_prefix = p.top_cell_prefix;
        a.top_arrive = p.top_box_arrive;
        a.top_stay = p.top_box_stay;
        if (local_cells.get()) {
            a.cell_starts = local_cells.get();
            a.cell_level = local_top_level;
            if (!p.top_cell_prefix) {           // the build's own counts are the global ones
                a.top_level = local_top_level;
                a.top_prefix = local_cells.get();
            }
        }
        a.keep_empty = p.skip_prune ? 1 : 0;
    };
    {
        BuildArgs a;
        base_args(a);
        T mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
        for (int ax = 0; ax < D; ++ax) { mn[ax] = (T) p.bbox_min[ax]; mx[ax] = (T) p.bbox_max[ax]; }
        init_root_kernel<T, D><<<1, 128, 0, ctx->stream>>>(a, d_ls.get(), (T *) st->centers.get(),
                (int32_t) N, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], tickets.get(),
                (const T *) st->rootbox.get());
    }
    st->nboxes = 1;
    st->level_start = {0, 1};

    // ---- continuation keys: the particles of the boxes `pr` selects among [pr.b0, pr.b0 + nb)
    // (boxes of level L1 = st->L that have to split, or may have to) are re-keyed for the levels
    // L1+1..31 (keygen2_kernel), sorted -- the key prefixed by the box's rank, so that ONE global
    // stable sort keeps the boxes apart -- and put back into their ranges: keys_oth becomes the key
    // array of the deeper levels, ids follows.  cand[i] = 1 for the boxes that were re-keyed.
    const int L2_cont = KEY_AXIS_BITS - st->L;
    auto rekey_boxes = [&](RekeyPred pr, int nb, Buf<uint8_t> &cand) -> int {
        const int L1 = st->L, L2 = L2_cont;
        const int keybits2 = D * L2 + st->capbits;
        Buf<int32_t> seg_rank, seg_off, seg_box, seg_start;
        BT_CHECK(seg_rank.alloc(ctx->pool, nb + 1));
        BT_CHECK(seg_off.alloc(ctx->pool, nb + 1));
        BT_CHECK(cand.alloc(ctx->pool, nb));
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, RekeyFlag{pr}, nb, seg_rank.get(),
                                                          (int32_t *) nullptr, true)));
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, RekeyCount{pr}, nb, seg_off.get(),
                                                          (int32_t *) nullptr, true)));
        int32_t h_tot[2] = {0, 0};
        BT_CHECK(bt::d2h(ctx, &h_tot[0], seg_rank.get() + nb, 4));
        BT_CHECK(bt::d2h(ctx, &h_tot[1], seg_off.get() + nb, 4));
        BT_CHECK(bt::sync_stream(ctx));
        ctx->n_host_syncs++;
        const int32_t nseg = h_tot[0];
        const int64_t M = h_tot[1];
        int segbits = 1;
        while (((int64_t) 1 << segbits) < nseg) ++segbits;
        if (keybits2 + segbits > 64) {
            set_error("%d boxes of level %d have to be refined below the reach of the 64-bit "
                      "Morton key: more than its continuation can tell apart", nseg, L1);
            return BT_ERR_UNSUPPORTED;
        }
        BT_CHECK(seg_box.alloc(ctx->pool, nseg));
        BT_CHECK(seg_start.alloc(ctx->pool, nseg + 1));
        rekey_segments_kernel<<<(unsigned) div_up(nb + 1, 256), 256, 0, ctx->stream>>>(
            nb, pr, seg_rank.get(), seg_off.get(), cand.get(), seg_box.get(), seg_start.get());
        Buf<uint64_t> k2a, k2b;
        Buf<uint32_t> i2a, i2b;
        Buf<int32_t> positions;
        BT_CHECK(k2a.alloc(ctx->pool, M));
        BT_CHECK(k2b.alloc(ctx->pool, M));
        BT_CHECK(i2a.alloc(ctx->pool, M));
        BT_CHECK(i2b.alloc(ctx->pool, M));
        BT_CHECK(positions.alloc(ctx->pool, M));
        Keygen2Args<T, D> ka2{};
        ka2.packed = (const T *) st->packed.get();
        ka2.src_radii = (const T *) p.source_radii;
        ka2.tgt_radii = (const T *) p.target_radii;
        ka2.nsources = st->nsources;
        for (int ax = 0; ax < D; ++ax) {
            ka2.bbox_min[ax] = (T) p.bbox_min[ax];
            ka2.bbox_max[ax] = (T) p.bbox_max[ax];
        }
        ka2.stick_out_factor = (T) p.stick_out_factor;
        ka2.L1 = L1; ka2.L2 = L2; ka2.capbits = st->capbits; ka2.norm = p.extent_norm;
        ka2.point_skip_levels = std::min(point_skip_raw, KEY_AXIS_BITS);
        ka2.ids = ids;
        ka2.box_start = st->box_start.get();
        ka2.seg_box = seg_box.get(); ka2.seg_start = seg_start.get();
        ka2.nseg = nseg; ka2.m = M;
        const unsigned kblocks = (unsigned) div_up(M, 256);
        if (M > 0) {
            if (EXT) keygen2_kernel<T, D, true><<<kblocks, 256, 0, ctx->stream>>>(ka2, k2a.get(), i2a.get(), positions.get());
            else keygen2_kernel<T, D, false><<<kblocks, 256, 0, ctx->stream>>>(ka2, k2a.get(), i2a.get(), positions.get());
            bool in_b = false;
            BT_CHECK(radix_sort_pairs<uint64_t>(ctx, k2a.get(), i2a.get(), k2b.get(), i2b.get(), M, 0,
                                                keybits2 + segbits, false, &in_b));
            // the re-keyed boxes keep their ranges: compact index j <-> position positions[j]
            // (ascending), so the sorted pairs go back in order; keys_oth is free after the
            // full sort and becomes the key array of the deeper levels
            scatter_rekeyed_kernel<<<kblocks, 256, 0, ctx->stream>>>(
                M, positions.get(), in_b ? k2b.get() : k2a.get(), in_b ? i2b.get() : i2a.get(),
                keys_oth, ids);
        }
        if (p.refine_weights) {
            // positions inside the re-keyed boxes changed: weight prefix sums again
            GatherWeight gw{p.refine_weights, ids};
            BT_CHECK((device_exclusive_scan<int64_t, int64_t>(ctx, gw, N, st->wprefix.get(),
                                                              (int64_t *) nullptr, true)));
        }
        BT_HIP_CHECK(hipGetLastError());
        return BT_OK;
    };

    const bool level_restricted = p.kind == BT_KIND_ADAPTIVE_LEVEL_RESTRICTED;
    if (level_restricted && N > 0) {
        // a level-restricted tree may split any leaf later: when the level loop first needs a
        // box below the reach of the key, ALL non-empty boxes of that level are re-keyed
        // (boxes [b0, b0 + nb) in creation order: all of them at first, later the new ones)
        std::function<int(int64_t, int64_t)> rekey_level = [&](int64_t b0, int64_t nb) -> int {
            RekeyPred pr{st->box_count.get(), st->wprefix.get(), st->box_start.get(), (int32_t) b0,
                         p.max_leaf_refine_weight, 1, st->box_level.get(), st->L};
            Buf<uint8_t> cand;
            return rekey_boxes(pr, (int) nb, cand);
        };
        BT_CHECK((lr_build_boxes<T, D>(ctx, st, keys, keys_oth, L2_cont, rekey_level)));
    }
    // tree_build.py:676: the level loop is not entered at all when the root is not
    // overfull -- this also keeps a non-adaptive tree, which otherwise splits every
    // box of a level, at a single box
    bool enter_loop = N > 0 && !level_restricted;
    if (enter_loop && p.kind == BT_KIND_NON_ADAPTIVE) {
        int64_t total = N;
        if (st->wprefix.get()) {
            BT_CHECK(bt::d2h(ctx, &total, st->wprefix.get() + N, 8));
            BT_CHECK(bt::sync_stream(ctx));
        }
        if (total <= (int64_t) p.max_leaf_refine_weight) enter_loop = false;
    }

    // descriptors of the level kernels' look-back (generation-tagged, never cleared
    // between launches; cleared when the 16-bit tag wraps)
    Buf<uint64_t> sl_desc;
    auto ensure_desc = [&]() -> int {
        const int64_t need = st->cap / (256 / C * SL_SUB) + 2;
        if (sl_desc.size() < need) {
            if (uint64_t *z = (uint64_t *) bt::zero_alloc(ctx, (size_t) need * 8)) {
                sl_desc.set_external(z, need);
            } else {
                BT_CHECK(sl_desc.alloc(ctx->pool, need));
                BT_HIP_CHECK(hipMemsetAsync(sl_desc.get(), 0, (size_t) need * 8, ctx->stream));
            }
        }
        return BT_OK;
    };
    uint32_t sl_gen = 0;

    // Packed keys -> (full key, id) arrays in the current order: the tree got deeper than
    // the packed path bits reach (with_keys), or the continuation below needs the ids.
    auto unpack_keys = [&](bool with_keys) -> int {
        if (!st->ids_b.get()) BT_CHECK(st->ids_b.alloc(ctx->pool, N));
        ids = st->ids_a.get(); ids_other = st->ids_b.get();
        const uint64_t mask = ((uint64_t) 1 << pk_idbits) - 1;
        const unsigned nblk = (unsigned) div_up(N, 256);
        if (EXT)
            rekey_full_kernel<T, D, true><<<nblk, 256, 0, ctx->stream>>>(
                N, keys_cur, mask, ka, (const T *) st->packed.get(), ids, with_keys ? keys_oth : nullptr);
        else
            rekey_full_kernel<T, D, false><<<nblk, 256, 0, ctx->stream>>>(
                N, keys_cur, mask, ka, (const T *) st->packed.get(), ids, with_keys ? keys_oth : nullptr);
        BT_HIP_CHECK(hipGetLastError());
        if (with_keys) std::swap(keys_cur, keys_oth);
        packed = false;
        return BT_OK;
    };

    // Levels first_level .. on one key array (`kkeys` addresses the levels loff+1 ..
    // loff+Lkey).  Launches are queued in batches without looking at the result; the
    // state comes back once per batch.  *need_more: a box of level loff+Lkey must split
    // (and a continuation key exists).
    auto level_loop = [&](const uint64_t *&kkeys, int Lkey, int loff, const uint8_t *cand,
                          bool can_continue, int first_level, bool *need_more) -> int {
        int next = first_level;
        const int deepest = loff + Lkey + 1;         // its launch only reports "too deep"
        // depth estimate for the first batch: points on a (D-1)-dimensional set fill
        // 2^(D-1) children per split
        int batch = 4;
        if (loff == 0) {
            const double per_leaf = std::max(1.0, (double) N / std::max(1, p.max_leaf_refine_weight));
            const int fan = D > 1 ? D - 1 : 1;
            batch = std::max(4, (int) std::ceil(std::log2(per_leaf) / fan) + 2);
        }
        while (true) {
            if (loff == 0 && D * next > sorted_high_bits) {
                // deeper than the sorted key bits reach: order all bits now.  Box ranges
                // stay valid (the order of the top bits does not change); ties keep
                // whatever order they have, the fix-up sorts leaves by user id anyway.
                if (packed) BT_CHECK(unpack_keys(true));
                bool in_b = false;
                BT_CHECK(radix_sort_pairs<uint64_t>(ctx, keys_cur, ids, keys_oth, ids_other, N, 0,
                                                    keybits, false, &in_b));
                if (in_b) { std::swap(keys_cur, keys_oth); std::swap(ids, ids_other); }
                kkeys = keys_cur;
                sorted_high_bits = keybits;
            }
            int last = std::min(deepest, next + batch - 1);
            if (loff == 0) last = std::min(last, std::max(next, sorted_high_bits / D));
            BT_CHECK(ensure_desc());
            for (int l = next; l <= last; ++l) {
                BuildArgs a;
                base_args(a);
                a.keys = kkeys;
                a.level = l; a.L = (packed && loff == 0) ? pk_levels : Lkey; a.loff = loff;
                a.can_continue = can_continue ? 1 : 0;
                a.cand = (l == loff + 1) ? cand : nullptr;
                sl_gen += 1;
                if ((sl_gen & 0xffffu) == 0) {       // tag wrapped: no stale word may survive
                    BT_HIP_CHECK(hipMemsetAsync(sl_desc.get(), 0, (size_t) sl_desc.size() * 8, ctx->stream));
                    sl_gen += 1;
                }
                // parents of level l-1: at most C^(l-1) (first key) and at most the capacity
                double bound = loff == 0 ? std::pow((double) C, l - 1) : (double) st->cap;
                bound = std::min(bound, (double) st->cap);
                const int64_t tiles = (int64_t) std::ceil(bound / (256 / C * SL_SUB));
                const unsigned grid = (unsigned) std::max<int64_t>(
                    1, std::min<int64_t>(tiles, (int64_t) ctx->num_cus * 8));
                if (EXT)
                    split_level_kernel<T, D, true><<<grid, 256, 0, ctx->stream>>>(
                        a, d_ls.get(), (T *) st->centers.get(), (T) p.root_extent, (int32_t) st->cap,
                        first_level, sl_desc.get(), sl_gen, tickets.get() + l,
                        (const T *) st->rootbox.get());
                else
                    split_level_kernel<T, D, false><<<grid, 256, 0, ctx->stream>>>(
                        a, d_ls.get(), (T *) st->centers.get(), (T) p.root_extent, (int32_t) st->cap,
                        first_level, sl_desc.get(), sl_gen, tickets.get() + l,
                        (const T *) st->rootbox.get());
            }
            BT_HIP_CHECK(hipGetLastError());
            BT_CHECK(bt::d2h(ctx, h_ls, d_ls.get(), sizeof(LoopState)));
            // (the status word -- look-back timeouts of the sort, depth limit -- comes back
            // in the same wait)
            BT_CHECK(check_status(ctx));
            ctx->n_host_syncs++;
            if (h_ls->overflow) {
                // the children of level `lv` did not fit: grow (keeping the boxes of the
                // levels above) and repeat from that level
                const int lv = h_ls->overflow;
                st->nboxes = h_ls->level_start[lv];
                BT_CHECK(ensure_box_capacity(ctx, st, std::max<int64_t>(st->cap * 2, st->nboxes + 1024),
                                             sizeof(T)));
                reset_loop_kernel<<<1, 128, 0, ctx->stream>>>(d_ls.get(), tickets.get(), lv);
                next = lv;
                continue;
            }
            if (h_ls->need_more) *need_more = true;
            // the loop goes on iff the last queued level was created with an overfull
            // child (tree_build.py:1228-1230)
            const bool created_last = h_ls->level_start[last + 1] > h_ls->level_start[last];
            if (h_ls->done || !created_last || !h_ls->oversize[last] || last >= deepest) break;
            next = last + 1;
            batch = 4;
        }
        // host copy of the level starts: levels [0, nl) exist
        int nl = first_level;
        while (nl <= BT_MAX_LEVELS && h_ls->level_start[nl + 1] > h_ls->level_start[nl]) ++nl;
        st->level_start.resize((size_t) first_level + 1);
        for (int l = first_level; l < nl; ++l) st->level_start.push_back(h_ls->level_start[l + 1]);
        st->nboxes = st->level_start.back();
        return BT_OK;
    };

    // levels addressable below the first key (the per-axis cell index has 31 bits)
    const int L2 = L2_cont;
    bool need_more = false;
    bool status_read = false;      // the level loop's last wait brought the status word
    if (enter_loop) {
        BT_CHECK(level_loop(keys, st->L, 0, nullptr, L2 > 0, 1, &need_more));
        status_read = true;
    }
    if (p.compute_root_box) {
        // the root box the device computed, for the host-side uses below and the caller
        if (!status_read) { BT_CHECK(check_status(ctx)); status_read = true; }
        const T *h = reinterpret_cast<const T *>(st->h_rootbox);
        for (int ax = 0; ax < D; ++ax) {
            st->p.bbox_min[ax] = (double) h[ax];
            st->p.bbox_max[ax] = (double) h[3 + ax];
        }
        st->p.root_extent = (double) h[6];
    }
    if (need_more) {
        // ---- continuation below level L1 = st->L (keygen2_kernel) ---------------------
        if (packed) BT_CHECK(unpack_keys(false));
        const int L1 = st->L;
        const int b0 = st->level_start[L1];
        const int nb = st->level_start[L1 + 1] - b0;
        RekeyPred pr{st->box_count.get(), st->wprefix.get(), st->box_start.get(), b0,
                     p.max_leaf_refine_weight, p.kind != BT_KIND_NON_ADAPTIVE, nullptr, 0};
        Buf<uint8_t> cand;
        BT_CHECK(rekey_boxes(pr, nb, cand));
        const uint64_t *keys2 = keys_oth;
        bool dummy = false;
        // the launch of level L1+1 on the first key took that level's tickets
        reset_loop_kernel<<<1, 128, 0, ctx->stream>>>(d_ls.get(), tickets.get(), L1 + 1);
        BT_CHECK(level_loop(keys2, L2, L1, cand.get(), false, L1 + 1, &dummy));
        // (the scratch buffers of this block are released after the work that reads them
        // has been queued on the stream; the pool hands memory to this stream only)
    }
    if (!status_read) BT_CHECK(check_status(ctx));
    BT_CHECK(mark(ctx, st, "boxes"));

    // ---- within-box order fix-up ----------------------------------------------
    st->ids = ids;
    st->ids_other = ids_other;
    st->pk = packed ? keys_cur : nullptr;
    st->pk_mask = packed ? ((uint64_t) 1 << pk_idbits) - 1 : 0;
    st->fixup_done = false;
    BT_CHECK(fixup_launch(ctx, st));
    BT_CHECK(mark(ctx, st, "fixup"));

    // keys are no longer needed
    st->keys_a.reset();
    st->keys_b.reset();
    st->wprefix.reset();

    // (no wait here: the fix-up kernel runs while the caller allocates the output arrays;
    // everything the sizes below depend on was read in the level loop)

    out->nboxes = st->nboxes;
    out->aligned_nboxes = div_up(st->nboxes, 32) * 32;
    out->nlevels = (int32_t) st->level_start.size() - 1;
    out->key_levels = st->L;
    for (int ax = 0; ax < BT_MAX_DIMS; ++ax) {
        out->bbox_min[ax] = ax < D ? p.bbox_min[ax] : 0.0;
        out->bbox_max[ax] = ax < D ? p.bbox_max[ax] : 0.0;
    }
    out->root_extent = p.root_extent;
    for (size_t i = 0; i < st->level_start.size() && i <= BT_MAX_LEVELS; ++i)
        out->level_start_box_nrs[i] = st->level_start[i];
    st->built = true;
    return BT_OK;
}

template <class T, int D>
int tree_export_impl(bt_context *ctx, TreeState *st, const bt_tree_arrays *o)
{
    constexpr int C = 1 << D;
    const bt_tree_params &p = st->p;
    const int64_t N = st->N, B = st->nboxes;
    const int64_t aligned = div_up(B, 32) * 32;
    const bool sat = st->sat;
    auto blocks = [](int64_t n) { return (unsigned) std::max<int64_t>(1, div_up(n, 256)); };

    BT_CHECK(mark(ctx, st, "(host gap)"));
    BT_CHECK(fixup_finish(ctx, st));
    // ---- source prefix (separate targets), over the ids in their final order -------------
    if (!st->sat && !st->src_before.get()) {
        const int64_t nw = div_up(N, 64);
        Buf<int32_t> counts;
        BT_CHECK(st->src_bits.alloc(ctx->pool, nw + 1));
        BT_CHECK(st->src_before.alloc(ctx->pool, nw + 2));
        BT_CHECK(counts.alloc(ctx->pool, nw + 1));
        // (one more word than positions need: S(N) looks at word N / 64 when N is a multiple of 64)
        BT_HIP_CHECK(hipMemsetAsync(st->src_bits.get() + nw, 0, 8, ctx->stream));
        BT_HIP_CHECK(hipMemsetAsync(counts.get() + nw, 0, 4, ctx->stream));
        source_bits_kernel<<<blocks(nw * 64), 256, 0, ctx->stream>>>(N, st->ids, (uint32_t) st->nsources,
                                                                   st->src_bits.get(), counts.get());
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, ScanCounts{counts.get()}, nw + 1,
                                                          st->src_before.get(), (int32_t *) nullptr, true)));
    }
    BT_CHECK(mark(ctx, st, "srcscan"));
    BT_CHECK(mark(ctx, st, "leaves"));
    const uint32_t *final_ids = st->ids;
    // ---- ids -------------------------------------------------------------------
    if (N > 0) {
        if (sat && N >= ((int64_t) 1 << 22)) {
            // sorted_target_ids = inverse of the sort permutation.  A direct scatter
            // costs a DRAM burst per 4-byte write (2.26 ms at 10^8), scattering into four
            // destination windows of ~100 MB one after the other 1.86 ms (8 windows: 2.47, every
            // window re-reads the ids; dropped); grouping the (id, position) pairs by the id's top byte
            // first (one 32-bit onesweep pass that synthesises the positions) makes
            // the scatter local.
            int bits = 0;
            while (((int64_t) 1 << bits) < N) ++bits;
            Buf<uint32_t> grouped_ids, positions;
            BT_CHECK(grouped_ids.alloc(ctx->pool, N));
            BT_CHECK(positions.alloc(ctx->pool, N));
            bool in_b = false;
            // (user_source_ids is a copy of the ids: the sort's histogram pass writes it)
            ctx->sort_copy_keys = o->user_source_ids;
            BT_CHECK(radix_sort_pairs<uint32_t>(ctx, const_cast<uint32_t *>(final_ids), nullptr,
                                                grouped_ids.get(), positions.get(), N, bits - 8, bits,
                                                true, &in_b));
            // blocks rounded up to a multiple of 8 so that every XCD gets whole stretches
            // (measured at 10^8: 1.53 ms for the stage, 1.72 without the XCD mapping,
            // 1.95 with a second pass, 1.95 for the four-window scatter)
            const unsigned nb = (unsigned) ((blocks(N) + 7) / 8 * 8);
            scatter_inverse_kernel<<<nb, 256, 0, ctx->stream>>>(
                N, grouped_ids.get(), positions.get(), o->sorted_target_ids);
        } else if (sat) {
            same_ids_kernel<<<blocks(N), 256, 0, ctx->stream>>>(N, final_ids, o->user_source_ids,
                                                              o->sorted_target_ids);
        } else {
            BT_CHECK(st->srcntgt_target_ids.alloc(ctx->pool, st->ntargets));
            split_ids_kernel<<<blocks(N), 256, 0, ctx->stream>>>(
                N, st->ids, st->src_before.get(), (uint32_t) st->nsources, o->user_source_ids,
                st->srcntgt_target_ids.get(), o->sorted_target_ids);
        }
    }
    BT_CHECK(mark(ctx, st, "ids"));

    // ---- coordinates and radii (tbk:1170-1186, tree_build.py:1569-1622) ----------
    {
        const T *packed = (const T *) st->packed.get();
        GatherOut<T, D> gs, gt;
        for (int ax = 0; ax < D; ++ax) {
            gs.out[ax] = (T *) o->sources[ax];
            gt.out[ax] = sat ? nullptr : (T *) o->targets[ax];
        }
        if (st->nsources > 0)
            gather_packed_kernel<T, D><<<blocks(st->nsources), 256, 0, ctx->stream>>>(
                st->nsources, o->user_source_ids, packed, gs);
        if (!sat && st->ntargets > 0)
            gather_packed_kernel<T, D><<<blocks(st->ntargets), 256, 0, ctx->stream>>>(
                st->ntargets, st->srcntgt_target_ids.get(), packed, gt);
    }
    if (p.source_radii && o->source_radii && st->nsources > 0)
        gather_kernel<T><<<blocks(st->nsources), 256, 0, ctx->stream>>>(
            st->nsources, o->user_source_ids, 0, (const T *) p.source_radii, (T *) o->source_radii);
    if (p.target_radii && o->target_radii && st->ntargets > 0)
        gather_kernel<T><<<blocks(st->ntargets), 256, 0, ctx->stream>>>(
            st->ntargets, st->srcntgt_target_ids.get(), (int32_t) st->nsources,
            (const T *) p.target_radii, (T *) o->target_radii);
    BT_CHECK(mark(ctx, st, "gather"));

    // ---- per-box arrays -----------------------------------------------------------
    {
        BoxInfoArgs a{};
        a.nboxes = (int) B; a.aligned = aligned; a.C = C; a.D = D;
        a.sat = sat; a.have_extent = st->have_extent;
        a.box_start = st->box_start.get(); a.box_count = st->box_count.get();
        a.box_parent = st->box_parent.get(); a.box_nonchild = st->box_nonchild.get();
        a.box_child = st->box_child.get(); a.box_level = st->box_level.get();
        a.box_haschild = st->box_haschild.get();
        a.src_prefix = SrcPrefix{st->src_bits.get(), st->src_before.get()};
        a.o_src_starts = o->box_source_starts; a.o_src_nonchild = o->box_source_counts_nonchild;
        a.o_src_cumul = o->box_source_counts_cumul;
        a.o_tgt_starts = o->box_target_starts; a.o_tgt_nonchild = o->box_target_counts_nonchild;
        a.o_tgt_cumul = o->box_target_counts_cumul;
        a.o_parent = o->box_parent_ids; a.o_child = o->box_child_ids;
        a.o_levels = o->box_levels; a.o_flags = o->box_flags;
        a.centers = st->centers.get(); a.o_centers = o->box_centers; a.csize = (int) sizeof(T);
        // (box_extent_kernel writes every box of every level: only the padding needs zeros)
        a.o_extents[0] = o->box_source_bounding_box_min;
        a.o_extents[1] = o->box_source_bounding_box_max;
        a.o_extents[2] = sat ? nullptr : o->box_target_bounding_box_min;
        a.o_extents[3] = sat ? nullptr : o->box_target_bounding_box_max;
        a.o_level_starts = o->level_start_box_nrs;
        a.o_sizes = o->box_subtree_sizes;
        a.n_level_starts = (int) std::min<size_t>(st->level_start.size(), BT_MAX_LEVELS + 1);
        for (int i = 0; i < a.n_level_starts; ++i) a.level_starts[i] = st->level_start[(size_t) i];
        box_info_kernel<<<blocks(aligned), 256, 0, ctx->stream>>>(a);
    }
    BT_CHECK(mark(ctx, st, "boxinfo"));

    // ---- box extents, bottom-up (tree_build.py:1730-1806) ----------------------------
    const int nlevels = (int) st->level_start.size() - 1;
    for (int round = 0; round < 2; ++round) {
        if (round == 1 && sat) continue;
        T *bmin = (T *) (round == 0 ? o->box_source_bounding_box_min : o->box_target_bounding_box_min);
        T *bmax = (T *) (round == 0 ? o->box_source_bounding_box_max : o->box_target_bounding_box_max);

        // (with the leaves done the deepest level has nothing left)
        for (int lev = nlevels - 1; lev >= 0; --lev) {
            ExtentArgs<T, D> a;
            a.leaves_done = 0;
            a.b0 = st->level_start[lev];
            a.nb = st->level_start[lev + 1] - a.b0;
            a.aligned = aligned;
            a.starts = round == 0 ? o->box_source_starts : o->box_target_starts;
            a.counts_nonchild = round == 0 ? o->box_source_counts_nonchild
                                           : o->box_target_counts_nonchild;
            a.child = o->box_child_ids;
            a.centers = (const T *) o->box_centers;
            for (int ax = 0; ax < D; ++ax)
                a.part[ax] = (const T *) (round == 0 ? o->sources[ax] : o->targets[ax]);
            a.radii = (const T *) (round == 0 ? (p.source_radii ? o->source_radii : nullptr)
                                              : (p.target_radii ? o->target_radii : nullptr));
            a.bmin = bmin; a.bmax = bmax;
            a.sizes = round == 0 ? o->box_subtree_sizes : nullptr;
            box_extent_kernel<T, D><<<blocks((int64_t) a.nb * 16), 256, 0, ctx->stream>>>(a);
        }
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(mark(ctx, st, "extents"));
    // (a stream-ordered context does not wait here: the caller's arrays are ordered by
    // the stream, boxtree_hip.h bt_set_stream_ordered)
    if (!ctx->stream_ordered) ctx->n_host_syncs++;
    return bt::finish_call(ctx);
}

template <class T>
int dispatch_dims_build(bt_context *ctx, TreeState *st, bt_tree_sizes *out)
{
    switch (st->D) {
    case 1: return tree_build_impl<T, 1>(ctx, st, out);
    case 2: return tree_build_impl<T, 2>(ctx, st, out);
    case 3: return tree_build_impl<T, 3>(ctx, st, out);
    }
    return BT_ERR_INVALID;
}

template <class T>
int dispatch_dims_export(bt_context *ctx, TreeState *st, const bt_tree_arrays *o)
{
    switch (st->D) {
    case 1: return tree_export_impl<T, 1>(ctx, st, o);
    case 2: return tree_export_impl<T, 2>(ctx, st, o);
    case 3: return tree_export_impl<T, 3>(ctx, st, o);
    }
    return BT_ERR_INVALID;
}

}  // namespace

int bt_trav_stage_times(bt_context *ctx, bt_stage_times *out, int n);   // bt_trav.hip

namespace bt {

// Bounding box of dense coordinate arrays, left on the device: d_mm[0..dims) = min(d_mm, min),
// d_mm[dims..2 dims) = min(d_mm, -max) -- no host wait (the multi-GPU exchange all-reduces it).
int bbox_minmax_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                       const void *radii, int64_t n, double *d_mm)
{
    return coord_kind == BT_F64 ? bbox_fold_impl<double>(ctx, dims, coords, radii, n, d_mm)
                                : bbox_fold_impl<float>(ctx, dims, coords, radii, n, d_mm);
}

// The ownership cell of particles with extents (bt_mgpu_exchange): the key kernel's own stop
// test (particle_key) over the levels 1..k+1.  A particle that sticks out of the boxes of level
// cap + 1 stays in its box of level cap <= k: it counts as STAYING there (hist_stay, index
// (C^cap - 1) / (C - 1) + path) and its cell is the first level-k cell under that box -- the
// rank that owns that cell gets the particle.  hist_cells counts the cells so assigned.
template <class T, int D>
__global__ __launch_bounds__(1024) void ext_cells_kernel(KeygenArgs<T, D> a, int k, uint32_t *cells,
        int32_t *hist_cells, int32_t *hist_stay, int ncells, const int32_t *weights,
        unsigned long long *hist_stay_weight /* or null: stayers are counted only */)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist_ext[];
    const bool lds = ncells <= (1 << 15);
    if (lds) for (int c = threadIdx.x; c < ncells; c += 1024) s_hist_ext[c] = 0;
    // Down to which level can a point (radius 0) not stop?  The build's own bound
    // (tree_build_impl: the margin stick_out_factor / 2 box sizes against 128 spacings of T at
    // the magnitude of the coordinates), from the root box on the device: with an ordinary
    // stick-out factor it exceeds k + 1 and points skip the test; with a factor near zero they
    // do stay in boxes, here as in the build.
    {
        const T *rb = a.rootbox;
        double scale = fabs((double) rb[6]);
        for (int ax = 0; ax < D; ++ax) scale = fmax(scale, fmax(fabs((double) rb[ax]), fabs((double) rb[3 + ax])));
        const double eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
        const double ratio = (double) a.stick_out_factor * (double) rb[6] / (256.0 * eps * scale);
        int skip = 0;
        if (ratio > 1.0) skip = (int) floor(log2(ratio)) - 1;      // (one level of slack: log2 here vs. the host's)
        a.point_skip_levels = skip < 0 ? 0 : (skip > a.L ? a.L : skip);
    }
    __syncthreads();
    constexpr int UNR = 4;
    const int64_t stride = (int64_t) gridDim.x * 1024 * UNR;
    for (int64_t i0 = (int64_t) blockIdx.x * 1024 * UNR + threadIdx.x; i0 < a.n; i0 += stride) {
        T x[UNR][D], radius[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + (int64_t) u * 1024;
#pragma unroll
            for (int ax = 0; ax < D; ++ax) x[u][ax] = i < a.n ? a.tgt[ax][i] : a.rootbox[ax];
            radius[u] = (i < a.n && a.tgt_radii) ? a.tgt_radii[i] : (T) 0;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = i0 + (int64_t) u * 1024;
            if (i >= a.n) continue;
            const uint64_t key = particle_key<T, D, true>(a, x[u], radius[u]);      // L = k + 1
            const int cap = (int) (key & (((uint64_t) 1 << CAPBITS_EXT) - 1));
            const uint32_t cell = (uint32_t) ((key >> CAPBITS_EXT) >> D);     // path of level k, zeros below cap
            cells[i] = cell;
            if (lds) atomicAdd(&s_hist_ext[cell], 1u); else atomicAdd(&hist_cells[cell], 1);
            if (cap <= k) {
                const int64_t at = top_box_index<D>((uint64_t) (cell >> (D * (k - cap))), cap);
                atomicAdd(&hist_stay[at], 1);
                if (hist_stay_weight) atomicAdd(&hist_stay_weight[at], (unsigned long long) (weights ? weights[i] : 1));
            }
        }
    }
    __syncthreads();
    if (lds)
        for (int c = threadIdx.x; c < ncells; c += 1024) {
            const uint32_t v = s_hist_ext[c];
            if (v) atomicAdd(&hist_cells[c], (int32_t) v);
        }
}

template <class T, int D>
int ext_cells_impl(bt_context *ctx, const void *const *coords, const void *radii, int64_t n,
                   const void *d_rootbox, int k, double stick_out_factor, int norm, uint32_t *cells,
                   int32_t *hist_cells, int32_t *hist_stay, const int32_t *weights, int64_t *hist_stay_weight)
{
    if (n == 0) return BT_OK;
    KeygenArgs<T, D> a{};
    for (int ax = 0; ax < D; ++ax) a.tgt[ax] = (const T *) coords[ax];
    a.tgt_radii = (const T *) radii;
    a.nsources = 0; a.n = n; a.src_stride = 1; a.tgt_stride = 1;
    a.rootbox = (const T *) d_rootbox;
    a.stick_out_factor = (T) stick_out_factor;
    a.L = k + 1;
    a.norm = norm;
    a.point_skip_levels = 0;
    const int ncells = 1 << (D * k);
    const unsigned blocks = (unsigned) std::min<int64_t>(div_up(n, 4096), ctx->num_cus);
    ext_cells_kernel<T, D><<<blocks, 1024, ncells <= (1 << 15) ? (size_t) ncells * 4 : 0, ctx->stream>>>(
        a, k, cells, hist_cells, hist_stay, ncells, weights, (unsigned long long *) hist_stay_weight);
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

int morton_cells_ext_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                            const void *radii, int64_t n, const void *d_rootbox, int level,
                            double stick_out_factor, int extent_norm, uint32_t *cells_out,
                            int32_t *hist_cells, int32_t *hist_stay, const int32_t *weights,
                            int64_t *hist_stay_weight)
{
#define EC(T, D) return ext_cells_impl<T, D>(ctx, coords, radii, n, d_rootbox, level, stick_out_factor, \
                                             extent_norm, cells_out, hist_cells, hist_stay, weights, hist_stay_weight)
    if (coord_kind == BT_F64) { if (dims == 1) EC(double, 1); else if (dims == 2) EC(double, 2); else EC(double, 3); }
    else { if (dims == 1) EC(float, 1); else if (dims == 2) EC(float, 2); else EC(float, 3); }
#undef EC
}

}  // namespace bt

extern "C" {

int bt_bbox(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
            const void *radii, int64_t n, double *out_min, double *out_max)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !coords || !out_min || !out_max || dims < 1 || dims > BT_MAX_DIMS || n < 0) {
        set_error("bt_bbox: invalid argument");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    if (coord_kind == BT_F64) return bbox_impl<double>(ctx, dims, coords, radii, n, out_min, out_max);
    if (coord_kind == BT_F32) return bbox_impl<float>(ctx, dims, coords, radii, n, out_min, out_max);
    set_error("bt_bbox: unknown coord_kind %d", coord_kind);
    return BT_ERR_INVALID;
}

int bt_tree_build(bt_context *ctx, const bt_tree_params *p, bt_tree_sizes *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !p || !out) { set_error("bt_tree_build: NULL argument"); return BT_ERR_INVALID; }
    host_trace("build:enter");
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    if (p->dims < 1 || p->dims > BT_MAX_DIMS) {
        set_error("bt_tree_build: dims must be 1..3 (got %d)", p->dims);
        return BT_ERR_INVALID;
    }
    if (p->coord_kind != BT_F32 && p->coord_kind != BT_F64) {
        set_error("bt_tree_build: unknown coord_kind %d", p->coord_kind);
        return BT_ERR_INVALID;
    }
    if (p->kind != BT_KIND_ADAPTIVE && p->kind != BT_KIND_NON_ADAPTIVE
            && p->kind != BT_KIND_ADAPTIVE_LEVEL_RESTRICTED) {
        set_error("unknown tree kind %d", p->kind);
        return BT_ERR_INVALID;
    }
    if (p->max_leaf_refine_weight <= 0) {
        set_error("'max_leaf_refine_weight' must be positive");
        return BT_ERR_INVALID;
    }
    const bool sat = p->ntargets < 0;
    const bool have_extent = p->source_radii || p->target_radii;
    if (p->top_cell_prefix) {
        if (p->top_level < 1 || p->dims * p->top_level > 30) {
            set_error("bt_tree_build: top_level %d out of range", p->top_level);
            return BT_ERR_INVALID;
        }
        if (p->kind != BT_KIND_ADAPTIVE) {
            set_error("sharded builds (top_cell_prefix) support kind='adaptive' only");
            return BT_ERR_UNSUPPORTED;
        }
        if ((have_extent || p->refine_weights) && (!p->top_box_arrive || !p->top_box_stay)) {
            set_error("sharded builds of particles with extents or refine weights need the tables "
                      "top_box_arrive / top_box_stay (bt_mgpu_exchange makes them)");
            return BT_ERR_UNSUPPORTED;
        }
    }
    if (have_extent && sat) {
        set_error("must specify targets when specifying any kind of radii");
        return BT_ERR_INVALID;
    }
    if (have_extent && p->extent_norm != BT_NORM_LINF && p->extent_norm != BT_NORM_L2) {
        set_error("bad extent_norm %d", p->extent_norm);
        return BT_ERR_INVALID;
    }
    const int64_t N = p->nsources + (sat ? 0 : p->ntargets);
    if (p->nsources < 0 || N >= ((int64_t) 1 << 31)) {
        set_error("particle count %lld outside [0, 2^31)", (long long) N);
        return BT_ERR_INVALID;
    }
    for (int ax = 0; ax < p->dims; ++ax) {
        if ((p->nsources > 0 && !p->sources[ax]) || (!sat && p->ntargets > 0 && !p->targets[ax])) {
            set_error("bt_tree_build: NULL coordinate array");
            return BT_ERR_INVALID;
        }
        // max == min on every axis (a single point, or all points coincident) is a
        // tree of one box unless that box would have to split, which no depth can
        // achieve: MaxLevelsExceeded, like the level loop upstream
        if (!p->compute_root_box && !(p->bbox_max[ax] >= p->bbox_min[ax]) && N > 0) {
            set_error("bt_tree_build: empty bounding box on axis %d", ax);
            return BT_ERR_INVALID;
        }
    }
    if (p->compute_root_box
            && (have_extent || p->top_cell_prefix || p->kind == BT_KIND_ADAPTIVE_LEVEL_RESTRICTED
                || p->source_stride > 1 || p->target_stride > 1 || N == 0)) {
        set_error("compute_root_box: point particles in dense arrays, kind 'adaptive' or "
                  "'non-adaptive', a self-contained build of at least one particle");
        return BT_ERR_UNSUPPORTED;
    }

    bt_free_tree_state(ctx);
    TreeState *st = new TreeState();
    ctx->tree = st;
    st->p = *p;
    st->D = p->dims; st->C = 1 << p->dims;
    st->f64 = p->coord_kind == BT_F64;
    st->sat = sat; st->have_extent = have_extent;
    st->nsources = p->nsources; st->ntargets = sat ? p->nsources : p->ntargets;
    st->N = N;
    st->capbits = have_extent ? CAPBITS_EXT : 0;
    // levels addressable by the 64-bit key (per-axis value must fit 31 bits)
    st->L = std::min(31, (63 - st->capbits) / st->D);

    BT_CHECK(bt::zero_begin(ctx));          // (resets the status word too)
    int s = st->f64 ? dispatch_dims_build<double>(ctx, st, out)
                    : dispatch_dims_build<float>(ctx, st, out);
    if (s != BT_OK) { bt::drop_pending_reads(ctx); bt_free_tree_state(ctx); }
    host_trace("build:leave");
    return s;
}

int bt_tree_export(bt_context *ctx, const bt_tree_arrays *o)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !o) { set_error("bt_tree_export: NULL argument"); return BT_ERR_INVALID; }
    TreeState *st = ctx->tree;
    if (!st || !st->built) {
        set_error("bt_tree_export: no tree has been built on this context");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    const bool need_tgt = !st->sat;
    // (zero-length particle arrays may come as NULL: an empty rank of a sharded build)
    if ((!o->user_source_ids && st->nsources > 0) || (!o->sorted_target_ids && st->ntargets > 0)
            || !o->box_source_starts
            || !o->box_source_counts_nonchild || !o->box_source_counts_cumul
            || !o->box_parent_ids || !o->box_child_ids || !o->box_centers || !o->box_levels
            || !o->box_flags || !o->box_source_bounding_box_min || !o->box_source_bounding_box_max
            || (need_tgt && (!o->box_target_starts || !o->box_target_counts_nonchild
                             || !o->box_target_counts_cumul || !o->box_target_bounding_box_min
                             || !o->box_target_bounding_box_max))) {
        set_error("bt_tree_export: NULL output array");
        return BT_ERR_INVALID;
    }
    for (int ax = 0; ax < st->D; ++ax)
        if ((st->nsources > 0 && !o->sources[ax]) || (need_tgt && st->ntargets > 0 && !o->targets[ax])) {
            set_error("bt_tree_export: NULL coordinate output");
            return BT_ERR_INVALID;
        }
    host_trace("export:enter");
    // (the export goes on taking from the zeroed block the build started: no clearing here,
    // and the status word keeps what the build's last kernels may have reported)
    int s = st->f64 ? dispatch_dims_export<double>(ctx, st, o)
                    : dispatch_dims_export<float>(ctx, st, o);
    host_trace("export:leave");
    return s;
}

int bt_get_stage_times(bt_context *ctx, bt_stage_times *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !out) return BT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    TreeState *st = ctx->tree;
    int n = 0;
    if (st) {
        if (!st->events.empty()) (void) hipEventSynchronize(st->events.back().second);
        for (size_t i = 1; i < st->events.size() && n < BT_NUM_STAGES; ++i) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, st->events[i - 1].second, st->events[i].second)
                    != hipSuccess) {
                (void) hipGetLastError();
                ms = -1.f;
            }
            out->ms[n] = ms;
            out->name[n] = st->events[i].first;
            ++n;
        }
    }
    out->n = bt_trav_stage_times(ctx, out, n);
    return BT_OK;
}

}  // extern "C"
