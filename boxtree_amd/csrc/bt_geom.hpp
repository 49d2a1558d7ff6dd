// Geometry and walk helpers shared by the traversal and area-query kernels.
#pragma once

#include "bt_common.hpp"

namespace {

// Packed per-box record for the tree walks: one cache line holds everything an
// adjacency test needs (the reference's SoA layout costs 5 lines per test).
template <class T, int D>
struct alignas(16) Node {
    T c[D];
    uint32_t lf;        // level | flags << 8
};

// child_t entries of the traversal carry two flag bits of the child next to its
// number (boxes < 2^28 is checked on the host): a tree walk then reads one word per
// step instead of the child's node.
constexpr uint32_t CH_ID_MASK = (1u << 28) - 1u;
constexpr uint32_t CH_SRC = 1u << 28;       // child is a source box
constexpr uint32_t CH_HSC = 1u << 29;       // child has source child boxes

template <class T, int D, bool PACK_FLAGS = false>
__global__ __launch_bounds__(256) void pack_nodes_kernel(int32_t nboxes, int64_t aligned,
        const T *centers, const uint8_t *levels, const uint8_t *flags, const int32_t *child,
        Node<T, D> *nodes, int32_t *child_t)
{
    constexpr int C = 1 << D;
    const int32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    Node<T, D> n;
#pragma unroll
    for (int i = 0; i < D; ++i) n.c[i] = centers[aligned * i + b];
    n.lf = (uint32_t) levels[b] | ((uint32_t) flags[b] << 8);
    nodes[b] = n;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        const int32_t c = child[(int64_t) m * aligned + b];
        uint32_t e = (uint32_t) c;
        if (PACK_FLAGS && c > 0 && c < nboxes) {
            const uint8_t cf = flags[c];
            if (cf & BT_BOX_IS_SOURCE_BOX) e |= CH_SRC;
            if (cf & BT_BOX_HAS_SOURCE_CHILD_BOXES) e |= CH_HSC;
        }
        child_t[(int64_t) b * C + m] = (int32_t) e;
    }
}

// root_extent * 1 / 2^(level + 1) (traversal.py:234-235).  A division by a power of two is an
// exact scaling: ldexp returns the same bits (both are the correctly rounded value of the same real
// number, subnormal results included) in ONE instruction (v_ldexp_f64), where the IEEE f64
// division the expression compiles to takes about twenty -- and the list walks evaluate it per
// candidate box, in kernels that are bound by vector-ALU issue (LAB_NOTES.md section 9).
__device__ __forceinline__ double level_to_rad(double root_extent, int level)
{
    return __builtin_ldexp(root_extent, -(level + 1));
}
__device__ __forceinline__ float level_to_rad(float root_extent, int level)
{
    return __builtin_ldexpf(root_extent, -(level + 1));
}

// traversal.py:279-305
template <class T, int D>
__device__ __forceinline__ bool adj_nbhd(T root_extent, const T *tc, int tl, T nbhd,
                                         const T *sc, int sl)
{
    const T target_rad = level_to_rad(root_extent, tl);
    const T source_rad = level_to_rad(root_extent, sl);
    const T rad_sum = ((2 * (nbhd - 1) + 1) * target_rad + source_rad);
    const T slack = rad_sum + ((target_rad < source_rad) ? target_rad : source_rad);
    T l_inf = 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        T d = tc[i] - sc[i];
        d = (d < 0) ? -d : d;
        l_inf = (d > l_inf) ? d : l_inf;
    }
    return l_inf <= slack;
}

template <class T, int D>
__device__ __forceinline__ bool adj(T root_extent, const T *tc, int tl, const T *sc, int sl)
{
    return adj_nbhd<T, D>(root_extent, tc, tl, (T) 1, sc, sl);
}

// walk state: traversal.py:98-160.  The stack lives in LDS (one column per
// thread): private arrays would go to scratch, and kernels that use scratch get
// only ~8 wave slots per CU here (measured on list13_kernel).  An entry packs the
// parent box and the child slot to resume at: box | slot << 28 (boxes < 2^28 is
// checked on the host).
constexpr int WALK_THREADS = 256;

struct Walk {
    int32_t *stk;      // LDS, element i at stk[i * WALK_THREADS]
    int size;
    int32_t parent;
    int mnr;
    bool go;
    __device__ __forceinline__ explicit Walk(int32_t *lds_column) : stk(lds_column) {}
    __device__ __forceinline__ void init(int32_t start) { size = 0; parent = start; mnr = 0; go = true; }
    template <int C>
    __device__ __forceinline__ void advance()
    {
        while (true) {
            ++mnr;
            if (mnr < C) break;
            go = size > 0;
            if (go) {
                --size;
                const int32_t e = stk[size * WALK_THREADS];
                parent = e & 0x0fffffff;
                mnr = e >> 28;
            } else break;
        }
    }
    __device__ __forceinline__ void push(int32_t nb)
    {
        stk[size * WALK_THREADS] = parent | (mnr << 28);
        ++size;
        parent = nb; mnr = 0;
    }
};

// dynamic LDS of the walk kernels: [walk_cap][256] stack entries, then (list 3)
// [nlevels][256] per-level counters
extern __shared__ __attribute__((aligned(16))) int32_t s_walk_lds[];

// ---- emitters ---------------------------------------------------------------

struct CountEmit {
    int32_t n = 0;
    __device__ __forceinline__ void operator()(int32_t) { ++n; }
};
struct WriteEmit {
    int32_t *p;
    __device__ __forceinline__ void operator()(int32_t b) { *p++ = b; }
};

}  // namespace
