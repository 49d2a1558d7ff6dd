// Random 32-byte record gather under the cache-policy bits of gfx950's global loads: does any
// of sc0 / sc1 / nt make the L2 fetch less than a 128-byte line per 32-byte record?
// Also: records read as one dwordx4 pair (32 B) vs. a 64-byte-aligned layout is not tried --
// the input layout is the caller's.   hipcc --offload-arch=gfx950 -O3 gather_policy.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void make_ids(uint32_t n, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] = hash32(i) % n;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LOAD_PAIR(POL)                                                                            \
    asm volatile("global_load_dwordx4 %0, %2, off " POL "\n\t"                                    \
                 "global_load_dwordx4 %1, %2, off offset:16 " POL "\n\t"                          \
                 "s_waitcnt vmcnt(0)"                                                             \
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory")

template <int POLICY, int PER>
__global__ __launch_bounds__(256) void gather(uint32_t n, const uint32_t *ids, const double4 *rec,
                                              double *x, double *y, double *z)
{
    const uint32_t base = (blockIdx.x * 256 + threadIdx.x);
    #pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = base + (uint32_t) k * gridDim.x * 256;
        if (i >= n) return;
        const double4 *p = rec + ids[i];
        u32x4 a, b;
        if (POLICY == 0) LOAD_PAIR("");
        else if (POLICY == 1) LOAD_PAIR("nt");
        else if (POLICY == 2) LOAD_PAIR("sc1");
        else if (POLICY == 3) LOAD_PAIR("sc0 sc1");
        else if (POLICY == 4) LOAD_PAIR("sc0 sc1 nt");
        else if (POLICY == 5) LOAD_PAIR("sc1 nt");
        else LOAD_PAIR("sc0 nt");
        union { u32x4 v[2]; double d[4]; } u;
        u.v[0] = a; u.v[1] = b;
        x[i] = u.d[0]; y[i] = u.d[1]; z[i] = u.d[2];
    }
}

template <int POLICY, int PER>
static void run(const char *name, uint32_t n, const uint32_t *ids, const double4 *rec, double *x, double *y,
                double *z, hipEvent_t e0, hipEvent_t e1)
{
    const uint32_t grid = (n + 256 * PER - 1) / (256 * PER);
    gather<POLICY, PER><<<grid, 256>>>(n, ids, rec, x, y, z);
    CK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r) gather<POLICY, PER><<<grid, 256>>>(n, ids, rec, x, y, z);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-12s per-thread %d: %.3f ms\n", name, PER, ms / reps);
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t) atoll(argv[1]) : 100000000u;
    uint32_t *ids; double4 *rec; double *x, *y, *z;
    CK(hipMalloc(&ids, (size_t) n * 4));
    CK(hipMalloc(&rec, (size_t) n * 32));
    CK(hipMalloc(&x, (size_t) n * 8)); CK(hipMalloc(&y, (size_t) n * 8)); CK(hipMalloc(&z, (size_t) n * 8));
    CK(hipMemset(rec, 0, (size_t) n * 32));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    make_ids<<<(n + 255) / 256, 256>>>(n, ids);
    run<0, 1>("plain", n, ids, rec, x, y, z, e0, e1);
    run<1, 1>("nt", n, ids, rec, x, y, z, e0, e1);
    run<2, 1>("sc1", n, ids, rec, x, y, z, e0, e1);
    run<3, 1>("sc0 sc1", n, ids, rec, x, y, z, e0, e1);
    run<4, 1>("sc0 sc1 nt", n, ids, rec, x, y, z, e0, e1);
    run<5, 1>("sc1 nt", n, ids, rec, x, y, z, e0, e1);
    run<6, 1>("sc0 nt", n, ids, rec, x, y, z, e0, e1);
    run<0, 4>("plain", n, ids, rec, x, y, z, e0, e1);
    run<1, 4>("nt", n, ids, rec, x, y, z, e0, e1);
    run<4, 4>("sc0 sc1 nt", n, ids, rec, x, y, z, e0, e1);
    return 0;
}
