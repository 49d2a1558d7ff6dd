// Random 32-byte record gather with the accesses confined to a sliding window of W bytes:
// how much does locality at the scale of the L2 (4 MB per XCD) / the Infinity Cache (256 MB)
// buy the tree-order coordinate gather?  hipcc --offload-arch=gfx950 -O3 gather_window.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// ids[i]: window of i (consecutive stretches of `wrecs` positions share a window) + random offset
__global__ void make_ids(uint32_t n, uint32_t wrecs, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t w0 = (i / wrecs) * wrecs;
    const uint32_t span = min(wrecs, n - w0);
    ids[i] = w0 + hash32(i) % span;
}

__global__ __launch_bounds__(256) void gather(uint32_t n, const uint32_t *ids, const double4 *rec,
                                              double *x, double *y, double *z)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double4 r = rec[ids[i]];
    x[i] = r.x; y[i] = r.y; z[i] = r.z;
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
    const double mbs[] = {0.5, 1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 512, 1024, 4096};
    for (double mb : mbs) {
        uint64_t wrecs = (uint64_t) (mb * 1048576.0 / 32.0);
        if (wrecs > n) wrecs = n;
        make_ids<<<(n + 255) / 256, 256>>>(n, (uint32_t) wrecs, ids);
        gather<<<(n + 255) / 256, 256>>>(n, ids, rec, x, y, z);
        CK(hipEventRecord(e0));
        const int reps = 3;
        for (int r = 0; r < reps; ++r) gather<<<(n + 255) / 256, 256>>>(n, ids, rec, x, y, z);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        printf("window %8.1f MB: %.3f ms  (%.2f G records/s, %.0f GB/s of 60-byte algorithmic traffic)\n",
               mb, ms, n / ms * 1e-6, 60.0 * n / ms * 1e-6);
        if (wrecs == n) break;
    }
    return 0;
}
