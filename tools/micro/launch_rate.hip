// Wave launch rate on gfx950: how long does a kernel of N threads take when its waves do
// (almost) nothing?  build: hipcc -O3 --offload-arch=gfx950 launch_rate.hip -o launch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_empty(const int *in, int *out, long n)
{
    long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (in == nullptr) return;          // never loads
    out[i] = 1;
}

__global__ void k_load_exit(const int *in, int *out, long n)
{
    long i = (long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (in[i >> 5] == 12345) out[i] = 1;    // one load per thread (32 threads share a word), then exit
}

__global__ void k_stride(const int *in, int *out, long n)
{
    // persistent form of k_load_exit
    long acc = 0;
    for (long i = (long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long) gridDim.x * blockDim.x)
        if (in[i >> 5] == 12345) acc++;
    if (acc) out[0] = (int) acc;
}

int main()
{
    const long n = 170000000;
    int *in, *out;
    hipMalloc(&in, (n / 32 + 1) * 4);
    hipMalloc(&out, n * 4);
    hipMemset(in, 0, (n / 32 + 1) * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bs : {64, 128, 256, 512, 1024}) {
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (variant == 0) k_empty<<<(unsigned) ((n + bs - 1) / bs), bs>>>(nullptr, out, n);
                else k_load_exit<<<(unsigned) ((n + bs - 1) / bs), bs>>>(in, out, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("block %4d  %-10s %.3f ms  %.2f Gwaves/s\n", bs, variant ? "load+exit" : "empty", best,
                   n / 64.0 / (best * 1e-3) / 1e9);
        }
    }
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            k_stride<<<blocks, 256>>>(in, out, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("grid-stride %5d blocks of 256: %.3f ms\n", blocks, best);
    }
    return 0;
}
