// A chain of small DEPENDENT kernels (the shape of a 10^5-point build: ~90 launches of a few
// microseconds each between four host waits), queued on a stream against the same chain replayed
// as a captured HIP graph -- does a graph shorten the gaps between dependent launches on gfx950?
// build: hipcc -O3 --offload-arch=gfx950 graph_chain.hip -o graph_chain ; run: ./graph_chain [launches] [blocks]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_step(const int *in, int *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + 1;          // reads what the previous launch wrote
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 90;
    const int blocks = argc > 2 ? atoi(argv[2]) : 64;
    const int n = blocks * 256;
    int *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto chain = [&]() {
        for (int l = 0; l < launches; ++l) {
            k_step<<<blocks, 256, 0, s>>>(l & 1 ? b : a, l & 1 ? a : b, n);
        }
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // -- stream
    for (int rep = 0; rep < 3; ++rep) { chain(); CK(hipStreamSynchronize(s)); }
    float best_stream = 1e9f; double best_stream_host = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s)); chain(); CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_stream) best_stream = ms;
        const double h = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (h < best_stream_host) best_stream_host = h;
    }
    // -- graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    float best_graph = 1e9f; double best_graph_host = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_graph) best_graph = ms;
        const double h = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (h < best_graph_host) best_graph_host = h;
    }
    printf("%d dependent launches of %d x 256 threads: stream %.3f ms on the device (%.1f us per launch), "
           "%.3f ms on the host clock; graph %.3f ms (%.1f us per launch), %.3f ms on the host clock\n",
           launches, blocks, best_stream, 1e3 * best_stream / launches, best_stream_host, best_graph,
           1e3 * best_graph / launches, best_graph_host);
    return 0;
}
