// TEST INFRASTRUCTURE -- known-answer tests of the emulator itself (hip_emu.hpp, emu_sched.cpp): the
// wave-wide operations against their documented semantics, the two behaviours the kernels of
// boxtree_amd/csrc rely on (lock step of a wave's LDS traffic; a wave-wide operation is ONE operation
// wherever the host compiler puts its call) and the scheduler's modes.  Built and run by
// tests/test_emu.py:   emu_selftest   (exit code 0 = ok; EMU_ORDER may be set)
#include <hip/hip_runtime.h>

#include <vector>

static int g_fail = 0;
#define EXPECT(cond)                                                                   \
    do {                                                                               \
        if (!(cond)) { ++g_fail; if (g_fail < 20) printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

// ---- shuffles, ballot, DPP, lane-local helpers: one wave ---------------------------------------------
__global__ __launch_bounds__(64) void k_wave_ops(int *out /* [16][64] */)
{
    const int lane = threadIdx.x;
    int r = 0;
    out[r++ * 64 + lane] = __shfl(lane * 10, 5, 64);                    // every lane reads lane 5
    out[r++ * 64 + lane] = __shfl(lane * 10, 3, 16);                    // lane 3 of the own 16-lane section
    out[r++ * 64 + lane] = __shfl_up(lane, 1, 64);                      // lane 0 keeps its own
    out[r++ * 64 + lane] = __shfl_down(lane, 2, 8);                     // within sections of 8
    out[r++ * 64 + lane] = __shfl_xor(lane, 1, 64);
    out[r++ * 64 + lane] = (int) __popcll(__ballot(lane % 3 == 0));
    out[r++ * 64 + lane] = (int) (__ballot(lane >= 32) >> 32);
    // DPP: row_shr:1 (bound_ctrl off: lane 0 of a row keeps `old`), row_bcast:15 into rows 1 and 3,
    // quad_perm [1,0,3,2] (0xB1), row_shl:2 with bound_ctrl (zeros shifted in)
    out[r++ * 64 + lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x111, 0xf, 0xf, false);
    out[r++ * 64 + lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x142, 0xa, 0xf, false);
    out[r++ * 64 + lane] = __builtin_amdgcn_update_dpp(0, lane, 0xB1, 0xf, 0xf, false);
    out[r++ * 64 + lane] = __builtin_amdgcn_update_dpp(-7, lane, 0x102, 0xf, 0xf, true);
    out[r++ * 64 + lane] = (int) __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // = lane
    out[r++ * 64 + lane] = (int) __builtin_amdgcn_bitop3_b32(0xF0F0F0F0u, 0xCCCCCCCCu, 0xAAAAAAAAu, 0x90);
    out[r++ * 64 + lane] = __builtin_amdgcn_sbfe(0x50, 4, 3);          // bits 4..6 of 0x50 = 101b -> -3
    // a ballot that only some lanes take part in: the others are not in the mask
    uint64_t part = 0;
    if (lane & 1) part = __ballot(true);
    out[r++ * 64 + lane] = (int) __popcll(part);
}

// ---- lock step of LDS traffic: every lane reads the counter, one lane per value updates it -----------
__global__ __launch_bounds__(64) void k_lockstep(int *out)
{
    __shared__ int s_count[4];
    const int lane = threadIdx.x;
    if (lane < 4) s_count[lane] = 100 * lane;
    __builtin_amdgcn_wave_barrier();
    const int d = lane & 3;
    int total = 0;
    for (int it = 0; it < 3; ++it) {
        const int old = s_count[d];               // all lanes of the wave read ...
        if (lane < 4) s_count[d] = old + 16;      // ... before the first lane of each value writes
        total += old;
        __builtin_amdgcn_wave_barrier();
    }
    out[lane] = total;                            // 3 * 100 d + 16 * (0 + 1 + 2)
}

// ---- one operation wherever the compiler puts the call: the pattern that was cloned once --------------
__global__ __launch_bounds__(128) void k_ternary_scan(const int *in, int n, int *out)
{
    __shared__ int s_w[2];
    const int i = threadIdx.x;
    int v = (i < n) ? in[i] : 0;
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d, 64); if ((i & 63) >= d) incl += u; }
    if ((i & 63) == 63) s_w[i >> 6] = incl;
    __syncthreads();
    out[i] = incl + (i >= 64 ? s_w[0] : 0);
}

// ---- workgroups in ticket order, a spin on a predecessor, dynamic LDS ---------------------------------
__global__ __launch_bounds__(64) void k_tickets(int *ticket, int *flags, int *out)
{
    extern __shared__ int s_dyn[];
    __shared__ int s_t;
    if (threadIdx.x == 0) s_t = atomicAdd(ticket, 1);
    __syncthreads();
    const int t = s_t;
    s_dyn[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (t > 0) while (flags[t - 1] == 0) __builtin_amdgcn_s_sleep(1);
        flags[t] = 1;
    }
    out[t * 64 + threadIdx.x] = s_dyn[63 - threadIdx.x] + (int) blockIdx.x * 0;
}

int main()
{
    {
        std::vector<int> o(16 * 64, 0);
        k_wave_ops<<<1, 64, 0, nullptr>>>(o.data());
        for (int l = 0; l < 64; ++l) {
            int r = 0;
            EXPECT(o[r++ * 64 + l] == 50);
            EXPECT(o[r++ * 64 + l] == ((l & ~15) + 3) * 10);
            EXPECT(o[r++ * 64 + l] == (l == 0 ? 0 : l - 1));
            EXPECT(o[r++ * 64 + l] == ((l & 7) + 2 < 8 ? l + 2 : l));
            EXPECT(o[r++ * 64 + l] == (l ^ 1));
            EXPECT(o[r++ * 64 + l] == 22);
            EXPECT(o[r++ * 64 + l] == -1);
            EXPECT(o[r++ * 64 + l] == ((l & 15) == 0 ? -7 : l - 1));
            EXPECT(o[r++ * 64 + l] == ((l >> 4) == 1 ? 15 : (l >> 4) == 3 ? 47 : -7));
            EXPECT(o[r++ * 64 + l] == (l ^ 1));
            EXPECT(o[r++ * 64 + l] == ((l & 15) + 2 < 16 ? l + 2 : 0));
            EXPECT(o[r++ * 64 + l] == l);
            EXPECT((unsigned) o[r++ * 64 + l] == (0xF0F0F0F0u & ~(0xCCCCCCCCu ^ 0xAAAAAAAAu)));
            EXPECT(o[r++ * 64 + l] == -3);
            EXPECT(o[r++ * 64 + l] == ((l & 1) ? 32 : 0));       // the odd lanes alone took part
        }
    }
    {
        std::vector<int> o(64, -1);
        k_lockstep<<<1, 64, 0, nullptr>>>(o.data());
        for (int l = 0; l < 64; ++l) EXPECT(o[l] == 300 * (l & 3) + 48);
    }
    {
        const int n = 1;                          // thread 0 alone takes the load's arm of the ternary
        std::vector<int> in(128, 0), o(128, -1);
        in[0] = 2083;
        k_ternary_scan<<<1, 128, 0, nullptr>>>(in.data(), n, o.data());
        for (int l = 0; l < 128; ++l) EXPECT(o[l] == 2083);
    }
    {
        const int nb = 37;
        int ticket = 0;
        std::vector<int> flags(nb, 0), o(nb * 64, -1);
        k_tickets<<<nb, 64, 64 * sizeof(int), nullptr>>>(&ticket, flags.data(), o.data());
        EXPECT(ticket == nb);
        for (int t = 0; t < nb; ++t) for (int l = 0; l < 64; ++l) EXPECT(o[t * 64 + l] == t);
    }
    if (g_fail) { printf("emu_selftest: %d checks FAILED\n", g_fail); return 1; }
    printf("emu_selftest: ok\n");
    return 0;
}
