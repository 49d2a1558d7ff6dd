// TEST INFRASTRUCTURE -- a CPU emulator of the HIP execution model, just large enough to run the
// kernels of boxtree_amd/csrc UNMODIFIED on the host, so that their logic can be compared with the
// oracle where no GPU is at hand (tests/emu/README.md).  Never part of the product: nothing under
// boxtree_amd/ includes or loads anything from here; the product library is built by hipcc for
// gfx950 and has no CPU path.
//
// Model: a kernel launch runs its workgroups one after the other (in blockIdx order, which is also
// the order tickets are taken in); the threads of a workgroup are fibers.  A fiber runs until it
// reaches __syncthreads(), a wave-wide operation (ballot, shuffle, DPP move, wave barrier) or a
// sleep; a wave-wide operation completes when every unfinished lane of the wave is waiting, and the
// lanes waiting at the same call site form one "active mask" -- which is how converged code behaves
// on the hardware.  Device memory is host memory; streams are synchronous.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace emu {

struct uint3_t { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct ThreadCtx {                    // what a fiber sees of itself
    uint3_t tid, bid, bdim, gdim;
    int lane, wave;
    void *dyn_lds;
    size_t dyn_lds_bytes;
    int lds_phase;                    // 0: none since the last meeting point, 1: loaded, 2: stored
};
extern thread_local ThreadCtx *g_cur;

// wave-wide exchange: every lane deposits `v`; returns the record of the lanes that met at this
// call site (mask + their values)
struct WaveRec { uint64_t mask; uint64_t val[64]; };
__attribute__((noduplicate, convergent)) const WaveRec &wave_meet(uint64_t v, const void *site);
__attribute__((noduplicate, convergent)) void block_barrier();
void spin_yield();

using KernelThunk = void (*)(void *);
void run_grid(dim3 grid, dim3 block, size_t dyn_lds_bytes, KernelThunk fn, void *arg);
void trace_launch(const char *name, dim3 grid, dim3 block, size_t lds);

}  // namespace emu

// ---- language -----------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// LDS variables live in one linker section: the emulator recognises LDS addresses by range
#define __shared__ static __attribute__((section("emu_lds")))
#define __constant__ static const

#define threadIdx (::emu::g_cur->tid)
#define blockIdx (::emu::g_cur->bid)
#define blockDim (::emu::g_cur->bdim)
#define gridDim (::emu::g_cur->gdim)
using dim3 = ::emu::dim3;

static inline int emu_lane() { return ::emu::g_cur->lane; }

// ---- launch: kernel<<<grid, block, lds, stream>>>(args...) becomes emu_launch(...) (transform.py)
#include <tuple>
#include <utility>
namespace emu {
template <class F, class Tuple, size_t... I>
inline void apply_thunk(F &f, Tuple &t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

// the arguments are evaluated once, here; every thread of the grid calls the kernel with them
template <class F, class... A>
inline void launch(const char *name, dim3 grid, dim3 block, size_t lds, void * /*stream*/, F call, A &&... a)
{
    trace_launch(name, grid, block, lds);
    using Tuple = std::tuple<typename std::decay<A>::type...>;
    struct Box { F call; Tuple args; };
    Box box{call, Tuple(std::forward<A>(a)...)};
    run_grid(grid, block, lds, [](void *p) {
        Box *b = (Box *) p;
        apply_thunk(b->call, b->args, std::index_sequence_for<A...>{});
    }, &box);
}
}  // namespace emu

// ---- barriers and wave-wide operations ----------------------------------------------------------
// Every meeting point is also a compiler barrier: LDS variables are statics of this translation unit
// whose addresses never leave it, so without one the optimizer would carry their values in
// registers across a call it can see does not touch them -- across __syncthreads(), that is.
#define EMU_MEMORY_BARRIER() __asm__ __volatile__("" ::: "memory")
static inline void __syncthreads() { EMU_MEMORY_BARRIER(); ::emu::block_barrier(); EMU_MEMORY_BARRIER(); }
static inline void __threadfence_block() { EMU_MEMORY_BARRIER(); }
static inline void __threadfence() { EMU_MEMORY_BARRIER(); }
#define __builtin_amdgcn_fence(...) EMU_MEMORY_BARRIER()
#define __builtin_amdgcn_s_sleep(n) do { EMU_MEMORY_BARRIER(); ::emu::spin_yield(); EMU_MEMORY_BARRIER(); } while (0)
#define EMU_SITE __builtin_return_address(0)
// A wave-wide operation is identified by its call site, so the host compiler must not clone a call
// (jump threading, unrolling, tail duplication) or move it under a branch: the attributes GPU
// compilers give such operations.
#define EMU_WAVE_OP __attribute__((noinline, noduplicate, convergent))
static inline __attribute__((always_inline)) const ::emu::WaveRec &emu_meet_(uint64_t v, const void *site)
{
    __asm__ __volatile__("" ::: "memory");
    const ::emu::WaveRec &r = ::emu::wave_meet(v, site);
    __asm__ __volatile__("" ::: "memory");
    return r;
}

EMU_WAVE_OP static void emu_wave_barrier_() { (void) emu_meet_(0, EMU_SITE); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier_()

EMU_WAVE_OP static uint64_t __ballot(int pred)
{
    const ::emu::WaveRec &r = emu_meet_(pred ? 1 : 0, EMU_SITE);
    uint64_t b = 0;
    for (int i = 0; i < 64; ++i) if (((r.mask >> i) & 1) && r.val[i]) b |= 1ull << i;
    return b;
}
// v_cmp into a lane mask: cond 33 = ICMP_NE (the only one used)
EMU_WAVE_OP static uint64_t emu_uicmp_(uint32_t a, uint32_t b, int cond)
{
    if (cond != 33) { fprintf(stderr, "emu: uicmp condition %d not modelled\n", cond); abort(); }
    const ::emu::WaveRec &r = emu_meet_(a != b ? 1 : 0, EMU_SITE);
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) if (((r.mask >> i) & 1) && r.val[i]) m |= 1ull << i;
    return m;
}
#define __builtin_amdgcn_uicmp(a, b, c) emu_uicmp_((a), (b), (c))

template <class T> static inline uint64_t emu_bits_(T v) { uint64_t u = 0; memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T emu_unbits_(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }

template <class T>
EMU_WAVE_OP static T __shfl(T v, int src, int width = 64)
{
    const ::emu::WaveRec &r = emu_meet_(emu_bits_(v), EMU_SITE);
    const int lane = emu_lane(), base = lane / width * width;
    const int j = base + (((src % width) + width) % width);
    return ((r.mask >> j) & 1) ? emu_unbits_<T>(r.val[j]) : v;
}
template <class T>
EMU_WAVE_OP static T __shfl_xor(T v, int m, int width = 64)
{
    const ::emu::WaveRec &r = emu_meet_(emu_bits_(v), EMU_SITE);
    const int lane = emu_lane(), base = lane / width * width;
    const int j = base + ((lane - base) ^ m);
    return (j < base + width && ((r.mask >> j) & 1)) ? emu_unbits_<T>(r.val[j]) : v;
}
template <class T>
EMU_WAVE_OP static T __shfl_up(T v, unsigned d, int width = 64)
{
    const ::emu::WaveRec &r = emu_meet_(emu_bits_(v), EMU_SITE);
    const int lane = emu_lane(), base = lane / width * width;
    const int j = lane - (int) d;
    return (j >= base && ((r.mask >> j) & 1)) ? emu_unbits_<T>(r.val[j]) : v;
}
template <class T>
EMU_WAVE_OP static T __shfl_down(T v, unsigned d, int width = 64)
{
    const ::emu::WaveRec &r = emu_meet_(emu_bits_(v), EMU_SITE);
    const int lane = emu_lane(), base = lane / width * width;
    const int j = lane + (int) d;
    return (j < base + width && ((r.mask >> j) & 1)) ? emu_unbits_<T>(r.val[j]) : v;
}

// DPP move (v_mov_b32_dpp): the controls used in csrc -- quad_perm, row_shl / row_shr, row_mirror,
// row_half_mirror, row_bcast:15 / :31.  A lane whose row or bank is masked off, or whose source is
// out of range or inactive, keeps `old` (bound_ctrl: 0 instead, for an out-of-range source).
EMU_WAVE_OP static int emu_update_dpp_(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    const ::emu::WaveRec &r = emu_meet_((uint32_t) src, EMU_SITE);
    const int lane = emu_lane(), row = lane >> 4, in_row = lane & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) return old;
    int j = -1;
    if (ctrl >= 0 && ctrl <= 0xFF) j = (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int k = in_row + (ctrl - 0x100); j = k < 16 ? row * 16 + k : -1; }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int k = in_row - (ctrl - 0x110); j = k >= 0 ? row * 16 + k : -1; }
    else if (ctrl == 0x140) j = row * 16 + 15 - in_row;
    else if (ctrl == 0x141) j = (lane & ~7) + 7 - (lane & 7);
    else if (ctrl == 0x142) j = row >= 1 ? row * 16 - 1 : -1;
    else if (ctrl == 0x143) j = row >= 2 ? 31 : -1;
    else { fprintf(stderr, "emu: dpp control 0x%x not modelled\n", ctrl); abort(); }
    if (j < 0) return bound_ctrl ? 0 : old;
    if (!((r.mask >> j) & 1)) return old;
    return (int) (uint32_t) r.val[j];
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp_((old), (src), (ctrl), (rm), (bm), (bc))

// lane-local
static inline uint32_t emu_mbcnt_lo_(uint32_t mask, uint32_t v)
{
    const int lane = emu_lane();
    return v + (uint32_t) __builtin_popcount(lane >= 32 ? mask : (mask & ((1u << lane) - 1u)));
}
static inline uint32_t emu_mbcnt_hi_(uint32_t mask, uint32_t v)
{
    const int lane = emu_lane();
    return v + (uint32_t) (lane > 32 ? __builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0);
}
#define __builtin_amdgcn_mbcnt_lo(m, v) emu_mbcnt_lo_((m), (v))
#define __builtin_amdgcn_mbcnt_hi(m, v) emu_mbcnt_hi_((m), (v))
static inline uint32_t emu_bitop3_(uint32_t a, uint32_t b, uint32_t c, uint32_t tt)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) {
        const int idx = (int) ((((a >> i) & 1) << 2) | (((b >> i) & 1) << 1) | ((c >> i) & 1));
        r |= ((tt >> idx) & 1u) << i;
    }
    return r;
}
#define __builtin_amdgcn_bitop3_b32(a, b, c, tt) emu_bitop3_((a), (b), (c), (tt))
static inline int emu_sbfe_(int v, int off, int width)
{
    const uint32_t u = ((uint32_t) v >> off) & (width >= 32 ? ~0u : ((1u << width) - 1u));
    const uint32_t sign = 1u << (width - 1);
    return (int) ((u ^ sign) - sign);
}
#define __builtin_amdgcn_sbfe(v, o, w) emu_sbfe_((v), (o), (w))

// vector types and bit casts of the HIP headers
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(16) double2 { double x, y; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline long long __double_as_longlong(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline double __longlong_as_double(long long i) { double d; memcpy(&d, &i, 8); return d; }
static inline int __double2loint(double d) { long long i; memcpy(&i, &d, 8); return (int) (unsigned) (i & 0xffffffffll); }
static inline int __double2hiint(double d) { long long i; memcpy(&i, &d, 8); return (int) (unsigned) ((unsigned long long) i >> 32); }
static inline double __hiloint2double(int hi, int lo)
{
    const unsigned long long u = ((unsigned long long) (unsigned) hi << 32) | (unsigned) lo;
    double d; memcpy(&d, &u, 8); return d;
}

// device-side min / max of the HIP headers
template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, int b) { return a < (unsigned) b ? a : (unsigned) b; }
static inline unsigned min(int a, unsigned b) { return (unsigned) a < b ? (unsigned) a : b; }

static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long) v) : 64; }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned) v) : 32; }

// ---- atomics (workgroups run one after the other, fibers one at a time: plain operations) -------
// (kept out of the load / store instrumentation below: an atomic is one indivisible access)
#define EMU_ATOMIC __attribute__((noinline, no_sanitize("coverage"))) static
template <class T> EMU_ATOMIC T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
EMU_ATOMIC unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { auto o = *p; *p = o + v; return o; }
template <class T> EMU_ATOMIC T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <class T> EMU_ATOMIC T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <class T> EMU_ATOMIC T atomicMax(T *p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <class T> EMU_ATOMIC T atomicMin(T *p, T v) { T o = *p; *p = o < v ? o : v; return o; }
#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void) (*(p) = (v)))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))

// ---- runtime ------------------------------------------------------------------------------------
typedef int hipError_t;
typedef void *hipStream_t;
typedef struct EmuEvent { double t; } *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; char name[64]; char gcnArchName[64]; int warpSize; };

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void *p);
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated HIP error"; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e);
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr);
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int dev);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
