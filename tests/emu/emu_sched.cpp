// TEST INFRASTRUCTURE -- fibers and the workgroup scheduler of the HIP emulator (hip_emu.hpp).
#include "hip_emu.hpp"

#include <chrono>
#include <mutex>
#include <vector>

#include <sys/mman.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

namespace emu {

thread_local ThreadCtx *g_cur = nullptr;

namespace {

// ---- context switch (x86-64 SysV): callee-saved registers + stack pointer ----------------------
extern "C" void emu_switch(void **save_sp, void *load_sp);
__asm__(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");

enum State : uint8_t { RUN, WAIT_WAVE, WAIT_MEM, WAIT_BLOCK, DONE };

constexpr size_t STACK_BYTES = 256 << 10;
constexpr size_t DYN_LDS_GUARD = 64 << 10;     // behind the dynamic LDS of a launch: accesses there are reported

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    State st = DONE;
    bool spun = false;
    ThreadCtx ctx{};
    uint64_t dep = 0;               // value deposited at a wave-wide operation
    const void *site = nullptr;
    const WaveRec *rec = nullptr;
};

struct Sched {
    std::vector<Fiber> fibers;      // capacity grows to the largest workgroup seen
    std::vector<WaveRec> recs;      // [wave][64] records of the latest resolution
    std::vector<char> dyn;
    void *main_sp = nullptr;
    Fiber *cur = nullptr;
    KernelThunk fn = nullptr;
    void *arg = nullptr;
};
thread_local Sched *g_s = nullptr;

Sched &S()
{
    if (!g_s) g_s = new Sched();
    return *g_s;
}

void to_scheduler()
{
    Sched &s = S();
    Fiber *f = s.cur;
    emu_switch(&f->sp, s.main_sp);
}

extern "C" void emu_fiber_entry()
{
    Sched &s = S();
    s.fn(s.arg);
    s.cur->st = DONE;
    to_scheduler();
    abort();                        // a finished fiber is never resumed
}

void prepare(Fiber &f)
{
    if (!f.stack) {
        f.stack = (char *) mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == (char *) MAP_FAILED) { perror("emu: mmap of a fiber stack"); abort(); }
    }
    // initial frame as emu_switch leaves one: [mxcsr|fpcw pad][r15 r14 r13 r12 rbx rbp][return address]
    uintptr_t top = ((uintptr_t) f.stack + STACK_BYTES) & ~(uintptr_t) 15;
    uint64_t *sp = (uint64_t *) top;
    *--sp = 0;                                  // alignment: entry sees rsp % 16 == 8 after `ret`
    *--sp = (uint64_t) (uintptr_t) &emu_fiber_entry;
    for (int i = 0; i < 6; ++i) *--sp = 0;      // rbp rbx r12 r13 r14 r15
    uint32_t csr[2] = {0x1F80u, 0x037Fu};       // default MXCSR, x87 control word
    --sp;
    memcpy(sp, csr, 8);
    f.sp = sp;
    f.st = RUN;
    f.spun = false;
}

void resume(Sched &s, Fiber &f)
{
    s.cur = &f;
    g_cur = &f.ctx;
    emu_switch(&s.main_sp, f.sp);
    s.cur = nullptr;
    g_cur = nullptr;
}

}  // namespace

const WaveRec &wave_meet(uint64_t v, const void *site)
{
    Sched &s = S();
    Fiber *f = s.cur;
    f->dep = v; f->site = site; f->st = WAIT_WAVE;
    f->ctx.lds_phase = 0;
    to_scheduler();
    return *f->rec;
}

void block_barrier()
{
    Sched &s = S();
    s.cur->st = WAIT_BLOCK;
    s.cur->ctx.lds_phase = 0;
    to_scheduler();
}

// Lock step within a wave, as far as LDS goes.  On the hardware the lanes of a wave execute one
// instruction together: every lane's LDS load of instruction k happens before any lane's LDS store
// of instruction k + 1, and the other way round.  Fibers run one lane at a time, so a lane that
// turns from loading LDS to storing it (or from storing to loading) first lets every other lane of
// its wave catch up to a meeting point of its own.  (Called by the compiler's load / store hooks.)
extern "C" char __start_emu_lds[], __stop_emu_lds[];
static inline void lds_access(const void *addr, int kind /* 1 load, 2 store */)
{
    ThreadCtx *c = g_cur;
    if (!c) return;
    const char *a = (const char *) addr;
    const char *dyn = (const char *) c->dyn_lds;
    const bool lds = (a >= __start_emu_lds && a < __stop_emu_lds) || (a >= dyn && a < dyn + c->dyn_lds_bytes);
    if (!lds) {
        // an access just behind the dynamic LDS the launch asked for: on the hardware that is another
        // workgroup's memory or a fault
        if (a >= dyn + c->dyn_lds_bytes && a < dyn + c->dyn_lds_bytes + DYN_LDS_GUARD) {
            fprintf(stderr, "emu: thread (%u,%u,%u) of workgroup (%u,%u,%u) touches dynamic LDS at byte %zu, "
                    "the launch asked for %zu\n", c->tid.x, c->tid.y, c->tid.z, c->bid.x, c->bid.y, c->bid.z,
                    (size_t) (a - dyn), c->dyn_lds_bytes);
            abort();
        }
        return;
    }
    if (c->lds_phase != 0 && c->lds_phase != kind) {
        Sched &s = S();
        s.cur->st = WAIT_MEM;
        to_scheduler();
    }
    c->lds_phase = kind;
}

void spin_yield()
{
    Sched &s = S();
    s.cur->spun = true;
    to_scheduler();
}

static void on_segv(int sig, siginfo_t *si, void *)
{
    void *bt[48];
    const int n = backtrace(bt, 48);
    fprintf(stderr, "[emu] signal %d at address %p; backtrace:\n", sig, si->si_addr);
    backtrace_symbols_fd(bt, n, 2);
    _exit(139);
}

static int order_mode()
{
    static const int m = [] {
        const char *e = getenv("EMU_ORDER");
        if (!e) return 0;
        if (!strcmp(e, "reverse")) return 1;
        if (!strncmp(e, "shuffle", 7)) return 2;
        return 0;
    }();
    return m;
}
static uint64_t order_next()
{
    static thread_local uint64_t x = [] {
        const char *e = getenv("EMU_ORDER");
        const char *c = e ? strchr(e, ':') : nullptr;
        return (uint64_t) (c ? atoll(c + 1) : 1) * 0x9E3779B97F4A7C15ull + 12345u;
    }();
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return x;
}

void trace_launch(const char *name, dim3 grid, dim3 block, size_t lds)
{
    static const bool on = [] {
        const char *e = getenv("EMU_TRACE");
        const bool t = e && atoi(e);
        if (t) {
            static char altstack[1 << 16];
            stack_t ss{altstack, 0, sizeof altstack};
            sigaltstack(&ss, nullptr);
            struct sigaction sa{};
            sa.sa_sigaction = on_segv;
            sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
            sigaction(SIGSEGV, &sa, nullptr);
            sigaction(SIGBUS, &sa, nullptr);
        }
        return t;
    }();
    if (on) fprintf(stderr, "[emu] %s <<<(%u,%u,%u), (%u,%u,%u), %zu>>>\n", name, grid.x, grid.y, grid.z, block.x, block.y, block.z, lds);
}

void run_grid(dim3 grid, dim3 block, size_t dyn_lds_bytes, KernelThunk fn, void *arg)
{
    // (LDS variables are process-wide statics: one kernel at a time, whichever host thread launches)
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> launch_lock(launch_mutex);
    Sched &s = S();
    if (s.cur) { fprintf(stderr, "emu: nested kernel launch\n"); abort(); }
    const unsigned nthreads = block.x * block.y * block.z;
    if (nthreads == 0 || (uint64_t) grid.x * grid.y * grid.z == 0) return;
    if (nthreads > 1024) { fprintf(stderr, "emu: workgroup of %u threads\n", nthreads); abort(); }
    if (s.fibers.size() < nthreads) s.fibers.resize(nthreads);
    const unsigned nwaves = (nthreads + 63) / 64;
    if (s.recs.size() < (size_t) nwaves * 64) s.recs.resize((size_t) nwaves * 64);
    s.dyn.assign(dyn_lds_bytes + 64 + DYN_LDS_GUARD, 0);
    void *dyn = (void *) (((uintptr_t) s.dyn.data() + 63) & ~(uintptr_t) 63);
    s.fn = fn; s.arg = arg;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber &f = s.fibers[t];
            prepare(f);
            f.ctx.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            f.ctx.bid = {bx, by, bz};
            f.ctx.bdim = {block.x, block.y, block.z};
            f.ctx.gdim = {grid.x, grid.y, grid.z};
            f.ctx.lane = (int) (t & 63); f.ctx.wave = (int) (t >> 6);
            f.ctx.dyn_lds = dyn; f.ctx.dyn_lds_bytes = dyn_lds_bytes; f.ctx.lds_phase = 0;
        }
        unsigned done = 0;
        uint64_t idle_passes = 0;
        while (done < nthreads) {
            bool progress = false;
            for (unsigned wi = 0; wi < nwaves; ++wi) {
                // EMU_ORDER=reverse | shuffle: the order in which the waves of a workgroup, and the
                // lanes of a wave, get their turn between two meeting points -- nothing a correct
                // kernel may depend on
                unsigned w = wi;
                if (order_mode() == 1) w = nwaves - 1 - wi;
                else if (order_mode() == 2) w = (wi + (unsigned) (order_next() % nwaves)) % nwaves;
                const unsigned lo = w * 64, hi = std::min(nthreads, lo + 64);
                for (;;) {
                    const unsigned rot = order_mode() == 2 ? (unsigned) (order_next() & 63u) : 0u;
                    for (unsigned ti = 0; ti < hi - lo; ++ti) {
                        unsigned t = lo + ti;
                        if (order_mode() == 1) t = hi - 1 - ti;
                        else if (order_mode() == 2) t = lo + (ti + rot) % (hi - lo);
                        Fiber &f = s.fibers[t];
                        if (f.st != RUN) continue;
                        f.spun = false;
                        resume(s, f);
                        if (f.st == DONE) ++done;
                        if (!(f.st == RUN && f.spun)) progress = true;
                    }
                    bool any_run = false, any_wave = false, any_mem = false;
                    for (unsigned t = lo; t < hi; ++t) {
                        any_run = any_run || s.fibers[t].st == RUN;
                        any_wave = any_wave || s.fibers[t].st == WAIT_WAVE;
                        any_mem = any_mem || s.fibers[t].st == WAIT_MEM;
                    }
                    if (any_run) break;           // lanes that sleep in a spin loop: let the other waves run
                    if (any_mem) {
                        // every lane has caught up: the lanes at a turn of their LDS traffic go on
                        // (before any wave-wide operation completes: they may be on their way to it)
                        for (unsigned t = lo; t < hi; ++t)
                            if (s.fibers[t].st == WAIT_MEM) s.fibers[t].st = RUN;
                        progress = true;
                        continue;
                    }
                    if (!any_wave) break;         // all finished or at the workgroup's barrier
                    // every unfinished lane of the wave waits: the lanes at one call site meet
                    bool grouped[64] = {};
                    unsigned nrec = 0;
                    for (unsigned t = lo; t < hi; ++t) {
                        Fiber &f = s.fibers[t];
                        if (f.st != WAIT_WAVE || grouped[t - lo]) continue;
                        WaveRec &r = s.recs[(size_t) w * 64 + nrec++];
                        r.mask = 0;
                        for (unsigned u = t; u < hi; ++u) {
                            Fiber &g = s.fibers[u];
                            if (g.st == WAIT_WAVE && !grouped[u - lo] && g.site == f.site) {
                                grouped[u - lo] = true;
                                r.mask |= 1ull << (u - lo);
                                r.val[u - lo] = g.dep;
                                g.rec = &r;
                            }
                        }
                    }
                    for (unsigned t = lo; t < hi; ++t) if (grouped[t - lo]) s.fibers[t].st = RUN;
                    progress = true;
                }
            }
            // the workgroup's barrier: every unfinished thread waits at it
            unsigned waiting = 0;
            for (unsigned t = 0; t < nthreads; ++t) waiting += s.fibers[t].st == WAIT_BLOCK;
            if (waiting && waiting + done == nthreads) {
                for (unsigned t = 0; t < nthreads; ++t) if (s.fibers[t].st == WAIT_BLOCK) s.fibers[t].st = RUN;
                progress = true;
            }
            if (!progress) {
                if (++idle_passes > 2000000) {
                    fprintf(stderr, "emu: workgroup (%u,%u,%u) makes no progress: %u of %u threads done, %u at the barrier "
                            "(a wave operation under divergence, or a spin on a later workgroup)\n", bx, by, bz, done, nthreads, waiting);
                    abort();
                }
            } else {
                idle_passes = 0;
            }
        }
    }
    s.fn = nullptr; s.arg = nullptr;
}

}  // namespace emu

// ---- the compiler's load / store hooks (-fsanitize-coverage=trace-loads,trace-stores) ---------------
extern "C" {
void __sanitizer_cov_load1(const void *a) { emu::lds_access(a, 1); }
void __sanitizer_cov_load2(const void *a) { emu::lds_access(a, 1); }
void __sanitizer_cov_load4(const void *a) { emu::lds_access(a, 1); }
void __sanitizer_cov_load8(const void *a) { emu::lds_access(a, 1); }
void __sanitizer_cov_load16(const void *a) { emu::lds_access(a, 1); }
void __sanitizer_cov_store1(const void *a) { emu::lds_access(a, 2); }
void __sanitizer_cov_store2(const void *a) { emu::lds_access(a, 2); }
void __sanitizer_cov_store4(const void *a) { emu::lds_access(a, 2); }
void __sanitizer_cov_store8(const void *a) { emu::lds_access(a, 2); }
void __sanitizer_cov_store16(const void *a) { emu::lds_access(a, 2); }
void __sanitizer_cov_trace_pc_guard(uint32_t *) {}
void __sanitizer_cov_trace_pc_guard_init(uint32_t *, uint32_t *) {}
}

// ---- runtime ------------------------------------------------------------------------------------
static double emu_now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
// Device memory comes back POISONED (0xA5 bytes; EMU_POISON=0: as malloc leaves it): hipMalloc does not
// clear memory, and fresh pages from the host allocator are zero more often than not -- a kernel
// that relies on zeros it never wrote would pass here and fail on the hardware.
static bool emu_poison()
{
    static const bool on = [] { const char *e = getenv("EMU_POISON"); return !e || atoi(e); }();
    return on;
}
hipError_t hipMalloc(void **p, size_t n)
{
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    if (emu_poison()) memset(q, 0xA5, n ? n : 1);
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent{0.0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float) (b->t - a->t); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    p->multiProcessorCount = 8;         // (grids sized by the CU count stay small)
    p->totalGlobalMem = (size_t) 16 << 30;
    p->warpSize = 64;
    snprintf(p->name, sizeof p->name, "HIP emulator (CPU fibers)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "emu");
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) { *free_b = (size_t) 8 << 30; *total_b = (size_t) 16 << 30; return hipSuccess; }
