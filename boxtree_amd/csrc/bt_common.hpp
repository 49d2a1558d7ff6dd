// boxtree_amd -- MI355X (gfx950) native tree build + FMM traversal.
// Common host-side plumbing: error handling, the context, a caching device
// allocator.  Wave size is 64 everywhere (CDNA4).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/boxtree_hip.h"

namespace bt {

constexpr int WAVE = 64;

void set_error(const char *fmt, ...);

#define BT_HIP_CHECK(expr)                                                        \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) {                                                   \
            ::bt::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,         \
                            hipGetErrorString(e_));                               \
            return BT_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

#define BT_CHECK(expr)                                                            \
    do {                                                                          \
        int s_ = (expr);                                                          \
        if (s_ != BT_OK) return s_;                                               \
    } while (0)

inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// BT_HOST_TRACE=1: host-side timestamps (CLOCK_MONOTONIC, microseconds -- the clock of
// Python's time.monotonic_ns) of the stage marks, to see where the host waits.
inline void host_trace(const char *name)
{
    static const bool on = [] { const char *e = getenv("BT_HOST_TRACE"); return e && atoi(e); }();
    if (!on) return;
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    fprintf(stderr, "[bt-host] %-14s %.1f\n", name, ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3);
}

// Device error flags polled by the host at its sync points.
struct DeviceStatus {
    int lookback_timeout;   // onesweep look-back spin bound hit
    int max_levels;         // a box at the deepest representable level must split
    int internal;           // assertion-like failures
    int pad;
};

constexpr int STATUS_SLOTS = 16;

// Caching device allocator: blocks are kept until bt_destroy / trim.  All
// allocations are 256-byte aligned (hipMalloc guarantees it).
class Pool {
public:
    ~Pool() { release_all(); }
    int alloc(void **out, size_t bytes);
    void free(void *p);
    void release_all();
    void release_idle();     // blocks that are not in use go back to the device
    size_t bytes_reserved() const { return reserved_; }

private:
    struct Block { void *ptr; size_t size; bool used; };
    std::vector<Block> blocks_;
    size_t reserved_ = 0;
    size_t purge_slack_ = (size_t) 1 << 30;
};

// RAII handle on a pool allocation.
template <class T>
class Buf {
public:
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    Buf(Buf &&o) noexcept { *this = std::move(o); }
    Buf &operator=(Buf &&o) noexcept {
        if (this != &o) {
            reset();
            pool_ = o.pool_; p_ = o.p_; n_ = o.n_;
            o.p_ = nullptr; o.n_ = 0; o.pool_ = nullptr;
        }
        return *this;
    }
    ~Buf() { reset(); }
    int alloc(Pool &pool, int64_t n) {
        reset();
        pool_ = &pool;
        n_ = n;
        void *p = nullptr;
        int s = pool.alloc(&p, (size_t) (n > 0 ? n : 1) * sizeof(T));
        p_ = (T *) p;
        return s;
    }
    void reset() {
        if (p_ && pool_) pool_->free(p_);
        p_ = nullptr; n_ = 0;
    }
    // memory owned by someone else (the caller's output block): never freed here
    void set_external(T *p, int64_t n) {
        reset();
        pool_ = nullptr; p_ = p; n_ = n;
    }
    T *get() const { return p_; }
    int64_t size() const { return n_; }
    void swap(Buf &o) { std::swap(pool_, o.pool_); std::swap(p_, o.p_); std::swap(n_, o.n_); }

private:
    Pool *pool_ = nullptr;
    T *p_ = nullptr;
    int64_t n_ = 0;
};

}  // namespace bt

struct TreeState;   // bt_tree.hip
struct TravState;   // bt_trav.hip
struct AqState;     // bt_area_query.hip
struct MgpuState;   // bt_mgpu.hip

struct bt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    bt::Pool pool;
    bt::DeviceStatus *d_status = nullptr;     // device
    bt::DeviceStatus *h_status = nullptr;     // pinned host mirror: STATUS_SLOTS of them, one per
                                              // status read queued since the last wait
    TreeState *tree = nullptr;
    TravState *trav = nullptr;
    AqState *aq = nullptr;
    MgpuState *mgpu = nullptr;
    // timing of the last bt_radix_sort call (HIP events on ctx->stream)
    float last_sort_pass_ms = 0.f;
    int last_sort_passes = 0;
    // the next radix sort's histogram pass also copies its input keys here (consumed by
    // that call): the histogram reads every key anyway
    void *sort_copy_keys = nullptr;
    int64_t last_sort_n = 0;
    // same, for the last 64-bit-key sort (the tree build's main sort)
    float last_sort64_pass_ms = 0.f;
    int last_sort64_passes = 0;
    int64_t last_sort64_n = 0;
    float stage_ms[32] = {0};
    // single-pass scan (bt_prims.hpp): tile descriptors tagged with a generation,
    // one never-reset ticket counter; both survive from call to call
    uint64_t *scan_desc = nullptr;
    int64_t scan_desc_cap = 0;
    uint32_t *scan_ticket = nullptr;
    uint32_t scan_ticket_base = 0;
    uint32_t scan_gen = 0;
    // sort timing is resolved lazily (bt_get_sort_stats): events of the last sort
    void *sort_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int sort_ev_passes = 0, sort_ev_first_identity = 0;
    int sort_ev_key_bytes = 0;       // bytes per element a pass reads (and writes): 12 = (u64, u32)
                                     // pairs, 8 = keys only (the id packed into the key)
    int sort_ev_digit_bits = 8;
    int64_t sort_ev_n = 0;
    bool sort_ev_pending = false;
    // statistics of the last build (host-side counters)
    int n_host_syncs = 0;
    // small device-to-host reads (bt::d2h / bt::sync_stream): a pinned staging block and
    // the copies waiting for the next synchronisation
    char *h_ring = nullptr;
    size_t h_ring_cap = 0, h_ring_used = 0;
    struct PendingRead { void *dst; const char *src; size_t bytes; uint64_t seq; bool persistent; };
    uint64_t read_seq = 0;
    std::vector<PendingRead> pending_reads;
    // stream-ordered results (bt_set_stream_ordered): calls end with finish_call(), which
    // queues the status read instead of waiting for it; sync_stream examines it later
    bool stream_ordered = false;
    int status_inflight = 0;         // status reads queued since the last wait (slots of h_status)
    // a block of zeroed device memory for the small counters and flags of a call
    // (bt::zero_alloc): one memset per API call instead of one per counter
    char *zero_block = nullptr;
    size_t zero_cap = 0, zero_used = 0;
    bool stage_timing = true;        // record the per-stage events (bt_set_stage_timing)
    bool pinned_stores_ok = false;   // kernels may store into h_ring / h_status (probed at creation)
};

namespace bt {
int check_status(bt_context *ctx);   // sync + read device status flags
int reset_status(bt_context *ctx);
// Read `bytes` from device memory into `host_dst`; the value is there after the next
// sync_stream(ctx).  hipMemcpyAsync into pageable memory blocks the host for a staged copy
// (30-80 us of idle GPU per read, measured); here the copy goes to pinned memory as a queued
// command and any number of reads share one wait.
// `persistent`: the destination outlives the API call (context-owned); all other destinations
// are forgotten when the API call that queued them returns (CallScope), so an error exit
// between a d2h and its wait cannot make a later wait write into a dead stack frame.
int d2h(bt_context *ctx, void *host_dst, const void *dev_src, size_t bytes, bool persistent = false);
// First line of every extern "C" entry point that takes a context.
struct CallScope {
    bt_context *ctx;
    uint64_t seq0;
    explicit CallScope(bt_context *c) : ctx(c), seq0(c ? c->read_seq : 0) {}
    CallScope(const CallScope &) = delete;
    CallScope &operator=(const CallScope &) = delete;
    ~CallScope() {
        if (!ctx || ctx->pending_reads.empty()) return;
        auto &v = ctx->pending_reads;
        size_t k = 0;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].persistent || v[i].seq < seq0) v[k++] = v[i];
        v.resize(k);
    }
};
void drop_pending_reads(bt_context *ctx);   // error exits: wait, forget queued reads and verdicts
int sync_stream(bt_context *ctx);    // hipStreamSynchronize + delivery of the pending reads
                                     // + the verdict on a status read queued by finish_call
// Zeroed device memory for the duration of the current API call (256-byte aligned), or
// nullptr if the block is used up (the caller then allocates and clears its own).
// zero_begin, at the entry of an API call, clears what the previous call used.
void *zero_alloc(bt_context *ctx, size_t bytes);
int zero_begin(bt_context *ctx);
int finish_call(bt_context *ctx);    // end of an API call: check_status, or (stream-ordered
                                     // contexts) queue the status read and return
// Copy device memory into PINNED host memory as a queued command (a kernel that stores into
// the pinned block where the platform allows it, hipMemcpyAsync otherwise).
int copy_to_pinned(bt_context *ctx, void *pinned_dst, const void *dev_src, size_t bytes);

// ---- pieces of the multi-GPU exchange that live beside the single-GPU kernels they share
// code with (bt_tree.hip, bt_shard.hip); callers: bt_mgpu.hip
int bbox_minmax_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                       const void *radii, int64_t n, double *d_mm);
// ... for particles with extents: the cell of a particle that sticks out of the boxes of the top
// levels is the first cell under the box it stays in; hist_stay counts those per top box
// (levels 0..level, index (C^l - 1) / (C - 1) + path)
int morton_cells_ext_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords,
                            const void *radii, int64_t n, const void *d_rootbox, int level,
                            double stick_out_factor, int extent_norm, uint32_t *cells_out,
                            int32_t *hist_cells, int32_t *hist_stay, const int32_t *weights,
                            int64_t *hist_stay_weight);
// bt_morton_cells with the root box read from device memory ({min[3], max[3], extent, 0} in the
// coordinate type, the layout of the tree build's own root box); no wait
int morton_cells_device(bt_context *ctx, int dims, int coord_kind, const void *const *coords, int64_t n,
                        const void *d_rootbox, int level, uint32_t *cells_out, int32_t *hist_inout);
// bt_box_morton_paths / bt_let_build without their waits (errors raise the device status)
int box_paths_device(bt_context *ctx, int dims, int coord_kind, int64_t nboxes, int64_t aligned_nboxes,
                     const void *box_centers, const uint8_t *box_levels, const double *bbox_min,
                     double root_extent, uint64_t *paths);
int let_link_device(bt_context *ctx, int dims, int coord_kind, int nlevels, const int32_t *level_start_box_nrs,
                    const uint64_t *paths, int64_t aligned_nboxes, const double *bbox_min,
                    const double *bbox_max, double root_extent, int32_t *box_parent_ids,
                    int32_t *box_child_ids, void *box_centers);
// whist[cell] += refine weight of every particle (NULL: 1); 32-bit weights as values of the
// coordinates' width for the partition
int weight_hist_device(bt_context *ctx, const uint32_t *cells, const int32_t *weights, int64_t n,
                       int ncells, int64_t *whist);
int widen_weights_device(bt_context *ctx, const int32_t *weights, int64_t n, int elem_size, void *out);
// bt_partition_pack whose own-segment offsets are read from device memory when the kernel runs
// (d_self_offsets = {send offset, receive offset} in records); ncells: length of owner_of_cell
// (the table goes to LDS if it fits), 0 if unknown; no wait
int partition_pack_device(bt_context *ctx, int dims, int elem_size, const void *const *in,
                          const uint32_t *cells, int64_t n, const int32_t *owner_of_cell, int ncells, int nranks,
                          int self_rank, const int64_t *d_self_offsets, void *send, void *recv,
                          Buf<uint8_t> *keep_owners = nullptr, Buf<int32_t> *keep_offsets = nullptr);
// Any per-particle array (4- or 8-byte elements) over the send plan a partition_pack_device call
// kept (owners, offsets): forward, `lay` [n] gets the elements in the send layout (in == nullptr:
// the values iota_base + i) and `self` (owner order, or nullptr) the own segment at self_delta;
// reverse, out[i] = the element at particle i's place.  No wait.
int route_device(bt_context *ctx, int elem_size, bool reverse, const void *in, int64_t iota_base, void *lay,
                 void *self, void *out, const uint8_t *owners, const int32_t *offsets, int64_t n, int nranks,
                 int self_rank, int64_t self_delta);
}  // namespace bt
