// Context, error strings, caching device allocator.
#include "bt_common.hpp"

#include <algorithm>
#include <cstdarg>

namespace bt {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// Requests are rounded up to a size class (16 classes per power of two, at most
// 6.25 % over the request) and a cached block is reused only for its own class.
// The builders ask for the same sizes in the same order on every call, so after
// one repetition every class holds as many blocks as are ever live at once and no
// call reaches hipMalloc again.  (A best-fit search with a tolerance instead lets a
// request take a block that a later, larger request needs, which then allocates --
// a cascade that costs a multi-millisecond hipMalloc per step for many steps.)
static size_t size_class(size_t bytes)
{
    bytes = (bytes + 255) & ~(size_t) 255;
    int bits = 0;
    for (size_t v = bytes; v; v >>= 1) ++bits;
    if (bits <= 12) return bytes;
    const int shift = bits - 5;
    return ((bytes + ((size_t) 1 << shift) - 1) >> shift) << shift;
}

int Pool::alloc(void **out, size_t bytes)
{
    bytes = size_class(bytes);
    for (auto &b : blocks_)
        if (!b.used && b.size == bytes) {
            b.used = true;
            *out = b.ptr;
            return BT_OK;
        }
    // a new block is needed.  If the idle blocks (left over from calls with other
    // sizes) outweigh what is in use, give them back first so that a long-lived
    // context does not hoard memory other allocators in the process could use.
    {
        size_t idle = 0, busy = bytes;
        for (auto &b : blocks_) (b.used ? busy : idle) += b.size;
        // (the slack grows with every purge: call sequences whose stages take turns at the
        // pool -- exchange buffers, then the build's -- would otherwise free and re-allocate
        // gigabytes on every step: 135 ms per build at 1.25*10^8 points)
        if (idle > 2 * busy + purge_slack_) {
            // ... up to a quarter of the device: beyond that the trim could never fire again
            // and other allocators in the process (torch) would starve beside an idle cache
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); total_b = (size_t) 64 << 30; }
            purge_slack_ = std::min(purge_slack_ * 4, std::max(total_b / 4, (size_t) 1 << 30));
            std::vector<Block> keep;
            for (auto &b : blocks_) {
                if (b.used) keep.push_back(b);
                else { (void) hipFree(b.ptr); reserved_ -= b.size; }
            }
            blocks_.swap(keep);
        }
    }
    void *p = nullptr;
    static const bool trace = [] { const char *t = getenv("BT_POOL_TRACE"); return t && atoi(t); }();
    if (trace) {
        size_t idle = 0, busy = 0;
        for (auto &b : blocks_) (b.used ? busy : idle) += b.size;
        fprintf(stderr, "[bt-pool] hipMalloc %zu MB (busy %zu MB, idle %zu MB, %zu blocks)\n",
                bytes >> 20, busy >> 20, idle >> 20, blocks_.size());
    }
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        // drop the cache and retry once
        (void) hipGetLastError();
        std::vector<Block> keep;
        for (auto &b : blocks_) {
            if (b.used) keep.push_back(b);
            else { (void) hipFree(b.ptr); reserved_ -= b.size; }
        }
        blocks_.swap(keep);
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void) hipGetLastError();
            set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
            *out = nullptr;
            return BT_ERR_ALLOC;
        }
    }
    blocks_.push_back({p, bytes, true});
    reserved_ += bytes;
    *out = p;
    return BT_OK;
}

void Pool::free(void *p)
{
    for (auto &b : blocks_)
        if (b.ptr == p) { b.used = false; return; }
}

void Pool::release_idle()
{
    size_t k = 0;
    for (size_t i = 0; i < blocks_.size(); ++i) {
        if (blocks_[i].used) blocks_[k++] = blocks_[i];
        else { (void) hipFree(blocks_[i].ptr); reserved_ -= blocks_[i].size; }
    }
    blocks_.resize(k);
}

void Pool::release_all()
{
    for (auto &b : blocks_) (void) hipFree(b.ptr);
    blocks_.clear();
    reserved_ = 0;
}

// descriptor storage and ticket base of the next single-pass scan (bt_prims.hpp)
int scan_prepare(bt_context *ctx, int64_t ntiles, uint32_t *gen, uint32_t *ticket_base)
{
    if (!ctx->scan_ticket) {
        BT_HIP_CHECK(hipMalloc((void **) &ctx->scan_ticket, 256));
        BT_HIP_CHECK(hipMemsetAsync(ctx->scan_ticket, 0, 256, ctx->stream));
        ctx->scan_ticket_base = 0;
    }
    bool fresh = false;
    if (ntiles > ctx->scan_desc_cap) {
        int64_t cap = std::max<int64_t>(4096, ctx->scan_desc_cap);
        while (cap < ntiles) cap *= 2;
        if (ctx->scan_desc) {
            // earlier scans on this stream still read the old array
            BT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            (void) hipFree(ctx->scan_desc);
            ctx->scan_desc = nullptr;
        }
        BT_HIP_CHECK(hipMalloc((void **) &ctx->scan_desc, (size_t) cap * 8));
        ctx->scan_desc_cap = cap;
        fresh = true;
    }
    ctx->scan_gen = (ctx->scan_gen + 1) & ((1u << 22) - 1);
    if (ctx->scan_gen == 0) { ctx->scan_gen = 1; fresh = true; }     // tags wrapped
    if (fresh)
        BT_HIP_CHECK(hipMemsetAsync(ctx->scan_desc, 0, (size_t) ctx->scan_desc_cap * 8, ctx->stream));
    *gen = ctx->scan_gen;
    *ticket_base = ctx->scan_ticket_base;
    ctx->scan_ticket_base += (uint32_t) ntiles;
    return BT_OK;
}

// Device-to-host reads are done by a kernel that stores straight into the pinned block
// (hipHostMalloc memory is mapped into the device's address space): a hipMemcpyAsync is a
// runtime blit with ~6 us of pipeline bubble either side, measured in the kernel trace; a
// kernel queues like any other launch.
__global__ __launch_bounds__(256) void copy_words_kernel(const uint32_t *src, uint32_t *dst, size_t nwords)
{
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < nwords; i += (size_t) gridDim.x * 256)
        dst[i] = src[i];
}

__global__ void fill_word_kernel(uint32_t *p, uint32_t v) { *p = v; }

__global__ __launch_bounds__(256) void copy_bytes_kernel(const unsigned char *src, unsigned char *dst,
                                                         size_t n)
{
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
        dst[i] = src[i];
}

int copy_to_pinned(bt_context *ctx, void *pinned_dst, const void *dev_src, size_t bytes)
{
    if (!ctx->pinned_stores_ok) {
        BT_HIP_CHECK(hipMemcpyAsync(pinned_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return BT_OK;
    }
    if (((uintptr_t) dev_src | (uintptr_t) pinned_dst | bytes) % 4 == 0) {
        const size_t nw = bytes / 4;
        copy_words_kernel<<<(unsigned) std::min<size_t>(64, (nw + 255) / 256), 256, 0, ctx->stream>>>(
            (const uint32_t *) dev_src, (uint32_t *) pinned_dst, nw);
    } else {
        copy_bytes_kernel<<<(unsigned) std::min<size_t>(64, (bytes + 255) / 256), 256, 0, ctx->stream>>>(
            (const unsigned char *) dev_src, (unsigned char *) pinned_dst, bytes);
    }
    BT_HIP_CHECK(hipGetLastError());
    return BT_OK;
}

int d2h(bt_context *ctx, void *host_dst, const void *dev_src, size_t bytes, bool persistent)
{
    if (bytes == 0) return BT_OK;
    const size_t need = (bytes + 15) & ~(size_t) 15;
    if (ctx->h_ring && need > ctx->h_ring_cap - ctx->h_ring_used && !ctx->pending_reads.empty())
        BT_CHECK(sync_stream(ctx));
    if (!ctx->h_ring || need > ctx->h_ring_cap - ctx->h_ring_used) {
        BT_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return BT_OK;
    }
    char *slot = ctx->h_ring + ctx->h_ring_used;
    ctx->h_ring_used += need;
    BT_CHECK(copy_to_pinned(ctx, slot, dev_src, bytes));
    ctx->pending_reads.push_back({host_dst, slot, bytes, ctx->read_seq++, persistent});
    return BT_OK;
}

static int status_verdict(const DeviceStatus *st)
{
    if (st->lookback_timeout) {
        set_error("radix sort: decoupled look-back spin bound exceeded");
        return BT_ERR_INTERNAL;
    }
    if (st->internal == 70) {
        set_error("bt_let_build: a box has no parent among the boxes of the level above, or a "
                  "level is not in ascending Morton order");
        return BT_ERR_INVALID;
    }
    if (st->internal) {
        set_error("device-side consistency check failed (code %d)", st->internal);
        return BT_ERR_INTERNAL;
    }
    if (st->max_levels) {
        set_error("Level count exceeded the depth addressable by the 64-bit Morton key "
                  "(a large number of particles is indistinguishable at that depth).");
        return BT_ERR_MAX_LEVELS;
    }
    return BT_OK;
}

int sync_stream(bt_context *ctx)
{
    hipError_t e = hipStreamSynchronize(ctx->stream);
    for (auto &r : ctx->pending_reads) memcpy(r.dst, r.src, r.bytes);
    ctx->pending_reads.clear();
    ctx->h_ring_used = 0;
    // every status read queued since the last wait has its own slot: the first failure
    // (in the order the calls were made) is the one reported
    const int n = ctx->status_inflight;
    ctx->status_inflight = 0;
    BT_HIP_CHECK(e);
    for (int i = 0; i < n; ++i) {
        int v = status_verdict(&ctx->h_status[i]);
        if (v != BT_OK) return v;
    }
    return BT_OK;
}

// An error exit between a d2h() and its wait must not leave destinations queued: they are
// mostly locals of the frame that is being left.
void drop_pending_reads(bt_context *ctx)
{
    (void) hipStreamSynchronize(ctx->stream);
    ctx->pending_reads.clear();
    ctx->h_ring_used = 0;
    ctx->status_inflight = 0;
}

static int queue_status_read(bt_context *ctx)
{
    // a slot per queued read; a caller that makes more calls than there are slots without
    // ever waiting gets the verdicts so far first
    if (ctx->status_inflight >= STATUS_SLOTS) BT_CHECK(sync_stream(ctx));
    BT_CHECK(copy_to_pinned(ctx, &ctx->h_status[ctx->status_inflight], ctx->d_status, sizeof(DeviceStatus)));
    ++ctx->status_inflight;
    return BT_OK;
}

void *zero_alloc(bt_context *ctx, size_t bytes)
{
    const size_t need = (bytes + 255) & ~(size_t) 255;
    if (!ctx->zero_block || need > ctx->zero_cap - ctx->zero_used) return nullptr;
    void *p = ctx->zero_block + ctx->zero_used;
    ctx->zero_used += need;
    return p;
}

int zero_begin(bt_context *ctx)
{
    // (from offset 0: the status word is reset with the rest)
    if (ctx->zero_block)
        BT_HIP_CHECK(hipMemsetAsync(ctx->zero_block, 0, std::max<size_t>(ctx->zero_used, 256), ctx->stream));
    ctx->zero_used = 256;
    return BT_OK;
}

int finish_call(bt_context *ctx)
{
    if (!ctx->stream_ordered) return check_status(ctx);
    return queue_status_read(ctx);
}

int reset_status(bt_context *ctx)
{
    BT_HIP_CHECK(hipMemsetAsync(ctx->d_status, 0, sizeof(DeviceStatus), ctx->stream));
    return BT_OK;
}

int check_status(bt_context *ctx)
{
    BT_CHECK(queue_status_read(ctx));
    return sync_stream(ctx);
}

}  // namespace bt

void bt_free_tree_state(bt_context *ctx);   // bt_tree.hip
void bt_free_trav_state(bt_context *ctx);   // bt_trav.hip
void bt_free_aq_state(bt_context *ctx);     // bt_area_query.hip
void bt_free_mgpu_state(bt_context *ctx);   // bt_mgpu.hip

extern "C" {

int bt_abi_version(void) { return BT_ABI_VERSION; }

const char *bt_last_error_string(void) { return bt::g_last_error.c_str(); }

int bt_create(int device, void *hip_stream, bt_context **out)
{
    if (!out) { bt::set_error("bt_create: out is NULL"); return BT_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    BT_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        bt::set_error("bt_create: device %d out of range (have %d)", device, ndev);
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(device));
    bt_context *ctx = new bt_context();
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = (hipStream_t) hip_stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; BT_HIP_CHECK(e); }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        ctx->num_cus = prop.multiProcessorCount;
    // the zeroed block of a call; its first 256 bytes are the status word, so that the
    // memset at the entry of a build / traversal clears both
    static_assert(sizeof(bt::DeviceStatus) <= 256, "status word must fit its slot");
    ctx->zero_cap = 2 << 20;
    hipError_t e = hipMalloc((void **) &ctx->zero_block, ctx->zero_cap);
    if (e == hipSuccess) e = hipMemset(ctx->zero_block, 0, ctx->zero_cap);
    if (e == hipSuccess) {
        ctx->d_status = (bt::DeviceStatus *) ctx->zero_block;
        ctx->zero_used = 256;
        e = hipHostMalloc((void **) &ctx->h_status, bt::STATUS_SLOTS * sizeof(bt::DeviceStatus),
                          hipHostMallocDefault);
    }
    if (e == hipSuccess) {
        ctx->h_ring_cap = 256 << 10;
        e = hipHostMalloc((void **) &ctx->h_ring, ctx->h_ring_cap, hipHostMallocDefault);
    }
    if (e != hipSuccess) { bt_destroy(ctx); BT_HIP_CHECK(e); }
    memset(ctx->h_status, 0, bt::STATUS_SLOTS * sizeof(bt::DeviceStatus));
    int s = bt::reset_status(ctx);
    if (s != BT_OK) { bt_destroy(ctx); return s; }
    {
        // can a kernel store into the pinned block, and does the host see it after a wait?
        // (otherwise the reads fall back to hipMemcpyAsync)
        uint32_t *probe = (uint32_t *) ctx->h_ring;
        probe[0] = 0; probe[1] = 0;
        bt::copy_words_kernel<<<1, 256, 0, ctx->stream>>>((const uint32_t *) ctx->d_status, probe + 1, 1);
        bt::fill_word_kernel<<<1, 1, 0, ctx->stream>>>(probe, 0x600DF00Du);
        const bool ok = hipGetLastError() == hipSuccess
                        && hipStreamSynchronize(ctx->stream) == hipSuccess && probe[0] == 0x600DF00Du;
        if (!ok) (void) hipGetLastError();
        ctx->pinned_stores_ok = ok && !getenv("BT_NO_PINNED_STORES");
    }
    *out = ctx;
    return BT_OK;
}

void bt_destroy(bt_context *ctx)
{
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) hipStreamSynchronize(ctx->stream);
    bt_free_tree_state(ctx);
    bt_free_trav_state(ctx);
    bt_free_aq_state(ctx);
    bt_free_mgpu_state(ctx);
    ctx->pool.release_all();
    if (ctx->zero_block) (void) hipFree(ctx->zero_block);     // (holds the status word)
    for (void *&e : ctx->sort_ev)
        if (e) { (void) hipEventDestroy((hipEvent_t) e); e = nullptr; }
    if (ctx->scan_desc) (void) hipFree(ctx->scan_desc);
    if (ctx->scan_ticket) (void) hipFree(ctx->scan_ticket);
    if (ctx->h_status) (void) hipHostFree(ctx->h_status);
    if (ctx->h_ring) (void) hipHostFree(ctx->h_ring);
    if (ctx->own_stream && ctx->stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int bt_set_stream(bt_context *ctx, void *hip_stream)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    if (ctx->stream == (hipStream_t) hip_stream && !ctx->own_stream) return BT_OK;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_CHECK(bt::sync_stream(ctx));
    if (ctx->own_stream && ctx->stream) { (void) hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
    ctx->stream = (hipStream_t) hip_stream;
    return BT_OK;
}

int bt_set_stream_ordered(bt_context *ctx, int on)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    ctx->stream_ordered = on != 0;
    return BT_OK;
}

int bt_set_stage_timing(bt_context *ctx, int on)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    ctx->stage_timing = on != 0;
    return BT_OK;
}

int bt_synchronize(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    return bt::sync_stream(ctx);
}

int bt_release_cached(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    BT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->pool.release_idle();
    return BT_OK;
}

int bt_trim(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx) return BT_ERR_INVALID;
    BT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    bt_free_tree_state(ctx);
    bt_free_trav_state(ctx);
    bt_free_aq_state(ctx);
    bt_free_mgpu_state(ctx);
    ctx->pool.release_all();
    return BT_OK;
}

}  // extern "C"
