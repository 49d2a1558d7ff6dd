// Multi-GPU entries of the C ABI: a sharded tree build and the lists of a rank's own boxes,
// one process (or thread) per GPU.  The reference builds its tree on
// one rank and broadcasts it (boxtree/distributed/__init__.py:183-199); nothing here has a
// counterpart upstream.
//
//   1. bounding box: local min/max, all-reduce(min) over (min, -max)
//      -> the same root box on every rank (host arithmetic of tree_build.py:462-476);
//   2. level-k Morton-cell histogram, all-reduce(sum) -> every rank derives the same
//      top of the global tree and the same owner of every cell (bt_mgpu_plan);
//   3. stable partition by owner that carries the coordinates (interleaved), one grouped
//      send/recv round per 512 MiB of the largest peer message (a rank's own segment is
//      a device copy): the all-to-all-v over the point-to-point xGMI links;
//   4. the caller builds its subtrees with bt_tree_build on the returned shard
//      (sources = points + axis, source_stride = dims, bbox_*, top_level,
//      top_cell_prefix);
//   5. bt_mgpu_number: all-gather of the per-level box counts -> the numbers the rank's
//      boxes carry in the global (single-GPU) tree, global level starts, particle offsets;
//   6. bt_mgpu_let_build / _export: the local essential tree -- shared top levels (from the
//      plan, no communication), the rank's own subtrees, and the subtrees of other ranks'
//      cells within well_sep_is_n_away cells of its own, which their owners send as 16-byte
//      records (Morton path, level, flags, global number) in one all-to-all-v.  Boxes are
//      ordered by (level, Morton path) with one onesweep sort and linked by path lookup
//      (bt_shard.hip), so the LET is the global tree restricted to its boxes.
//
// Collectives go through a small communicator object (bt_mgpu_comm): RCCL (bound at run
// time by dlopen of the librccl.so already in the process, no link-time dependency), or
// ranks that are threads of one process and trade device pointers through a shared table
// -- the latter exists so that the multi-rank logic of steps 1-6 can run, and be compared
// with the single-GPU tree, on a box with one GPU (RCCL refuses two ranks on a device).
#include "bt_common.hpp"
#include "bt_prims.hpp"
#include "bt_sort.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

using namespace bt;

namespace {

// the few RCCL entry points used (signatures of rccl.h; the types are plain C)
typedef void *nccl_comm_t;
enum { NCCL_INT64 = 4, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_UINT8 = 1 };
enum { NCCL_SUM = 0, NCCL_MIN = 3 };
struct Nccl {
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

// The communicator a caller hands over was made by ONE loaded image of RCCL, and its entry
// points must come from that image: the caller may name it (bt_mgpu_use_rccl_library, before
// the first communicator), otherwise an image already in the process is taken (RTLD_NOLOAD;
// a library loaded under another path is found by its SONAME librccl.so.1) and only then a
// fresh one is loaded.
std::string &nccl_library_path()
{
    static std::string p;
    return p;
}

Nccl &nccl()
{
    static Nccl n = [] {
        Nccl r;
        void *h = nullptr;
        if (!nccl_library_path().empty())
            h = dlopen(nccl_library_path().c_str(), RTLD_NOW | RTLD_GLOBAL);
        for (int pass = 0; pass < 2 && !h; ++pass)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (h) break;
            }
        if (!h) return r;
        r.AllReduce = (decltype(r.AllReduce)) dlsym(h, "ncclAllReduce");
        r.AllGather = (decltype(r.AllGather)) dlsym(h, "ncclAllGather");
        r.Send = (decltype(r.Send)) dlsym(h, "ncclSend");
        r.Recv = (decltype(r.Recv)) dlsym(h, "ncclRecv");
        r.GroupStart = (decltype(r.GroupStart)) dlsym(h, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd)) dlsym(h, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString)) dlsym(h, "ncclGetErrorString");
        r.ok = r.AllReduce && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
        return r;
    }();
    return n;
}

#define BT_NCCL_CHECK(expr)                                                       \
    do {                                                                          \
        int e_ = (expr);                                                          \
        if (e_ != 0) {                                                            \
            ::bt::set_error("%s:%d: %s -> RCCL error %d (%s)", __FILE__, __LINE__, #expr, e_, \
                            nccl().GetErrorString ? nccl().GetErrorString(e_) : "?"); \
            return BT_ERR_INTERNAL;                                               \
        }                                                                         \
    } while (0)

#define BT_PEER_BARRIER(g)                                                        \
    do {                                                                          \
        if (!(g)->barrier()) {                                                    \
            ::bt::set_error("a peer rank of the local group failed");              \
            return BT_ERR_INTERNAL;                                               \
        }                                                                         \
    } while (0)

constexpr int64_t MESSAGE_LIMIT_BYTES = (int64_t) 512 << 20;   // see DESIGN.md (RCCL, > 1 GB)

// ranks as threads of one process: a table of pointers and a generation barrier
struct LocalGroup {
    int n = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    std::vector<const void *> ptr;
    std::vector<const int64_t *> off, cnt;
    bool failed = false;          // a rank left a collective entry with an error: the others
                                  // must not wait for it
    // false: a peer has failed (the caller returns an error too)
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (failed) return false;
        const uint64_t g = gen;
        if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g || failed; });
        return !failed;
    }
    void fail()
    {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        cv.notify_all();
    }
};

}  // namespace

struct bt_mgpu_comm {
    int kind = 0;                 // 0: RCCL, 1: threads of one process
    int rank = 0, nranks = 1;
    nccl_comm_t nccl = nullptr;
    LocalGroup *group = nullptr;
    bool self_loopback = false;   // RCCL: a rank's message to itself travels as ncclSend/ncclRecv too
};

namespace {

enum { RED_SUM_I64 = 0, RED_MIN_F64 = 1 };

// in-place all-reduce of a small device array (8-byte elements)
int comm_all_reduce(bt_mgpu_comm *c, hipStream_t stream, void *dev, size_t count, int what)
{
    if (c->kind == 0) {
        Nccl &nc = nccl();
        BT_NCCL_CHECK(nc.AllReduce(dev, dev, count, what == RED_SUM_I64 ? NCCL_INT64 : NCCL_FLOAT64,
                                   what == RED_SUM_I64 ? NCCL_SUM : NCCL_MIN, c->nccl, stream));
        return BT_OK;
    }
    LocalGroup *g = c->group;
    std::vector<int64_t> mine(count), res(count);
    BT_HIP_CHECK(hipMemcpyAsync(mine.data(), dev, count * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    g->ptr[c->rank] = mine.data();
    BT_PEER_BARRIER(g);
    for (size_t i = 0; i < count; ++i) {
        if (what == RED_SUM_I64) {
            int64_t s = 0;
            for (int q = 0; q < c->nranks; ++q) s += ((const int64_t *) g->ptr[q])[i];
            res[i] = s;
        } else {
            double s = ((const double *) g->ptr[0])[i];
            for (int q = 1; q < c->nranks; ++q) s = std::min(s, ((const double *) g->ptr[q])[i]);
            memcpy(&res[i], &s, 8);
        }
    }
    BT_PEER_BARRIER(g);           // every rank has read every vector
    BT_HIP_CHECK(hipMemcpyAsync(dev, res.data(), count * 8, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    return BT_OK;
}

// recv[q * bytes ...] = rank q's send[0 .. bytes)
int comm_all_gather(bt_mgpu_comm *c, hipStream_t stream, const void *send, void *recv, size_t bytes)
{
    if (c->kind == 0) {
        BT_NCCL_CHECK(nccl().AllGather(send, recv, bytes, NCCL_UINT8, c->nccl, stream));
        return BT_OK;
    }
    LocalGroup *g = c->group;
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    g->ptr[c->rank] = send;
    BT_PEER_BARRIER(g);
    for (int q = 0; q < c->nranks; ++q)
        BT_HIP_CHECK(hipMemcpyAsync((char *) recv + (size_t) q * bytes, g->ptr[q], bytes,
                                    hipMemcpyDeviceToDevice, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    BT_PEER_BARRIER(g);
    return BT_OK;
}

// all-to-all-v in bytes; offsets and counts are host arrays [nranks].  A rank's segment for
// itself is a device copy (skipped when self_done: the caller wrote it in place).  `biggest`:
// the largest peer-to-peer message of ANY rank (every rank runs the same number of rounds).
int comm_all_to_all_v(bt_mgpu_comm *c, hipStream_t stream, const char *send, const int64_t *s_off,
                      const int64_t *s_cnt, char *recv, const int64_t *r_off, const int64_t *r_cnt,
                      int64_t biggest, bool self_done, int32_t *rounds_out)
{
    const int me = c->rank, n = c->nranks;
    if (rounds_out) *rounds_out = 1;
    const bool loop_self = c->kind == 0 && c->self_loopback && !self_done;
    if (!self_done && !loop_self && s_cnt[me] > 0)
        BT_HIP_CHECK(hipMemcpyAsync(recv + r_off[me], send + s_off[me], (size_t) s_cnt[me],
                                    hipMemcpyDeviceToDevice, stream));
    if (c->kind == 1) {
        LocalGroup *g = c->group;
        BT_HIP_CHECK(hipStreamSynchronize(stream));
        g->ptr[me] = send; g->off[me] = s_off; g->cnt[me] = s_cnt;
        BT_PEER_BARRIER(g);
        for (int q = 0; q < n; ++q) {
            if (q == me) continue;
            const int64_t nb = g->cnt[q][me];
            if (nb != r_cnt[q]) {
                set_error("all-to-all-v: rank %d sends %lld bytes to rank %d, which expects %lld",
                          q, (long long) nb, me, (long long) r_cnt[q]);
                g->fail();
                return BT_ERR_INTERNAL;
            }
            if (nb > 0)
                BT_HIP_CHECK(hipMemcpyAsync(recv + r_off[q], (const char *) g->ptr[q] + g->off[q][me],
                                            (size_t) nb, hipMemcpyDeviceToDevice, stream));
        }
        BT_HIP_CHECK(hipStreamSynchronize(stream));
        BT_PEER_BARRIER(g);
        return BT_OK;
    }
    if (biggest <= 0) return BT_OK;
    Nccl &nc = nccl();
    const int64_t rounds = std::max<int64_t>(1, div_up(biggest, MESSAGE_LIMIT_BYTES));
    if (rounds_out) *rounds_out = (int32_t) rounds;
    auto cut = [&](int64_t cnt, int64_t j) { return (j * cnt) / rounds; };
    for (int64_t j = 0; j < rounds; ++j) {
        BT_NCCL_CHECK(nc.GroupStart());
        // (a failure inside the group still closes it: RCCL keeps an open group per thread)
        auto in_group = [&]() -> int {
            for (int peer = 0; peer < n; ++peer) {
                if (peer == me && !loop_self) continue;
                const int64_t s0 = cut(s_cnt[peer], j), s1 = cut(s_cnt[peer], j + 1);
                const int64_t r0 = cut(r_cnt[peer], j), r1 = cut(r_cnt[peer], j + 1);
                if (s1 > s0)
                    BT_NCCL_CHECK(nc.Send(send + s_off[peer] + s0, (size_t) (s1 - s0), NCCL_UINT8, peer,
                                          c->nccl, stream));
                if (r1 > r0)
                    BT_NCCL_CHECK(nc.Recv(recv + r_off[peer] + r0, (size_t) (r1 - r0), NCCL_UINT8, peer,
                                          c->nccl, stream));
            }
            return BT_OK;
        };
        const int gs = in_group();
        if (gs != BT_OK) { (void) nc.GroupEnd(); return gs; }
        BT_NCCL_CHECK(nc.GroupEnd());
    }
    return BT_OK;
}

__global__ __launch_bounds__(256) void widen_hist_kernel(int64_t n, const int32_t *in, int64_t *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// The top of the GLOBAL tree (levels 0..k), a pure function of the all-reduced level-k cell
// histogram (kind "adaptive", point particles, unit weights: a box splits iff it holds more
// than max_particles_in_box particles, tree_build_kernels.py:577-591; empty boxes pruned).
// Paths are Morton paths, x most significant in every digit -- the order of the box numbers
// within a level.
struct TopPlan {
    bool valid = false;
    int D = 0, k = 0, nranks = 0;
    int64_t mpb = 0;
    std::vector<std::vector<int64_t>> counts;     // [k+1][C^lev]
    std::vector<std::vector<char>> exists, split;
    std::vector<std::vector<int32_t>> index;      // number of a box among the existing boxes of its level
    std::vector<int32_t> nboxes;                  // [k+1]
    bool sep_targets = false;                     // the build has separate targets
    std::vector<std::vector<int64_t>> src_counts; // [k+1][C^lev]: the sources among `counts`
    std::vector<int32_t> owner;                   // [C^k]
    std::vector<int64_t> prefix;                  // [C^k + 1]
    double bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0}, root_extent = 0;
};

void compute_plan(int D, int k, int64_t mpb, int nranks, const int64_t *hist, TopPlan &pl)
{
    const int C = 1 << D;
    const int64_t ncells = (int64_t) 1 << (D * k);
    pl.D = D; pl.k = k; pl.mpb = mpb; pl.nranks = nranks;
    pl.counts.assign((size_t) k + 1, {});
    pl.counts[k].assign(hist, hist + ncells);
    for (int lev = k - 1; lev >= 0; --lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        pl.counts[lev].assign((size_t) n, 0);
        for (int64_t i = 0; i < n; ++i)
            for (int m = 0; m < C; ++m) pl.counts[lev][i] += pl.counts[lev + 1][i * C + m];
    }
    std::vector<int64_t> unit_start((size_t) ncells);
    pl.valid = mpb > 0;
    if (mpb > 0) {
        pl.exists.assign((size_t) k + 1, {});
        pl.split.assign((size_t) k + 1, {});
        pl.index.assign((size_t) k + 1, {});
        pl.nboxes.assign((size_t) k + 1, 0);
        pl.exists[0].assign(1, 1);
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            pl.split[lev].assign((size_t) n, 0);
            pl.index[lev].assign((size_t) n, 0);
            int32_t run = 0;
            for (int64_t i = 0; i < n; ++i) {
                pl.split[lev][i] = pl.exists[lev][i] && pl.counts[lev][i] > mpb;
                run += pl.exists[lev][i] ? 1 : 0;
                pl.index[lev][i] = run - 1;
            }
            pl.nboxes[lev] = run;
            if (lev < k) {
                pl.exists[lev + 1].assign((size_t) n * C, 0);
                for (int64_t i = 0; i < n * C; ++i)
                    pl.exists[lev + 1][i] = pl.split[lev][i / C] && pl.counts[lev + 1][i] > 0;
            }
        }
        // frontier: the first cell of the top-tree leaf a cell lies in
        for (int64_t c = 0; c < ncells; ++c) {
            int leaf_level = k;
            for (int lev = k - 1; lev >= 0; --lev)
                if (!pl.split[lev][c >> (D * (k - lev))]) leaf_level = lev;
            const int sh = D * (k - leaf_level);
            unit_start[c] = (c >> sh) << sh;
        }
    } else {
        for (int64_t c = 0; c < ncells; ++c) unit_start[c] = c;
    }
    // contiguous Morton ranges balanced by particle count: a cell goes to the rank whose
    // ideal range contains its first particle
    const int64_t total = pl.counts[0][0];
    pl.prefix.assign((size_t) ncells + 1, 0);
    for (int64_t c = 0; c < ncells; ++c) pl.prefix[c + 1] = pl.prefix[c] + hist[c];
    pl.owner.assign((size_t) ncells, 0);
    for (int64_t c = 0; c < ncells; ++c) {
        const int64_t u = unit_start[c];
        const int64_t o = (int64_t) (((__int128) pl.prefix[u] * nranks) / std::max<int64_t>(total, 1));
        pl.owner[c] = (int32_t) std::min<int64_t>(o, nranks - 1);
    }
}

}  // namespace

struct MgpuState {
    Buf<unsigned char> points;       // received particles, interleaved [n_owned][dims]
    Buf<unsigned char> tpoints;      // ... separate targets
    Buf<int64_t> cell_prefix;        // [C^top_level + 1]
    TopPlan plan;                    // of the last exchange on this context
    hipEvent_t ev[2] = {nullptr, nullptr};   // around the payload all-to-all-v
    ~MgpuState() { for (auto &e : ev) if (e) (void) hipEventDestroy(e); }
    // local essential tree between bt_mgpu_let_build and bt_mgpu_let_export
    Buf<uint64_t> let_paths;         // [B] level-major, Morton order within a level
    Buf<int32_t> let_meta, let_gid;  // level | flags << 8; global box number
    Buf<int8_t> let_mask;            // 1: lists are built for this box on this rank
    int let_nlevels = 0, let_dims = 0, let_kind = 0;
    std::vector<int32_t> let_level_starts;
};

void bt_free_mgpu_state(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    delete ctx->mgpu;
    ctx->mgpu = nullptr;
}

namespace {

MgpuState *mgpu_state(bt_context *ctx)
{
    if (!ctx->mgpu) ctx->mgpu = new MgpuState();
    return ctx->mgpu;
}

constexpr int TOPMAX = 16;            // top levels a numbering kernel looks up by path

struct NumberArgs {
    int k, nlevels;
    int32_t shift[BT_MAX_LEVELS + 1];  // deep levels: global = local + shift[level]
    int32_t gstart[TOPMAX];            // shared top levels: global start of the level
    int32_t toff[TOPMAX];              // ... and where its index table starts
    const int32_t *index;              // concatenated index tables of the top levels
    double bmin[3], root_extent;
};

// Morton path of a top box from its centre (box_paths_kernel of bt_shard.hip), then the
// box's number in the global tree: position among the existing boxes of its level.
template <class T, int D>
__global__ __launch_bounds__(256) void number_boxes_kernel(int64_t nboxes, int64_t aligned,
        const T *centers, const uint8_t *levels, NumberArgs a, int32_t *box_ids)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nboxes) return;
    const int level = levels[b];
    if (level > a.k) { box_ids[b] = (int32_t) b + a.shift[level]; return; }
    uint64_t path = 0;
    const double scale = (double) (1ull << level);
#pragma unroll
    for (int ax = 0; ax < D; ++ax) {
        const double t = ((double) centers[(int64_t) ax * aligned + b] - a.bmin[ax]) / a.root_extent * scale;
        int64_t v = (int64_t) floor(t);
        v = v < 0 ? 0 : (v >= (int64_t) scale ? (int64_t) scale - 1 : v);
        for (int bit = 0; bit < level; ++bit)
            path |= (uint64_t) ((v >> bit) & 1) << (D * bit + (D - 1 - ax));
    }
    box_ids[b] = a.gstart[level] + a.index[a.toff[level] + (int64_t) path];
}

// ---- local essential tree --------------------------------------------------------------------

// does peer q need box b (a box below the ownership level in one of the cells q's lists reach)?
struct NeedPred {
    const uint64_t *paths;
    const uint8_t *levels;
    const uint64_t *need_bits;       // [ncells][nwords]: bit q of cell c
    int64_t b0;                      // first deep box
    int k, D, nwords, q;
    __device__ int32_t operator()(int64_t i) const
    {
        const int64_t b = b0 + i;
        const int lev = levels[b];
        if (!need_bits) return 1;
        const uint64_t cell = paths[b] >> (D * (lev - k));
        return (int32_t) ((need_bits[cell * nwords + (q >> 6)] >> (q & 63)) & 1ull);
    }
};

// one 16-byte record per box: (path, level | flags << 8 | global id << 32)
__global__ __launch_bounds__(256) void let_pack_kernel(int64_t n, NeedPred pr, const int32_t *pos,
        const uint8_t *flags, const int32_t *gids, uint64_t *rec /* at the peer's offset */)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !pr(i)) return;
    const int64_t b = pr.b0 + i;
    const uint64_t meta = (uint64_t) pr.levels[b] | ((uint64_t) flags[b] << 8);
    rec[2 * (int64_t) pos[i]] = pr.paths[b];
    rec[2 * (int64_t) pos[i] + 1] = meta | ((uint64_t) (uint32_t) gids[b] << 32);
}

__global__ __launch_bounds__(256) void iota_kernel(int64_t n, int32_t *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t) i;
}

__global__ __launch_bounds__(256) void count_diff_kernel(int64_t n, const uint64_t *a, const uint64_t *b,
                                                         int64_t *bad)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const bool d = i < n && a[i] != b[i];
    const uint64_t m = __ballot(d);
    if (m && (threadIdx.x & 63) == (unsigned) __ffsll((long long) m) - 1)
        atomicAdd((unsigned long long *) bad, (unsigned long long) __popcll(m));
}

// deep boxes of the LET before the sort: this rank's own (contiguous in the local tree), then
// the halo records; key = level << pathbits | path
__global__ __launch_bounds__(256) void let_deep_kernel(int64_t n_mine, int64_t n_halo, int64_t b0,
        const uint64_t *paths, const uint8_t *levels, const uint8_t *flags, const int32_t *gids,
        const uint64_t *halo_rec, int pathbits, uint64_t *key, uint64_t *d_path, int32_t *d_meta,
        int32_t *d_gid)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mine + n_halo) return;
    uint64_t path;
    int32_t meta, gid;
    if (i < n_mine) {
        const int64_t b = b0 + i;
        path = paths[b];
        meta = (int32_t) levels[b] | ((int32_t) flags[b] << 8);
        gid = gids[b];
    } else {
        const int64_t j = i - n_mine;
        path = halo_rec[2 * j];
        const uint64_t w = halo_rec[2 * j + 1];
        meta = (int32_t) (w & 0xffffffffu);
        gid = (int32_t) (w >> 32);
    }
    d_path[i] = path; d_meta[i] = meta; d_gid[i] = gid;
    key[i] = ((uint64_t) (meta & 0xff) << pathbits) | path;
}

// sorted deep boxes into the LET arrays behind the top boxes; per level: count, and the first
// and last position of this rank's own boxes (they are one contiguous run: its cells are one
// Morton range)
// (boundaries of the sorted order, one writer each: atomics on a dozen addresses from
// millions of threads serialise in L2 -- 60 ms at 5*10^6 boxes)
struct LetLevelInfo {
    int32_t level_first[BT_MAX_LEVELS + 1];   // first sorted position of a level, or -1
    int32_t mine_runs[BT_MAX_LEVELS + 1];     // runs of this rank's boxes in the level (must be <= 1)
    int32_t mine_first[BT_MAX_LEVELS + 1];
    int32_t mine_last[BT_MAX_LEVELS + 1];
};

__global__ __launch_bounds__(256) void let_place_kernel(int64_t nd, int64_t n_mine, int64_t ntop,
        const uint32_t *order, const uint64_t *d_path, const int32_t *d_meta, const int32_t *d_gid,
        uint64_t *all_paths, int32_t *all_meta, int32_t *all_gid, int8_t *mask, LetLevelInfo *info)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= nd) return;
    const uint32_t o = order[i];
    const int32_t meta = d_meta[o];
    const bool mine = (int64_t) o < n_mine;
    all_paths[ntop + i] = d_path[o];
    all_meta[ntop + i] = meta;
    all_gid[ntop + i] = d_gid[o];
    mask[ntop + i] = mine ? 1 : 0;
    const int lev = meta & 0xff;
    int prev_lev = -1, next_lev = -1;
    bool prev_mine = false, next_mine = false;
    if (i > 0) { const uint32_t po = order[i - 1]; prev_lev = d_meta[po] & 0xff; prev_mine = (int64_t) po < n_mine; }
    if (i + 1 < nd) { const uint32_t no = order[i + 1]; next_lev = d_meta[no] & 0xff; next_mine = (int64_t) no < n_mine; }
    if (prev_lev != lev) info->level_first[lev] = (int32_t) i;
    if (mine && !(prev_mine && prev_lev == lev)) {
        info->mine_first[lev] = (int32_t) i;
        atomicAdd(&info->mine_runs[lev], 1);
    }
    if (mine && !(next_mine && next_lev == lev)) info->mine_last[lev] = (int32_t) i;
}

__global__ __launch_bounds__(256) void let_split_meta_kernel(int64_t n, const int32_t *meta,
                                                             uint8_t *levels, uint8_t *flags)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    levels[i] = (uint8_t) (meta[i] & 0xff);
    flags[i] = (uint8_t) ((meta[i] >> 8) & 0xff);
}

// need[q] = my non-empty cells within `ring` cells (Chebyshev) of a cell owned by rank q --
// the subtrees q's lists can reach -- as bit q of a word per cell.  The set of q's cells is
// dilated by `ring` along every axis of the cell grid and intersected with my cells.
void cells_needed_by(const TopPlan &pl, int rank, int ring, int nwords, std::vector<uint64_t> &bits,
                     std::vector<char> &any_for_peer)
{
    const int D = pl.D, k = pl.k, n = 1 << k;
    const int64_t ncells = (int64_t) 1 << (D * k);
    bits.assign((size_t) ncells * nwords, 0);
    any_for_peer.assign((size_t) pl.nranks, 0);
    if (pl.nranks < 2) return;
    // grid position <-> Morton index (x most significant in every digit), as tables
    int64_t gridn = 1;
    for (int ax = 0; ax < D; ++ax) gridn *= n;
    std::vector<int32_t> cell_of_grid((size_t) gridn);
    std::vector<int16_t> xyz_of_cell((size_t) ncells * 3, 0);
    for (int64_t c = 0; c < ncells; ++c) {
        int64_t gi = 0;
        for (int ax = 0; ax < D; ++ax) {
            int v = 0;
            for (int bit = 0; bit < k; ++bit) v |= (int) ((c >> (D * bit + (D - 1 - ax))) & 1) << bit;
            xyz_of_cell[(size_t) c * 3 + ax] = (int16_t) v;
            gi = gi * n + v;
        }
        cell_of_grid[(size_t) gi] = (int32_t) c;
    }
    // for each of my non-empty cells: the owners of the cells within `ring`
    for (int64_t c = 0; c < ncells; ++c) {
        if (pl.owner[c] != rank || pl.counts[k][c] <= 0) continue;
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        for (int ax = 0; ax < D; ++ax) {
            const int v = xyz_of_cell[(size_t) c * 3 + ax];
            lo[ax] = std::max(0, v - ring); hi[ax] = std::min(n - 1, v + ring);
        }
        int p[3];
        for (p[0] = lo[0]; p[0] <= hi[0]; ++p[0])
            for (p[1] = lo[1]; p[1] <= hi[1]; ++p[1])
                for (p[2] = lo[2]; p[2] <= hi[2]; ++p[2]) {
                    int64_t gi = 0;
                    for (int ax = 0; ax < D; ++ax) gi = gi * n + p[ax];
                    const int q = pl.owner[cell_of_grid[(size_t) gi]];
                    if (q == rank) continue;
                    bits[(size_t) c * nwords + (q >> 6)] |= 1ull << (q & 63);
                    any_for_peer[q] = 1;
                }
    }
}

}  // namespace

extern "C" {

// ---- communicators ---------------------------------------------------------------------------

int bt_mgpu_comm_rccl(void *nccl_comm, int rank, int nranks, bt_mgpu_comm **out)
{
    if (!nccl_comm || !out || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("bt_mgpu_comm_rccl: invalid argument");
        return BT_ERR_INVALID;
    }
    if (!nccl().ok) {
        set_error("bt_mgpu_comm_rccl: librccl.so could not be loaded");
        return BT_ERR_UNSUPPORTED;
    }
    bt_mgpu_comm *c = new bt_mgpu_comm();
    c->kind = 0; c->rank = rank; c->nranks = nranks; c->nccl = (nccl_comm_t) nccl_comm;
    const char *lb = getenv("BT_MGPU_SELF_LOOPBACK");
    c->self_loopback = lb && atoi(lb) != 0;
    *out = c;
    return BT_OK;
}

int bt_mgpu_local_group_create(int nranks, void **group)
{
    if (nranks < 1 || !group) { set_error("bt_mgpu_local_group_create: invalid argument"); return BT_ERR_INVALID; }
    LocalGroup *g = new LocalGroup();
    g->n = nranks;
    g->ptr.assign((size_t) nranks, nullptr);
    g->off.assign((size_t) nranks, nullptr);
    g->cnt.assign((size_t) nranks, nullptr);
    *group = g;
    return BT_OK;
}

void bt_mgpu_local_group_destroy(void *group) { delete (LocalGroup *) group; }

int bt_mgpu_comm_local(void *group, int rank, bt_mgpu_comm **out)
{
    LocalGroup *g = (LocalGroup *) group;
    if (!g || !out || rank < 0 || rank >= g->n) { set_error("bt_mgpu_comm_local: invalid argument"); return BT_ERR_INVALID; }
    bt_mgpu_comm *c = new bt_mgpu_comm();
    c->kind = 1; c->rank = rank; c->nranks = g->n; c->group = g;
    *out = c;
    return BT_OK;
}

void bt_mgpu_comm_destroy(bt_mgpu_comm *c) { delete c; }

int bt_mgpu_use_rccl_library(const char *path)
{
    if (!path || !*path) { set_error("bt_mgpu_use_rccl_library: empty path"); return BT_ERR_INVALID; }
    nccl_library_path() = path;
    return BT_OK;
}

int bt_mgpu_comm_set_self_loopback(bt_mgpu_comm *c, int on)
{
    if (!c) { set_error("bt_mgpu_comm_set_self_loopback: invalid argument"); return BT_ERR_INVALID; }
    c->self_loopback = on != 0;
    return BT_OK;
}

// Host part, a pure function of the all-reduced histogram (identical on every rank):
// owner rank of every level-`top_level` cell, and the exclusive prefix sums of the
// histogram.  All cells below a leaf of the global top tree go to one rank, so no global
// leaf straddles ranks.  max_particles_in_box <= 0: cells are assigned individually (no
// top-tree plan).
int bt_mgpu_plan(int dims, int top_level, int64_t max_particles_in_box, int nranks,
                 const int64_t *global_hist, int32_t *owner_of_cell, int64_t *cell_prefix)
{
    if (dims < 1 || dims > 3 || top_level < 1 || dims * top_level > 30 || nranks < 1
            || !global_hist || !owner_of_cell) {
        set_error("bt_mgpu_plan: invalid argument");
        return BT_ERR_INVALID;
    }
    TopPlan pl;
    compute_plan(dims, top_level, max_particles_in_box, nranks, global_hist, pl);
    std::copy(pl.owner.begin(), pl.owner.end(), owner_of_cell);
    if (cell_prefix) std::copy(pl.prefix.begin(), pl.prefix.end(), cell_prefix);
    return BT_OK;
}

static int bt_mgpu_exchange_body(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_params *p, bt_mgpu_shard *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !comm || !p || !out) {
        set_error("bt_mgpu_exchange: invalid argument");
        return BT_ERR_INVALID;
    }
    if (p->dims < 1 || p->dims > 3 || (p->coord_kind != BT_F32 && p->coord_kind != BT_F64) || p->n < 0
            || p->ntargets < 0) {
        set_error("bt_mgpu_exchange: bad dims / coord_kind / n");
        return BT_ERR_INVALID;
    }
    const int rank = comm->rank, nranks = comm->nranks;
    if (nranks > BT_MGPU_MAX_RANKS) {
        set_error("bt_mgpu_exchange: %d ranks; the one-sweep partition supports at most %d owners",
                  nranks, BT_MGPU_MAX_RANKS);
        return BT_ERR_UNSUPPORTED;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    const int D = p->dims;
    const bool f64 = p->coord_kind == BT_F64;
    const int es = f64 ? 8 : 4;
    // particle sets: sources, and separate targets if ANY rank has some (a property of the
    // job, agreed below with the bounding box: it sizes the collectives and decides the flags
    // of the shared top boxes, so it cannot be a rank's own view of its chunk)
    const int64_t nset[2] = {p->n, p->ntargets};
    const void *const *cset[2] = {p->coords, p->targets};
    hipStream_t stream = ctx->stream;

    // ---- 1. global bounding box -> root box --------------------------------------------
    double h_mm[7];
    for (int ax = 0; ax < D; ++ax) { h_mm[ax] = 1.7976931348623158e+308; h_mm[D + ax] = 1.7976931348623158e+308; }
    for (int s = 0; s < 2; ++s) {
        if (nset[s] == 0) continue;
        double lmin[3], lmax[3];
        BT_CHECK(bt_bbox(ctx, D, p->coord_kind, cset[s], nullptr, nset[s], lmin, lmax));
        for (int ax = 0; ax < D; ++ax) { h_mm[ax] = std::min(h_mm[ax], lmin[ax]); h_mm[D + ax] = std::min(h_mm[D + ax], -lmax[ax]); }
    }
    h_mm[2 * D] = p->ntargets > 0 ? -1.0 : 0.0;      // MIN: -1 if some rank has targets
    Buf<double> mm;
    BT_CHECK(mm.alloc(ctx->pool, 2 * D + 1));
    BT_HIP_CHECK(hipMemcpyAsync(mm.get(), h_mm, sizeof(double) * (2 * D + 1), hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_reduce(comm, stream, mm.get(), 2 * D + 1, RED_MIN_F64));
    BT_HIP_CHECK(hipMemcpyAsync(h_mm, mm.get(), sizeof(double) * (2 * D + 1), hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    const bool sep = h_mm[2 * D] < 0;
    const int nsets = sep ? 2 : 1;
    if (h_mm[0] > -h_mm[D]) {
        set_error("bt_mgpu_exchange: no rank has any particle");
        return BT_ERR_INVALID;
    }
    double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}, root_extent = 0;
    if (f64) {
        // tree_build.py:462-476 in the coordinate type
        double ext = 0;
        for (int ax = 0; ax < D; ++ax) ext = std::max(ext, (-h_mm[D + ax]) - h_mm[ax]);
        root_extent = ext * (1 + 1e-4);
        for (int ax = 0; ax < D; ++ax) { bmin[ax] = h_mm[ax]; bmax[ax] = bmin[ax] + root_extent; }
    } else {
        float ext = 0;
        for (int ax = 0; ax < D; ++ax) ext = std::max(ext, (float) (-h_mm[D + ax]) - (float) h_mm[ax]);
        const float re = ext * (float) (1 + 1e-4);
        root_extent = re;
        for (int ax = 0; ax < D; ++ax) { bmin[ax] = (float) h_mm[ax]; bmax[ax] = (float) ((float) h_mm[ax] + re); }
    }

    // ---- 2. cell histograms (sources, targets), all-reduced ------------------------------------
    const int k = p->top_level > 0 ? p->top_level : (D == 3 ? 5 : D == 2 ? 7 : 12);
    const int64_t ncells = (int64_t) 1 << (D * k);
    Buf<uint32_t> cells[2];
    Buf<int32_t> hist32, owner_d;
    Buf<int64_t> hist64;
    BT_CHECK(hist32.alloc(ctx->pool, 2 * ncells));
    BT_CHECK(hist64.alloc(ctx->pool, 2 * ncells));
    BT_CHECK(owner_d.alloc(ctx->pool, ncells));
    BT_HIP_CHECK(hipMemsetAsync(hist32.get(), 0, (size_t) ncells * 8, stream));
    for (int s = 0; s < nsets; ++s) {
        if (nset[s] == 0) continue;
        BT_CHECK(cells[s].alloc(ctx->pool, nset[s]));
        BT_CHECK(bt_morton_cells(ctx, D, p->coord_kind, cset[s], nset[s], bmin, bmax, k, cells[s].get(),
                                 hist32.get() + s * ncells));
    }
    widen_hist_kernel<<<(unsigned) div_up(2 * ncells, 256), 256, 0, stream>>>(2 * ncells, hist32.get(), hist64.get());
    BT_HIP_CHECK(hipGetLastError());
    std::vector<int32_t> h_local((size_t) 2 * ncells);
    BT_HIP_CHECK(hipMemcpyAsync(h_local.data(), hist32.get(), (size_t) ncells * 8, hipMemcpyDeviceToHost, stream));
    BT_CHECK(comm_all_reduce(comm, stream, hist64.get(), (size_t) (nsets * ncells), RED_SUM_I64));
    std::vector<int64_t> ghist2((size_t) 2 * ncells, 0), ghist((size_t) ncells);
    BT_HIP_CHECK(hipMemcpyAsync(ghist2.data(), hist64.get(), (size_t) (nsets * ncells) * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    for (int64_t c = 0; c < ncells; ++c) ghist[c] = ghist2[c] + ghist2[ncells + c];

    // ---- 3. plan (host, identical on all ranks), counts --------------------------------------
    MgpuState *ms = mgpu_state(ctx);
    TopPlan &pl = ms->plan;
    compute_plan(D, k, p->max_particles_in_box, nranks, ghist.data(), pl);
    // which top boxes hold sources / targets (flags of the shared top levels)
    pl.sep_targets = sep;
    pl.src_counts.assign((size_t) k + 1, {});
    pl.src_counts[k].assign(ghist2.begin(), ghist2.begin() + ncells);
    for (int lev = k - 1; lev >= 0; --lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        pl.src_counts[lev].assign((size_t) n, 0);
        for (int64_t i = 0; i < n; ++i)
            for (int m = 0; m < (1 << D); ++m) pl.src_counts[lev][i] += pl.src_counts[lev + 1][i * (1 << D) + m];
    }
    for (int ax = 0; ax < 3; ++ax) { pl.bbox_min[ax] = bmin[ax]; pl.bbox_max[ax] = bmax[ax]; }
    pl.root_extent = root_extent;
    BT_HIP_CHECK(hipMemcpyAsync(owner_d.get(), pl.owner.data(), (size_t) ncells * 4, hipMemcpyHostToDevice, stream));
    std::vector<int64_t> send_counts((size_t) 2 * nranks, 0);       // [set][owner]
    for (int s = 0; s < nsets; ++s)
        for (int64_t c = 0; c < ncells; ++c) send_counts[(size_t) s * nranks + pl.owner[c]] += h_local[(size_t) s * ncells + c];
    Buf<int64_t> counts_d;
    const int64_t row = 2 * nranks;
    BT_CHECK(counts_d.alloc(ctx->pool, row * (nranks + 1)));
    BT_HIP_CHECK(hipMemcpyAsync(counts_d.get(), send_counts.data(), (size_t) row * 8, hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_gather(comm, stream, counts_d.get(), counts_d.get() + row, (size_t) row * 8));
    std::vector<int64_t> matrix((size_t) row * nranks);      // [sender][set][receiver]
    BT_HIP_CHECK(hipMemcpyAsync(matrix.data(), counts_d.get() + row, matrix.size() * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    const int64_t rec = (int64_t) D * es;                       // bytes per particle

    // ---- 4. payload: interleaved coordinates, one exchange per particle set -------------------
    if (!ms->ev[0]) { BT_HIP_CHECK(hipEventCreate(&ms->ev[0])); BT_HIP_CHECK(hipEventCreate(&ms->ev[1])); }
    BT_HIP_CHECK(hipEventRecord(ms->ev[0], stream));
    int32_t rounds_total = 0;
    int64_t bytes_sent = 0;
    unsigned char *points_of[2] = {nullptr, nullptr};
    int64_t nrecv_of[2] = {0, 0};
    for (int s = 0; s < nsets; ++s) {
        const int64_t n = nset[s];
        std::vector<int64_t> s_off((size_t) nranks + 1, 0), r_off((size_t) nranks + 1, 0);
        std::vector<int64_t> s_cnt_b((size_t) nranks), r_cnt_b((size_t) nranks), s_off_b((size_t) nranks),
            r_off_b((size_t) nranks);
        int64_t biggest = 0;
        for (int r = 0; r < nranks; ++r) {
            const int64_t sc = send_counts[(size_t) s * nranks + r];
            const int64_t rc = matrix[(size_t) r * row + (size_t) s * nranks + rank];
            s_off[r + 1] = s_off[r] + sc;
            r_off[r + 1] = r_off[r] + rc;
            s_cnt_b[r] = sc * rec; r_cnt_b[r] = rc * rec;
            s_off_b[r] = s_off[r] * rec; r_off_b[r] = r_off[r] * rec;
            for (int q = 0; q < nranks; ++q)
                if (q != r) biggest = std::max(biggest, matrix[(size_t) r * row + (size_t) s * nranks + q] * rec);
        }
        const int64_t nrecv = r_off[nranks];
        Buf<unsigned char> send;
        BT_CHECK(send.alloc(ctx->pool, n * D * es));
        unsigned char *points = nullptr;
        const int64_t points_bytes = std::max<int64_t>(nrecv, 1) * D * es;
        Buf<unsigned char> &own = s == 0 ? ms->points : ms->tpoints;
        if (p->alloc) {
            points = (unsigned char *) p->alloc(p->alloc_user, points_bytes);
            if (!points) { set_error("bt_mgpu_exchange: the caller's allocator returned NULL"); return BT_ERR_ALLOC; }
            own.reset();
        } else {
            BT_CHECK(own.alloc(ctx->pool, points_bytes));
            points = own.get();
        }
        // one sweep over the coordinates: stable partition by owner into the send buffer, the
        // segment this rank keeps straight into the receive buffer (bt_shard.hip)
        const bool loop_self = comm->kind == 0 && comm->self_loopback;
        int32_t rounds = 1;
        if (!loop_self) {
            BT_CHECK(bt_partition_pack(ctx, D, es, cset[s], cells[s].get(), n, owner_d.get(), nranks, rank,
                                       s_off[rank], r_off[rank], send.get(), points));
            s_cnt_b[rank] = 0; r_cnt_b[rank] = 0;       // (own segment: packed in place)
        } else {
            // test switch (bt_mgpu_comm_set_self_loopback): the own segment is packed into the
            // send buffer like any other and makes the trip through ncclSend / ncclRecv
            BT_CHECK(bt_partition_pack(ctx, D, es, cset[s], cells[s].get(), n, owner_d.get(), nranks, rank,
                                       s_off[rank], s_off[rank], send.get(), send.get()));
            biggest = std::max(biggest, s_cnt_b[rank]);
        }
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send.get(), s_off_b.data(), s_cnt_b.data(),
                                   (char *) points, r_off_b.data(), r_cnt_b.data(), biggest, !loop_self, &rounds));
        BT_HIP_CHECK(hipStreamSynchronize(stream));     // the send buffer and the host vectors go out of scope
        rounds_total += rounds;
        bytes_sent += (n - (loop_self ? 0 : send_counts[(size_t) s * nranks + rank])) * rec;
        points_of[s] = points;
        nrecv_of[s] = nrecv;
    }
    BT_HIP_CHECK(hipEventRecord(ms->ev[1], stream));
    BT_CHECK(ms->cell_prefix.alloc(ctx->pool, ncells + 1));
    BT_HIP_CHECK(hipMemcpyAsync(ms->cell_prefix.get(), pl.prefix.data(), (size_t) (ncells + 1) * 8,
                                hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));     // host vectors above go out of scope

    out->n_owned = nrecv_of[0];
    out->points = points_of[0];
    out->n_owned_targets = nrecv_of[1];
    out->target_points = points_of[1];
    for (int ax = 0; ax < D; ++ax) { out->bbox_min[ax] = bmin[ax]; out->bbox_max[ax] = bmax[ax]; }
    out->root_extent = root_extent;
    out->top_level = k;
    out->top_cell_prefix = p->max_particles_in_box > 0 ? ms->cell_prefix.get() : nullptr;
    out->bytes_sent = bytes_sent;
    out->rounds = rounds_total;
    out->sep_targets = sep ? 1 : 0;
    (void) hipEventElapsedTime(&out->a2a_ms, ms->ev[0], ms->ev[1]);
    return BT_OK;
}

// ---- step 5: global numbering ------------------------------------------------------------------

static int bt_mgpu_number_body(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                   int32_t *box_ids, bt_mgpu_numbering *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !comm || !tree || !out || !tree->level_start_box_nrs || tree->nlevels < 1
            || tree->nlevels > BT_MAX_LEVELS || (tree->nboxes > 0 && (!box_ids || !tree->box_centers || !tree->box_levels))) {
        set_error("bt_mgpu_number: invalid argument");
        return BT_ERR_INVALID;
    }
    MgpuState *ms = ctx->mgpu;
    if (!ms || !ms->plan.valid) {
        set_error("bt_mgpu_number: no top-tree plan on this context (bt_mgpu_exchange with "
                  "max_particles_in_box > 0 comes first)");
        return BT_ERR_INVALID;
    }
    const TopPlan &pl = ms->plan;
    if (pl.D != tree->dims || pl.k >= TOPMAX) { set_error("bt_mgpu_number: plan / tree mismatch"); return BT_ERR_INVALID; }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    hipStream_t stream = ctx->stream;
    const int rank = comm->rank, nranks = comm->nranks, k = pl.k, nlev = tree->nlevels;
    constexpr int W = BT_MAX_LEVELS + 2;
    std::vector<int64_t> mine((size_t) W, 0), all((size_t) W * nranks);
    for (int l = 0; l < nlev; ++l) mine[l] = tree->level_start_box_nrs[l + 1] - tree->level_start_box_nrs[l];
    mine[BT_MAX_LEVELS] = tree->nsources;
    mine[BT_MAX_LEVELS + 1] = tree->ntargets;
    Buf<int64_t> gath;
    BT_CHECK(gath.alloc(ctx->pool, (int64_t) W * (nranks + 1)));
    BT_HIP_CHECK(hipMemcpyAsync(gath.get(), mine.data(), (size_t) W * 8, hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_gather(comm, stream, gath.get(), gath.get() + W, (size_t) W * 8));
    BT_HIP_CHECK(hipMemcpyAsync(all.data(), gath.get() + W, all.size() * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    // the deepest local tree; boxes of levels <= k are shared and numbered by Morton path
    // from the plan, deeper levels are the concatenation of the ranks' level slices
    int gl = 0;
    for (int r = 0; r < nranks; ++r) {
        int c = 0;
        for (int l = 0; l < BT_MAX_LEVELS; ++l) c += all[(size_t) r * W + l] > 0 ? 1 : 0;
        gl = std::max(gl, c);
    }
    out->nlevels = gl;
    NumberArgs na{};
    na.k = k; na.nlevels = nlev;
    int64_t start = 0;
    for (int l = 0; l < gl; ++l) {
        int64_t nl = 0;
        out->level_start_box_nrs[l] = (int32_t) start;
        if (l <= k) {
            nl = pl.nboxes[l];
        } else {
            int64_t before = 0;
            for (int r = 0; r < nranks; ++r) {
                if (r < rank) before += all[(size_t) r * W + l];
                nl += all[(size_t) r * W + l];
            }
            out->deep_base[l] = (int32_t) (start + before);
            if (l < nlev) na.shift[l] = out->deep_base[l] - tree->level_start_box_nrs[l];
        }
        start += nl;
        if (start > 0x7fffffff) { set_error("bt_mgpu_number: more than 2^31-1 boxes in the global tree"); return BT_ERR_UNSUPPORTED; }
    }
    out->level_start_box_nrs[gl] = (int32_t) start;
    out->nboxes = start;
    for (int r = 0; r < nranks; ++r) {
        out->nsources += all[(size_t) r * W + BT_MAX_LEVELS];
        out->ntargets += all[(size_t) r * W + BT_MAX_LEVELS + 1];
        if (r < rank) {
            out->source_offset += all[(size_t) r * W + BT_MAX_LEVELS];
            out->target_offset += all[(size_t) r * W + BT_MAX_LEVELS + 1];
        }
    }
    if (tree->nboxes == 0) return BT_OK;
    // index tables of the shared top levels
    std::vector<int32_t> index;
    const int ntop_levels = std::min(k + 1, nlev);
    for (int l = 0; l < ntop_levels; ++l) {
        na.toff[l] = (int32_t) index.size();
        na.gstart[l] = out->level_start_box_nrs[l];
        index.insert(index.end(), pl.index[l].begin(), pl.index[l].end());
    }
    Buf<int32_t> index_d;
    BT_CHECK(index_d.alloc(ctx->pool, (int64_t) index.size()));
    BT_HIP_CHECK(hipMemcpyAsync(index_d.get(), index.data(), index.size() * 4, hipMemcpyHostToDevice, stream));
    na.index = index_d.get();
    for (int ax = 0; ax < 3; ++ax) na.bmin[ax] = pl.bbox_min[ax];
    na.root_extent = pl.root_extent;
    const unsigned blocks = (unsigned) div_up(tree->nboxes, 256);
#define NB(T, D) number_boxes_kernel<T, D><<<blocks, 256, 0, stream>>>(tree->nboxes, tree->aligned_nboxes, \
        (const T *) tree->box_centers, tree->box_levels, na, box_ids)
    if (tree->coord_kind == BT_F64) { if (pl.D == 1) NB(double, 1); else if (pl.D == 2) NB(double, 2); else NB(double, 3); }
    else { if (pl.D == 1) NB(float, 1); else if (pl.D == 2) NB(float, 2); else NB(float, 3); }
#undef NB
    BT_HIP_CHECK(hipGetLastError());
    BT_HIP_CHECK(hipStreamSynchronize(stream));     // `index` goes out of scope
    return BT_OK;
}

// ---- step 6: local essential tree ----------------------------------------------------------------

static int bt_mgpu_let_build_body(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                      const int32_t *box_ids, const bt_mgpu_numbering *num, int well_sep_is_n_away,
                      bt_mgpu_let_sizes *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !comm || !tree || !num || !out || !tree->level_start_box_nrs || well_sep_is_n_away < 1
            || (tree->nboxes > 0 && (!box_ids || !tree->box_centers || !tree->box_levels || !tree->box_flags))) {
        set_error("bt_mgpu_let_build: invalid argument");
        return BT_ERR_INVALID;
    }
    MgpuState *ms = ctx->mgpu;
    if (!ms || !ms->plan.valid) {
        set_error("bt_mgpu_let_build: no top-tree plan on this context (bt_mgpu_exchange with "
                  "max_particles_in_box > 0 comes first)");
        return BT_ERR_INVALID;
    }
    const TopPlan &pl = ms->plan;
    if (pl.D != tree->dims || num->nlevels < 1 || num->nlevels > BT_MAX_LEVELS || tree->nlevels < 1
            || tree->nlevels > num->nlevels) {
        set_error("bt_mgpu_let_build: plan / tree / numbering mismatch (dims %d vs %d, %d local and %d "
                  "global levels)", pl.D, tree->dims, tree->nlevels, num->nlevels);
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_CHECK(bt::zero_begin(ctx));
    memset(out, 0, sizeof(*out));
    hipStream_t stream = ctx->stream;
    const int rank = comm->rank, nranks = comm->nranks, D = pl.D, k = pl.k;
    const int nlev_local = tree->nlevels, nlev = num->nlevels;
    const int64_t nb = tree->nboxes;
    const int ntop_levels = std::min(k + 1, nlev);
    const int lmax = std::max(nlev - 1, 1);
    const int pathbits = D * lmax;
    int levbits = 1;
    while ((1 << levbits) <= lmax) ++levbits;
    if (pathbits + levbits > 64) {
        set_error("bt_mgpu_let_build: a tree of %d levels does not fit the 64-bit (level, path) key", nlev);
        return BT_ERR_UNSUPPORTED;
    }

    // -- Morton paths of my boxes ----------------------------------------------------------------
    Buf<uint64_t> paths;
    BT_CHECK(paths.alloc(ctx->pool, nb));
    if (nb > 0)
        BT_CHECK(bt_box_morton_paths(ctx, D, tree->coord_kind, nb, tree->aligned_nboxes, tree->box_centers,
                                     tree->box_levels, pl.bbox_min, pl.root_extent, paths.get()));
    // my deep boxes (levels > k) are the tail of the level-major local tree
    const int64_t b0 = nlev_local > k + 1 ? tree->level_start_box_nrs[k + 1] : nb;
    const int64_t n_mine = nb - b0;

    // -- halo: my deep boxes in the cells other ranks' lists can reach ------------------------------
    const int nwords = (nranks + 63) / 64;
    std::vector<uint64_t> need_bits;
    std::vector<char> any_for_peer;
    cells_needed_by(pl, rank, well_sep_is_n_away, nwords, need_bits, any_for_peer);
    Buf<uint64_t> need_d;
    BT_CHECK(need_d.alloc(ctx->pool, (int64_t) need_bits.size()));
    BT_HIP_CHECK(hipMemcpyAsync(need_d.get(), need_bits.data(), need_bits.size() * 8, hipMemcpyHostToDevice, stream));
    std::vector<int> peers;
    for (int q = 0; q < nranks; ++q)
        if (q != rank && any_for_peer[q] && n_mine > 0) peers.push_back(q);
    Buf<int32_t> pos;                 // [peers][n_mine + 1]
    BT_CHECK(pos.alloc(ctx->pool, (int64_t) peers.size() * (n_mine + 1)));
    std::vector<int32_t> h_tot(peers.size(), 0);
    NeedPred pr{paths.get(), tree->box_levels, need_d.get(), b0, k, D, nwords, 0};
    for (size_t i = 0; i < peers.size(); ++i) {
        pr.q = peers[i];
        int32_t *pp = pos.get() + (int64_t) i * (n_mine + 1);
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, pr, n_mine, pp, (int32_t *) nullptr, true)));
        BT_CHECK(bt::d2h(ctx, &h_tot[i], pp + n_mine, 4));
    }
    BT_CHECK(bt::sync_stream(ctx));
    std::vector<int64_t> s_cnt((size_t) nranks, 0), s_off((size_t) nranks, 0);
    for (size_t i = 0; i < peers.size(); ++i) s_cnt[peers[i]] = h_tot[i];
    int64_t nsend = 0;
    for (int q = 0; q < nranks; ++q) { s_off[q] = nsend; nsend += s_cnt[q]; }
    Buf<uint64_t> send_rec;
    BT_CHECK(send_rec.alloc(ctx->pool, 2 * std::max<int64_t>(nsend, 1)));
    for (size_t i = 0; i < peers.size(); ++i) {
        if (h_tot[i] == 0) continue;
        pr.q = peers[i];
        let_pack_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
            n_mine, pr, pos.get() + (int64_t) i * (n_mine + 1), tree->box_flags, box_ids,
            send_rec.get() + 2 * s_off[peers[i]]);
    }
    BT_HIP_CHECK(hipGetLastError());
    // counts: every rank learns the whole matrix (and with it the largest message)
    Buf<int64_t> cm;
    BT_CHECK(cm.alloc(ctx->pool, (int64_t) nranks * (nranks + 1)));
    BT_HIP_CHECK(hipMemcpyAsync(cm.get(), s_cnt.data(), (size_t) nranks * 8, hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_gather(comm, stream, cm.get(), cm.get() + nranks, (size_t) nranks * 8));
    std::vector<int64_t> matrix((size_t) nranks * nranks);
    BT_HIP_CHECK(hipMemcpyAsync(matrix.data(), cm.get() + nranks, matrix.size() * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int64_t> r_cnt((size_t) nranks), r_off((size_t) nranks);
    int64_t nrecv = 0, biggest = 0;
    for (int q = 0; q < nranks; ++q) {
        r_cnt[q] = matrix[(size_t) q * nranks + rank];
        r_off[q] = nrecv;
        nrecv += r_cnt[q];
        for (int t = 0; t < nranks; ++t)
            if (t != q) biggest = std::max(biggest, matrix[(size_t) q * nranks + t] * 16);
    }
    Buf<uint64_t> halo_rec;
    BT_CHECK(halo_rec.alloc(ctx->pool, 2 * std::max<int64_t>(nrecv, 1)));
    if (comm->kind == 0 && comm->self_loopback && n_mine > 0) {
        // test switch: no peer of a one-rank world gets halo records, so the records of ALL my
        // deep boxes make the trip to myself through ncclSend / ncclRecv and are compared with
        // what was sent; they are not boxes of the LET (they are mine already)
        Buf<uint64_t> echo_s, echo_r;
        Buf<int32_t> ident;
        Buf<int64_t> bad;
        BT_CHECK(echo_s.alloc(ctx->pool, 2 * n_mine)); BT_CHECK(echo_r.alloc(ctx->pool, 2 * n_mine));
        BT_CHECK(ident.alloc(ctx->pool, n_mine)); BT_CHECK(bad.alloc(ctx->pool, 1));
        iota_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(n_mine, ident.get());
        NeedPred all = pr; all.need_bits = nullptr;
        let_pack_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
            n_mine, all, ident.get(), tree->box_flags, box_ids, echo_s.get());
        BT_HIP_CHECK(hipMemsetAsync(echo_r.get(), 0xff, (size_t) n_mine * 16, stream));
        BT_HIP_CHECK(hipMemsetAsync(bad.get(), 0, 8, stream));
        std::vector<int64_t> z((size_t) nranks, 0), c1((size_t) nranks, 0);
        c1[rank] = n_mine * 16;
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) echo_s.get(), z.data(), c1.data(),
                                   (char *) echo_r.get(), z.data(), c1.data(), n_mine * 16, false, nullptr));
        count_diff_kernel<<<(unsigned) div_up(2 * n_mine, 256), 256, 0, stream>>>(2 * n_mine, echo_s.get(),
                                                                                 echo_r.get(), bad.get());
        BT_HIP_CHECK(hipGetLastError());
        int64_t h_bad = -1;
        BT_HIP_CHECK(hipMemcpyAsync(&h_bad, bad.get(), 8, hipMemcpyDeviceToHost, stream));
        BT_HIP_CHECK(hipStreamSynchronize(stream));
        out->loopback_records = n_mine;
        out->loopback_mismatches = h_bad;
    }
    {
        std::vector<int64_t> sob((size_t) nranks), scb((size_t) nranks), rob((size_t) nranks), rcb((size_t) nranks);
        for (int q = 0; q < nranks; ++q) {
            sob[q] = s_off[q] * 16; scb[q] = s_cnt[q] * 16; rob[q] = r_off[q] * 16; rcb[q] = r_cnt[q] * 16;
        }
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send_rec.get(), sob.data(), scb.data(),
                                   (char *) halo_rec.get(), rob.data(), rcb.data(), biggest, false, nullptr));
        BT_HIP_CHECK(hipStreamSynchronize(stream));       // (host vectors)
    }

    // -- the box set: top levels from the plan, my deep boxes, the halo ----------------------------
    std::vector<uint64_t> t_paths;
    std::vector<int32_t> t_meta, t_gid;
    std::vector<int8_t> t_mine;
    std::vector<int32_t> level_starts(1, 0);
    for (int lev = 0; lev < ntop_levels; ++lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        for (int64_t pth = 0; pth < n; ++pth) {
            if (!pl.exists[lev][pth]) continue;
            const bool internal = pl.split[lev][pth];
            // tree.py:109-145 with sources = targets: children on both sides, or a leaf that is both
            int32_t flags = internal ? (BT_BOX_HAS_SOURCE_CHILD_BOXES | BT_BOX_HAS_TARGET_CHILD_BOXES)
                                     : (BT_BOX_IS_SOURCE_BOX | BT_BOX_IS_TARGET_BOX);
            if (!internal && pl.sep_targets) {
                // a leaf is a source box iff it holds sources, a target box iff targets (tbk:1258-1262)
                const int64_t ns = pl.src_counts[lev][pth], nt = pl.counts[lev][pth] - ns;
                flags = (ns > 0 ? BT_BOX_IS_SOURCE_BOX : 0) | (nt > 0 ? BT_BOX_IS_TARGET_BOX : 0);
            }
            // lists of the shared internal boxes are built by every rank, those of a top LEAF
            // only by the rank that owns its cells
            const int64_t first_cell = pth << (D * (k - lev));
            t_paths.push_back((uint64_t) pth);
            t_meta.push_back(lev | (flags << 8));
            t_gid.push_back(num->level_start_box_nrs[lev] + pl.index[lev][pth]);
            t_mine.push_back((internal || pl.owner[first_cell] == rank) ? 1 : 0);
        }
        level_starts.push_back((int32_t) t_paths.size());
    }
    const int64_t ntop = (int64_t) t_paths.size();
    const int64_t nd = n_mine + nrecv;
    const int64_t B = ntop + nd;
    if (B > 0x7fffffff) { set_error("bt_mgpu_let_build: more than 2^31-1 boxes"); return BT_ERR_UNSUPPORTED; }
    BT_CHECK(ms->let_paths.alloc(ctx->pool, B));
    BT_CHECK(ms->let_meta.alloc(ctx->pool, B));
    BT_CHECK(ms->let_gid.alloc(ctx->pool, B));
    BT_CHECK(ms->let_mask.alloc(ctx->pool, B));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_paths.get(), t_paths.data(), (size_t) ntop * 8, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_meta.get(), t_meta.data(), (size_t) ntop * 4, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_gid.get(), t_gid.data(), (size_t) ntop * 4, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_mask.get(), t_mine.data(), (size_t) ntop, hipMemcpyHostToDevice, stream));

    LetLevelInfo h_info{};
    for (int l = 0; l <= BT_MAX_LEVELS; ++l) { h_info.level_first[l] = -1; h_info.mine_last[l] = -1; }
    if (nd > 0) {
        Buf<uint64_t> key_a, key_b, d_path;
        Buf<uint32_t> ord_a, ord_b;
        Buf<int32_t> d_meta, d_gid;
        Buf<LetLevelInfo> info;
        BT_CHECK(key_a.alloc(ctx->pool, nd)); BT_CHECK(key_b.alloc(ctx->pool, nd));
        BT_CHECK(ord_a.alloc(ctx->pool, nd)); BT_CHECK(ord_b.alloc(ctx->pool, nd));
        BT_CHECK(d_path.alloc(ctx->pool, nd)); BT_CHECK(d_meta.alloc(ctx->pool, nd)); BT_CHECK(d_gid.alloc(ctx->pool, nd));
        BT_CHECK(info.alloc(ctx->pool, 1));
        let_deep_kernel<<<(unsigned) div_up(nd, 256), 256, 0, stream>>>(
            n_mine, nrecv, b0, paths.get(), tree->box_levels, tree->box_flags, box_ids, halo_rec.get(),
            pathbits, key_a.get(), d_path.get(), d_meta.get(), d_gid.get());
        BT_HIP_CHECK(hipGetLastError());
        // (level, Morton path) order; equal keys cannot occur (a box is sent by its one owner)
        bool in_b = false;
        BT_CHECK(radix_sort_pairs<uint64_t>(ctx, key_a.get(), ord_a.get(), key_b.get(), ord_b.get(), nd, 0,
                                            pathbits + levbits, true, &in_b));
        LetLevelInfo init{};
        for (int l = 0; l <= BT_MAX_LEVELS; ++l) { init.level_first[l] = -1; init.mine_first[l] = 0; init.mine_last[l] = -1; }
        BT_HIP_CHECK(hipMemcpyAsync(info.get(), &init, sizeof(init), hipMemcpyHostToDevice, stream));
        let_place_kernel<<<(unsigned) div_up(nd, 256), 256, 0, stream>>>(
            nd, n_mine, ntop, in_b ? ord_b.get() : ord_a.get(), d_path.get(), d_meta.get(), d_gid.get(),
            ms->let_paths.get(), ms->let_meta.get(), ms->let_gid.get(), ms->let_mask.get(), info.get());
        BT_HIP_CHECK(hipGetLastError());
        BT_HIP_CHECK(hipMemcpyAsync(&h_info, info.get(), sizeof(h_info), hipMemcpyDeviceToHost, stream));
    }
    BT_CHECK(bt::check_status(ctx));           // waits; the sort and the scans report here

    // -- level starts; the range of a level that holds this rank's boxes ---------------------------
    for (int lev = 0; lev < nlev; ++lev) {
        const int32_t s = level_starts[(size_t) lev];
        if (lev < ntop_levels) {
            out->active_level_ranges[lev][0] = s;
            out->active_level_ranges[lev][1] = level_starts[(size_t) lev + 1];
            continue;
        }
        // the sorted deep boxes of this level: from its first position to the next level's
        int32_t cnt = 0;
        if (h_info.level_first[lev] >= 0) {
            int32_t end = (int32_t) nd;
            for (int l2 = lev + 1; l2 <= BT_MAX_LEVELS; ++l2)
                if (h_info.level_first[l2] >= 0) { end = h_info.level_first[l2]; break; }
            cnt = end - h_info.level_first[lev];
        }
        if (h_info.mine_runs[lev] > 0) {
            if (h_info.mine_runs[lev] != 1) {
                set_error("bt_mgpu_let_build: the rank's boxes of level %d are not one run", lev);
                return BT_ERR_INTERNAL;
            }
            out->active_level_ranges[lev][0] = (int32_t) (ntop + h_info.mine_first[lev]);
            out->active_level_ranges[lev][1] = (int32_t) (ntop + h_info.mine_last[lev] + 1);
        } else {
            out->active_level_ranges[lev][0] = out->active_level_ranges[lev][1] = s;
        }
        level_starts.push_back(s + cnt);
    }
    if (level_starts.back() != (int32_t) B) {
        set_error("bt_mgpu_let_build: level counts (%d) do not add up to the box count (%lld)",
                  level_starts.back(), (long long) B);
        return BT_ERR_INTERNAL;
    }
    ms->let_level_starts = level_starts;
    ms->let_nlevels = nlev;
    ms->let_dims = D;
    ms->let_kind = tree->coord_kind;
    out->nboxes = B;
    out->aligned_nboxes = div_up(B, 32) * 32;
    out->nlevels = nlev;
    for (int l = 0; l <= nlev; ++l) out->level_start_box_nrs[l] = level_starts[(size_t) l];
    out->halo_boxes_sent = nsend;
    out->halo_boxes_received = nrecv;
    return BT_OK;
}

// A rank that leaves a collective entry with an error tells the local group, so that its peers
// (threads waiting at a barrier of the same collective) return an error instead of waiting.
static int peer_result(bt_mgpu_comm *comm, int status)
{
    if (status != BT_OK && comm && comm->kind == 1 && comm->group) comm->group->fail();
    return status;
}

int bt_mgpu_exchange(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_params *p, bt_mgpu_shard *out)
{
    return peer_result(comm, bt_mgpu_exchange_body(ctx, comm, p, out));
}

int bt_mgpu_number(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                   int32_t *box_ids, bt_mgpu_numbering *out)
{
    return peer_result(comm, bt_mgpu_number_body(ctx, comm, tree, box_ids, out));
}

int bt_mgpu_let_build(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_local_tree *tree,
                      const int32_t *box_ids, const bt_mgpu_numbering *num, int well_sep_is_n_away,
                      bt_mgpu_let_sizes *out)
{
    return peer_result(comm, bt_mgpu_let_build_body(ctx, comm, tree, box_ids, num, well_sep_is_n_away, out));
}

int bt_mgpu_let_export(bt_context *ctx, const bt_mgpu_let_arrays *o)
{
    bt::CallScope bt_call_scope_(ctx);
    MgpuState *ms = ctx ? ctx->mgpu : nullptr;
    if (!ctx || !o || !ms || ms->let_nlevels < 1 || !o->box_centers || !o->box_parent_ids || !o->box_child_ids
            || !o->box_levels || !o->box_flags) {
        set_error("bt_mgpu_let_export: invalid argument, or no bt_mgpu_let_build before it");
        return BT_ERR_INVALID;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    const TopPlan &pl = ms->plan;
    const int64_t B = ms->let_level_starts.back();
    const int64_t aligned = div_up(B, 32) * 32;
    hipStream_t stream = ctx->stream;
    let_split_meta_kernel<<<(unsigned) div_up(B, 256), 256, 0, stream>>>(B, ms->let_meta.get(), o->box_levels,
                                                                         o->box_flags);
    BT_HIP_CHECK(hipGetLastError());
    if (o->global_box_ids)
        BT_HIP_CHECK(hipMemcpyAsync(o->global_box_ids, ms->let_gid.get(), (size_t) B * 4, hipMemcpyDeviceToDevice, stream));
    if (o->target_boxes_mask)
        BT_HIP_CHECK(hipMemcpyAsync(o->target_boxes_mask, ms->let_mask.get(), (size_t) B, hipMemcpyDeviceToDevice, stream));
    BT_HIP_CHECK(hipMemsetAsync(o->box_parent_ids, 0, (size_t) B * 4, stream));
    const size_t cs = ms->let_kind == BT_F64 ? 8 : 4;
    BT_HIP_CHECK(hipMemsetAsync(o->box_centers, 0, (size_t) ms->let_dims * (size_t) aligned * cs, stream));
    BT_CHECK(bt_let_build(ctx, ms->let_dims, ms->let_kind, ms->let_nlevels, ms->let_level_starts.data(),
                          ms->let_paths.get(), aligned, pl.bbox_min, pl.bbox_max, pl.root_extent,
                          o->box_parent_ids, o->box_child_ids, o->box_centers));
    ms->let_paths.reset(); ms->let_meta.reset(); ms->let_gid.reset(); ms->let_mask.reset();
    ms->let_nlevels = 0;
    return BT_OK;
}

}  // extern "C"
