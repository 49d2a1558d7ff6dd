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
#include "bt_shm_group.hpp"

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
// the path nccl() bound its entry points through ("" = by search), once it has
bool &nccl_bound() { static bool b = false; return b; }
std::string &nccl_bound_path() { static std::string p; return p; }

Nccl &nccl()
{
    static Nccl n = [] {
        Nccl r;
        void *h = nullptr;
        nccl_bound() = true;
        nccl_bound_path() = nccl_library_path();
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

constexpr int64_t MESSAGE_LIMIT_BYTES = (int64_t) 512 << 20;   // see LAB_NOTES.md section 6 (RCCL, > 1 GB)

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
    int kind = 0;                 // 0: RCCL, 1: threads of one process, 2: processes over shared memory
    int rank = 0, nranks = 1;
    nccl_comm_t nccl = nullptr;
    LocalGroup *group = nullptr;
    bt::ShmGroup *shm = nullptr;  // kind 2 (owned)
    ~bt_mgpu_comm() { delete shm; }
    bool self_loopback = false;   // RCCL: a rank's message to itself travels as ncclSend/ncclRecv too
    // ranks as threads: what this rank publishes for its peers to read lives as long as the
    // communicator (a rank that leaves a collective on a failed barrier must not take the
    // vector its peers are still reading with it)
    std::vector<int64_t> published;
};

namespace {

enum { RED_SUM_I64 = 0, RED_MIN_F64 = 1 };

// kind 2: device bytes to and from the shared segment, on the context's stream, complete at return
int shm_to_host(void *user, void *host, const void *dev, size_t bytes)
{
    hipStream_t stream = *(hipStream_t *) user;
    if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return 1;
    return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1;
}
int shm_to_dev(void *user, void *dev, const void *host, size_t bytes)
{
    hipStream_t stream = *(hipStream_t *) user;
    if (hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return 1;
    return hipStreamSynchronize(stream) == hipSuccess ? 0 : 1;
}
#define BT_SHM_CHECK(c, expr)                                                     \
    do {                                                                          \
        if (!(expr)) {                                                            \
            ::bt::set_error("shared-memory group, rank %d: %s", (c)->rank, (c)->shm->last_error()); \
            return BT_ERR_INTERNAL;                                               \
        }                                                                         \
    } while (0)

// in-place all-reduce of a small device array (8-byte elements)
int comm_all_reduce(bt_mgpu_comm *c, hipStream_t stream, void *dev, size_t count, int what)
{
    if (c->kind == 0) {
        Nccl &nc = nccl();
        BT_NCCL_CHECK(nc.AllReduce(dev, dev, count, what == RED_SUM_I64 ? NCCL_INT64 : NCCL_FLOAT64,
                                   what == RED_SUM_I64 ? NCCL_SUM : NCCL_MIN, c->nccl, stream));
        return BT_OK;
    }
    if (c->kind == 2) {
        bt::ShmMover mv{&stream, shm_to_host, shm_to_dev};
        BT_SHM_CHECK(c, c->shm->all_reduce(mv, dev, count, what == RED_MIN_F64));
        return BT_OK;
    }
    LocalGroup *g = c->group;
    std::vector<int64_t> &mine = c->published;
    std::vector<int64_t> res(count);
    mine.resize(count);
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
    if (c->kind == 2) {
        bt::ShmMover mv{&stream, shm_to_host, shm_to_dev};
        BT_SHM_CHECK(c, c->shm->all_gather(mv, send, recv, bytes));
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
    if (c->kind == 2) {
        // (the rank's own segment went by device copy above, or was written in place)
        bt::ShmMover mv{&stream, shm_to_host, shm_to_dev};
        int rounds = 1;
        BT_SHM_CHECK(c, c->shm->all_to_all_v(mv, send, s_off, s_cnt, recv, r_off, r_cnt, /*skip_self=*/true, &rounds));
        if (rounds_out) *rounds_out = std::max(1, rounds);
        return BT_OK;
    }
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

// column `col` of [n][len] records: its low 32 bits as a dense int32 array (refine weights)
template <class U>
__global__ __launch_bounds__(256) void record_column_i32_kernel(int64_t n, int len, int col, const U *rec, int32_t *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (int32_t) (uint32_t) rec[i * len + col];
}

// column `col` of [n][len] records as a dense array
template <class U>
__global__ __launch_bounds__(256) void record_column_kernel(int64_t n, int len, int col, const U *rec, U *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = rec[i * len + col];
}

// mm = {min[D], -max[D], -1 if the rank has separate targets}: the identity of MIN first
__global__ void mm_init_kernel(double *mm, int D, double flag)
{
    for (int i = 0; i < 2 * D; ++i) mm[i] = 1.7976931348623158e+308;
    mm[2 * D] = flag;
}

// Root box from the all-reduced (min, -max), tree_build.py:462-476 in the coordinate type;
// out = {min[3], max[3], extent, 0}, the layout the key kernel reads (bt_tree.hip).
template <class T>
__global__ void root_box_from_mm_kernel(const double *mm, int D, T *out)
{
    T ext = 0;
    for (int ax = 0; ax < D; ++ax) {
        const T w = (T) (-mm[D + ax]) - (T) mm[ax];
        ext = w > ext ? w : ext;
    }
    const T re = ext * (T) (1 + 1e-4);
    for (int ax = 0; ax < 3; ++ax) {
        out[ax] = ax < D ? (T) mm[ax] : (T) 0;
        out[3 + ax] = ax < D ? (T) ((T) mm[ax] + re) : (T) 0;
    }
    out[6] = re;
    out[7] = (T) 0;
}

// where a rank's own segment goes, per particle set: {send offset, receive offset} in records;
// the receive offset is what the ranks below send to this one (column of the gathered matrix)
__global__ void self_offsets_kernel(const int64_t *matrix, int nranks, int rank, int64_t s_off0,
                                    int64_t s_off1, int loop_self, int64_t *out)
{
    const int64_t row = 2 * (int64_t) nranks;
    for (int s = 0; s < 2; ++s) {
        const int64_t so = s == 0 ? s_off0 : s_off1;
        int64_t r = 0;
        for (int q = 0; q < rank; ++q) r += matrix[q * row + (int64_t) s * nranks + rank];
        out[2 * s] = so;
        out[2 * s + 1] = loop_self ? so : r;
    }
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
    bool has_stay = false;                        // particles with extents: `stay` is filled
    double stick_out_factor = 0;                  // ... and this is how far a target may leave its box
    std::vector<std::vector<int64_t>> stay;       // [k+1][C^lev]: particles that stay in the box
    std::vector<std::vector<int64_t>> stay_src;   // ... the sources among them
    bool weighted = false;                        // refine weights: the same two tables for weights
    std::vector<std::vector<int64_t>> wcounts, wstay;
    std::vector<std::vector<int64_t>> counts;     // [k+1][C^lev]
    std::vector<std::vector<char>> exists, split;
    std::vector<std::vector<int32_t>> index;      // number of a box among the existing boxes of its level
    std::vector<int32_t> nboxes;                  // [k+1]
    bool sep_targets = false;                     // the build has separate targets
    std::vector<std::vector<int64_t>> src_counts; // [k+1][C^lev]: the sources among `counts`
    std::vector<int32_t> owner;                   // [C^k]
    std::vector<int64_t> unit_start;              // [C^k] scratch of compute_plan
    std::vector<int64_t> unit_a, unit_b, rank_from;   // ... more of it (kept: no allocation per call)
    std::vector<int64_t> prefix;                  // [C^k + 1]
    double bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0}, root_extent = 0;
};

// index of box (level, path) in a table over the boxes of levels 0..k
inline int64_t top_table_offset(int D, int level)
{
    int64_t off = 0, pw = 1;
    for (int l = 0; l < level; ++l) { off += pw; pw <<= D; }
    return off;
}

// hist: particles per level-k cell (a particle with an extent that stays in a box above level k
// counts for the first cell under that box); stay (or NULL: point particles): per box of levels
// 0..k the particles that stay in it, top_table_offset(level) + path.
// wcell / wstay (or NULL: unit weights): the same two tables for the particles' refine weights; the
// split rule then compares weights with mpb (= max_leaf_refine_weight), existence stays a matter
// of counts.
void compute_plan(int D, int k, int64_t mpb, int nranks, const int64_t *hist, const int64_t *stay, TopPlan &pl,
                  const int64_t *wcell = nullptr, const int64_t *wstay = nullptr)
{
    const int C = 1 << D;
    const int64_t ncells = (int64_t) 1 << (D * k);
    pl.D = D; pl.k = k; pl.mpb = mpb; pl.nranks = nranks;
    pl.has_stay = stay != nullptr;
    // counts[lev][i]: the particles that arrive in box (lev, i) -- its cumulative count
    pl.counts.resize((size_t) k + 1);
    pl.stay.resize(stay ? (size_t) k + 1 : 0);
    pl.counts[k].assign(hist, hist + ncells);
    if (stay) {
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            pl.stay[lev].assign(stay + top_table_offset(D, lev), stay + top_table_offset(D, lev) + n);
        }
        // the stayers of the boxes above level k were counted at their first cell
        for (int lev = 0; lev < k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            const int sh = D * (k - lev);
            for (int64_t i = 0; i < n; ++i) pl.counts[k][(size_t) (i << sh)] -= pl.stay[lev][(size_t) i];
        }
    }
    for (int lev = k - 1; lev >= 0; --lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        pl.counts[lev].resize((size_t) n);
        const int64_t *below = pl.counts[lev + 1].data();
        int64_t *here = pl.counts[lev].data();
        for (int64_t i = 0; i < n; ++i) {
            int64_t sum = stay ? pl.stay[lev][(size_t) i] : 0;
            for (int m = 0; m < C; ++m) sum += below[i * C + m];
            here[i] = sum;
        }
    }
    // the same pyramid for refine weights
    pl.weighted = wcell != nullptr;
    pl.wcounts.resize(wcell ? (size_t) k + 1 : 0);
    pl.wstay.resize(wcell ? (size_t) k + 1 : 0);
    if (wcell) {
        pl.wcounts[k].assign(wcell, wcell + ncells);
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            if (wstay) pl.wstay[lev].assign(wstay + top_table_offset(D, lev), wstay + top_table_offset(D, lev) + n);
            else pl.wstay[lev].assign((size_t) n, 0);
        }
        for (int lev = 0; lev < k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            const int sh = D * (k - lev);
            for (int64_t i = 0; i < n; ++i) pl.wcounts[k][(size_t) (i << sh)] -= pl.wstay[lev][(size_t) i];
        }
        for (int lev = k - 1; lev >= 0; --lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            pl.wcounts[lev].resize((size_t) n);
            for (int64_t i = 0; i < n; ++i) {
                int64_t sum = pl.wstay[lev][(size_t) i];
                for (int m = 0; m < C; ++m) sum += pl.wcounts[lev + 1][(size_t) (i * C + m)];
                pl.wcounts[lev][(size_t) i] = sum;
            }
        }
    }
    // unit[c]: the first cell of the top-tree leaf a cell lies in (cells under a leaf above
    // level k travel together), found top-down: -1 while the boxes above still split
    std::vector<int64_t> &unit_start = pl.unit_start;
    unit_start.resize((size_t) ncells);
    pl.valid = mpb > 0;
    if (mpb > 0) {
        pl.exists.resize((size_t) k + 1);
        pl.split.resize((size_t) k + 1);
        pl.index.resize((size_t) k + 1);
        pl.nboxes.assign((size_t) k + 1, 0);
        pl.exists[0].assign(1, 1);
        // (scratch kept in the plan: two 256-KiB vectors allocated and first touched on every
        // call cost the exchange a third of its one idle wait)
        std::vector<int64_t> &unit_here = pl.unit_a, &unit_next = pl.unit_b;
        unit_here.assign(1, -1);
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            pl.split[lev].resize((size_t) n);
            pl.index[lev].resize((size_t) n);
            const char *ex = pl.exists[lev].data();
            const int64_t *cnt = pl.counts[lev].data();
            const int64_t *st = stay ? pl.stay[lev].data() : nullptr;
            char *sp = pl.split[lev].data();
            int32_t *ix = pl.index[lev].data();
            int32_t run = 0;
            const int64_t *wc = wcell ? pl.wcounts[lev].data() : nullptr;
            const int64_t *ws = wcell ? pl.wstay[lev].data() : nullptr;
            for (int64_t i = 0; i < n; ++i) {
                // tbk:569-591: what is bound for the children decides (its weight, if weighted)
                sp[i] = ex[i] && (wc ? wc[i] - ws[i] : cnt[i] - (st ? st[i] : 0)) > mpb;
                run += ex[i] ? 1 : 0;
                ix[i] = run - 1;
            }
            pl.nboxes[lev] = run;
            if (lev < k) {
                pl.exists[lev + 1].resize((size_t) n * C);
                char *exn = pl.exists[lev + 1].data();
                const int64_t *cntn = pl.counts[lev + 1].data();
                unit_next.resize((size_t) n * C);
                const int sh = D * (k - lev);
                for (int64_t i = 0; i < n; ++i) {
                    // a box above the ownership level that does not split is a leaf (or absent):
                    // everything below it is one unit
                    const int64_t u = unit_here[(size_t) i] >= 0 ? unit_here[(size_t) i] : (sp[i] ? -1 : (i << sh));
                    for (int m = 0; m < C; ++m) {
                        exn[i * C + m] = sp[i] && cntn[i * C + m] > 0;
                        unit_next[(size_t) (i * C + m)] = u;
                    }
                }
                unit_here.swap(unit_next);
            }
        }
        for (int64_t c = 0; c < ncells; ++c) unit_start[(size_t) c] = unit_here[(size_t) c] >= 0 ? unit_here[(size_t) c] : c;
    } else {
        for (int64_t c = 0; c < ncells; ++c) unit_start[(size_t) c] = c;
    }
    // contiguous Morton ranges balanced by particle count: a cell goes to the rank whose
    // ideal range contains the first particle of its unit, floor(prefix * nranks / total).
    // Units ascend with the cells, so the rank only ever grows: no division per cell.
    // (The particles that stay in an internal box above level k sit at its first cell: they
    // precede everything below the box in the tree's particle order, and the rank that owns
    // that cell owns them -- a rank's particles stay one contiguous range of the global order.)
    const int64_t total = std::max<int64_t>(pl.counts[0][0], 1);
    pl.prefix.resize((size_t) ncells + 1);
    pl.prefix[0] = 0;
    for (int64_t c = 0; c < ncells; ++c) pl.prefix[(size_t) c + 1] = pl.prefix[(size_t) c] + hist[c];
    pl.owner.resize((size_t) ncells);
    // rank r + 1 begins at the first unit whose prefix reaches ceil((r + 1) * total / nranks):
    // (r + 1) * total <= prefix * nranks  <=>  prefix >= that threshold (positive integers)
    std::vector<int64_t> &from = pl.rank_from;
    from.resize((size_t) nranks);
    for (int q = 0; q + 1 < nranks; ++q)
        from[(size_t) q] = (int64_t) (((__int128) (q + 1) * total + nranks - 1) / nranks);
    int32_t r = 0;
    const int64_t *us = unit_start.data(), *pf = pl.prefix.data();
    int32_t *ow = pl.owner.data();
    for (int64_t c = 0; c < ncells; ++c) {
        if (us[c] == c)
            while (r < nranks - 1 && pf[c] >= from[(size_t) r]) ++r;
        ow[c] = r;
    }
}

}  // namespace

// Pinned host memory for what the device reads from, or writes to, the host during the calls of
// one sharded build: bump-allocated, handed back wholesale when the next exchange begins (or
// when it has grown large), after the stream has passed the last call that used it.
struct PinArena {
    struct Block { char *p; size_t cap; };
    std::vector<Block> blocks;
    size_t used = 0, total = 0;
    void *get(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t) 255;
        if (blocks.empty() || blocks.back().cap - used < bytes) {
            const size_t cap = std::max<size_t>(bytes, (size_t) 4 << 20);
            char *p = nullptr;
            if (hipHostMalloc((void **) &p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
            blocks.push_back({p, cap});
            used = 0;
        }
        void *r = blocks.back().p + used;
        used += bytes; total += bytes;
        return r;
    }
    void reset()
    {
        // keep the newest (largest-so-far) block
        while (blocks.size() > 1) { (void) hipHostFree(blocks.front().p); blocks.erase(blocks.begin()); }
        used = 0; total = 0;
    }
    ~PinArena() { for (auto &b : blocks) (void) hipHostFree(b.p); }
};

// The stable send plan of one particle set of the last exchange: what bt_mgpu_route needs to move
// any per-particle array between the caller's order and the owners' (receive-buffer) order.
struct RoutePlan {
    bool valid = false;
    int rank = 0, nranks = 0;
    bool loop_self = false;
    int64_t n = 0, nrecv = 0;                // particles of the chunk / owned after the exchange
    Buf<uint8_t> owners;                     // [n] owner of every chunk particle
    Buf<int32_t> offsets;                    // [nranks * nwaves + 1] scanned per-(owner, tile) counts
    std::vector<int64_t> s_cnt, r_cnt;       // records to / from every rank
    int64_t biggest = 0;                     // largest rank-to-rank message of the job, records
    int64_t chunk_offset = 0, total = 0;     // global id of the chunk's first particle; all chunks
};

struct MgpuState {
    PinArena arena;
    RoutePlan route[2];              // sources, separate targets
    Buf<unsigned char> points;       // received particles, interleaved [n_owned][dims]
    Buf<unsigned char> tpoints;      // ... separate targets ([n][dims + 1] with radii)
    Buf<unsigned char> tradii;       // ... their radii, dense (the tree build reads them so)
    Buf<int64_t> cell_prefix;        // [C^top_level + 1]
    Buf<int64_t> top_tables;         // extents: arrivals, stayers per box of levels 0..top_level
    TopPlan plan;                    // of the last exchange on this context
    std::vector<int64_t> ghist;      // combined global cell histogram of that exchange
    std::vector<int64_t> stay_all;   // ... and stayers (sources + targets) per top box
    std::vector<int64_t> wcell;      // ... and refine weights per cell
    Buf<int32_t> weights;            // received particles' refine weights: sources, then targets
    hipEvent_t ev[2] = {nullptr, nullptr};   // around the payload all-to-all-v
    hipEvent_t ev_counts = nullptr;          // the counts matrix has reached the host
    hipEvent_t ev_done = nullptr;            // the last exchange's use of the pinned block is over
    bool done_pending = false, a2a_pending = false;
    float a2a_ms = 0.f;
    // pinned host block of the exchange: what the host reads (root box, histograms, counts
    // matrix) and what the device reads from the host (owner table, send counts, cell prefix)
    char *pin = nullptr;
    size_t pin_cap = 0;
    ~MgpuState()
    {
        for (auto &e : ev) if (e) (void) hipEventDestroy(e);
        if (ev_counts) (void) hipEventDestroy(ev_counts);
        if (ev_done) (void) hipEventDestroy(ev_done);
        if (pin) (void) hipHostFree(pin);
    }
    // local essential tree between bt_mgpu_let_build and bt_mgpu_let_export
    Buf<uint64_t> let_paths;         // [B] level-major, Morton order within a level
    Buf<int32_t> let_meta, let_gid;  // level | flags << 8; global box number
    Buf<int8_t> let_mask;            // 1: lists are built for this box on this rank
    // targets with extents: target bounding boxes [2][D][B] (min, then max) in the coordinate
    // type, and cumulative source counts [B]
    Buf<unsigned char> let_tbb;
    Buf<int32_t> let_srccum;
    bool let_ext = false;
    Buf<int32_t> let_sizes;          // subtree sizes of the deep boxes (every rank passed its own)
    bool let_has_sizes = false;
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

// calls that read or write the pinned arena end with arena_mark (the stream has passed the call
// once ev_done has); the first call of a build, and any call that finds the arena large, waits
// for the mark and takes the arena back
int arena_mark(bt_context *ctx, MgpuState *ms)
{
    if (!ms->ev_done) BT_HIP_CHECK(hipEventCreateWithFlags(&ms->ev_done, hipEventDisableTiming));
    BT_HIP_CHECK(hipEventRecord(ms->ev_done, ctx->stream));
    ms->done_pending = true;
    return BT_OK;
}

int arena_reclaim(MgpuState *ms, bool always)
{
    if (!always && ms->arena.total < ((size_t) 64 << 20)) return BT_OK;
    if (ms->done_pending) { BT_HIP_CHECK(hipEventSynchronize(ms->ev_done)); ms->done_pending = false; }
    ms->arena.reset();
    return BT_OK;
}

#define BT_ARENA(ptr, T, ms, count)                                                              \
    T *ptr = (T *) (ms)->arena.get(sizeof(T) * (size_t) std::max<int64_t>((int64_t) (count), 1));   \
    if (!ptr) { set_error("pinned host memory: allocation of %lld bytes failed",                   \
                          (long long) (sizeof(T) * (size_t) (count))); return BT_ERR_ALLOC; }

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

// Where the deep boxes (levels > k) of one sender go in the LET.  Ranks own ascending Morton
// ranges of cells, a rank's boxes of a level are in Morton order and a peer sends its halo boxes
// in that order: the boxes of a level are, in Morton order, the blocks of the ranks in rank order
// -- positions follow from the per-(sender, level) counts, no sort.
struct LetBlocks {
    int32_t src_start[BT_MAX_LEVELS + 1];   // index of the sender's first box of a level in its sequence
    int32_t dst_start[BT_MAX_LEVELS + 1];   // LET position of that box
};

// this rank's own deep boxes: the tail [b0, nb) of its level-major local tree
__global__ __launch_bounds__(256) void let_scatter_own_kernel(int64_t n_mine, int64_t b0, LetBlocks blk,
        const uint64_t *paths, const uint8_t *levels, const uint8_t *flags, const int32_t *gids,
        uint64_t *all_paths, int32_t *all_meta, int32_t *all_gid, int8_t *mask)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mine) return;
    const int64_t b = b0 + i;
    const int lev = levels[b];
    const int64_t dst = (int64_t) blk.dst_start[lev] + (b - blk.src_start[lev]);
    all_paths[dst] = paths[b];
    all_meta[dst] = (int32_t) lev | ((int32_t) flags[b] << 8);
    all_gid[dst] = gids[b];
    mask[dst] = 1;
}

// the halo records of one peer: (path, level | flags << 8 | global id << 32), level-major
__global__ __launch_bounds__(256) void let_scatter_halo_kernel(int64_t n, LetBlocks blk, const uint64_t *rec,
        uint64_t *all_paths, int32_t *all_meta, int32_t *all_gid, int8_t *mask, DeviceStatus *status)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint64_t w = rec[2 * j + 1];
    const int32_t meta = (int32_t) (w & 0xffffffffu);
    const int lev = meta & 0xff;
    // (a record outside the block its level was announced with: the counts and the records disagree)
    if (lev > BT_MAX_LEVELS || j < blk.src_start[lev]) { atomicExch(&status->internal, 71); return; }
    const int64_t dst = (int64_t) blk.dst_start[lev] + (j - blk.src_start[lev]);
    all_paths[dst] = rec[2 * j];
    all_meta[dst] = meta;
    all_gid[dst] = (int32_t) (w >> 32);
    mask[dst] = 0;
}

__global__ void let_flag_kernel(int32_t *p) { *p = 1; }

// subtree sizes: of the boxes peer q needs, of this rank's own deep boxes, of a peer's halo boxes
__global__ __launch_bounds__(256) void let_pack_sizes_kernel(int64_t n, NeedPred pr, const int32_t *pos,
        const int32_t *sizes, int32_t *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !pr(i)) return;
    out[pos[i]] = sizes[pr.b0 + i];
}

__global__ __launch_bounds__(256) void let_scatter_own_sizes_kernel(int64_t n_mine, int64_t b0, LetBlocks blk,
        const uint8_t *levels, const int32_t *sizes, int32_t *all_sizes)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mine) return;
    const int64_t b = b0 + i;
    const int lev = levels[b];
    all_sizes[(int64_t) blk.dst_start[lev] + (b - blk.src_start[lev])] = sizes[b];
}

__global__ __launch_bounds__(256) void let_scatter_halo_sizes_kernel(int64_t n, LetBlocks blk, const uint64_t *rec,
        const int32_t *in, int32_t *all_sizes)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int lev = (int) (rec[2 * j + 1] & 0xffu);
    if (lev > BT_MAX_LEVELS || j < blk.src_start[lev]) return;      // (let_scatter_halo_kernel reports it)
    all_sizes[(int64_t) blk.dst_start[lev] + (j - blk.src_start[lev])] = in[j];
}

// the shared top boxes, one level per launch, bottom-up over the LET's child table
__global__ __launch_bounds__(256) void let_top_sizes_kernel(int32_t b0, int32_t nb, int C, int64_t aligned,
        const int32_t *child, int32_t *sizes)
{
    const int32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nb) return;
    const int32_t b = b0 + i;
    int32_t s = 1;
    for (int m = 0; m < C; ++m) {
        const int32_t c = child[(int64_t) m * aligned + b];
        if (c) s += sizes[c];
    }
    sizes[b] = s;
}

// extras of a box whose targets have extents: bounding box of its targets, cumulative source count
template <class T>
struct LetExtras {
    const T *bmin, *bmax;          // [D][aligned] of the local tree
    const int32_t *srccum;
    int64_t aligned;
    int D;
};

// extras of the boxes peer q needs: [n][2 D] coordinates (min.., max..), then [n] counts
template <class T>
__global__ __launch_bounds__(256) void let_pack_extras_kernel(int64_t n, NeedPred pr, const int32_t *pos,
        LetExtras<T> ex, T *out_box, int32_t *out_cnt)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !pr(i)) return;
    const int64_t b = pr.b0 + i, at = pos[i];
    for (int ax = 0; ax < ex.D; ++ax) {
        out_box[at * 2 * ex.D + ax] = ex.bmin[(int64_t) ax * ex.aligned + b];
        out_box[at * 2 * ex.D + ex.D + ax] = ex.bmax[(int64_t) ax * ex.aligned + b];
    }
    out_cnt[at] = ex.srccum[b];
}

// own deep boxes' extras into the LET ([2][D][B] boxes, [B] counts)
template <class T>
__global__ __launch_bounds__(256) void let_scatter_own_extras_kernel(int64_t n_mine, int64_t b0, LetBlocks blk,
        const uint8_t *levels, LetExtras<T> ex, int64_t B, T *tbb, int32_t *srccum)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n_mine) return;
    const int64_t b = b0 + i;
    const int lev = levels[b];
    const int64_t dst = (int64_t) blk.dst_start[lev] + (b - blk.src_start[lev]);
    for (int ax = 0; ax < ex.D; ++ax) {
        tbb[(int64_t) ax * B + dst] = ex.bmin[(int64_t) ax * ex.aligned + b];
        tbb[((int64_t) ex.D + ax) * B + dst] = ex.bmax[(int64_t) ax * ex.aligned + b];
    }
    srccum[dst] = ex.srccum[b];
}

template <class T>
__global__ __launch_bounds__(256) void let_scatter_halo_extras_kernel(int64_t n, LetBlocks blk, const uint64_t *rec,
        const T *in_box, const int32_t *in_cnt, int D, int64_t B, T *tbb, int32_t *srccum)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int lev = (int) (rec[2 * j + 1] & 0xffu);
    if (lev > BT_MAX_LEVELS || j < blk.src_start[lev]) return;
    const int64_t dst = (int64_t) blk.dst_start[lev] + (j - blk.src_start[lev]);
    for (int ax = 0; ax < D; ++ax) {
        tbb[(int64_t) ax * B + dst] = in_box[j * 2 * D + ax];
        tbb[((int64_t) D + ax) * B + dst] = in_box[j * 2 * D + D + ax];
    }
    srccum[dst] = in_cnt[j];
}

// shared top boxes: (min, -max) of the local tree's boxes of levels <= k into mm[gid][2 D]
// (doubles, +inf where this rank has no such box) for the all-reduce(MIN) over the ranks
template <class T>
__global__ __launch_bounds__(256) void let_top_tbb_gather_kernel(int64_t ntop_local, const int32_t *gids,
        LetExtras<T> ex, double *mm)
{
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= ntop_local) return;
    const int64_t g = gids[b];
    for (int ax = 0; ax < ex.D; ++ax) {
        mm[g * 2 * ex.D + ax] = (double) ex.bmin[(int64_t) ax * ex.aligned + b];
        mm[g * 2 * ex.D + ex.D + ax] = -(double) ex.bmax[(int64_t) ax * ex.aligned + b];
    }
}

__global__ __launch_bounds__(256) void fill_f64_kernel(int64_t n, double v, double *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}

// ... and back, into the first ntop boxes of the LET
template <class T>
__global__ __launch_bounds__(256) void let_top_tbb_place_kernel(int64_t ntop, int D, const double *mm,
        const int32_t *top_srccum, int64_t B, T *tbb, int32_t *srccum)
{
    const int64_t g = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (g >= ntop) return;
    for (int ax = 0; ax < D; ++ax) {
        tbb[(int64_t) ax * B + g] = (T) mm[g * 2 * D + ax];
        tbb[((int64_t) D + ax) * B + g] = (T) (-mm[g * 2 * D + D + ax]);
    }
    srccum[g] = top_srccum[g];
}

// [2][D][B] -> the caller's [D][aligned] arrays
template <class T>
__global__ __launch_bounds__(256) void let_tbb_export_kernel(int64_t B, int64_t aligned, int D, const T *tbb,
        T *bmin, T *bmax)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= aligned) return;
    for (int ax = 0; ax < D; ++ax) {
        bmin[(int64_t) ax * aligned + i] = i < B ? tbb[(int64_t) ax * B + i] : (T) 0;
        bmax[(int64_t) ax * aligned + i] = i < B ? tbb[((int64_t) D + ax) * B + i] : (T) 0;
    }
}

// row[q][l] = number of my level-l boxes peer q needs: differences of the scanned predicate at
// the level boundaries of the local tree
struct LocalLevels { int32_t nlevels; int32_t start[BT_MAX_LEVELS + 2]; };
__global__ void let_level_counts_kernel(const int32_t *pos, int64_t b0, int k, LocalLevels ls, int32_t *row_q)
{
    const int l = k + 1 + (int) threadIdx.x;
    if (l >= ls.nlevels) return;
    row_q[l] = pos[ls.start[l + 1] - b0] - pos[ls.start[l] - b0];
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
    // Targets with extents.  (i) The targets that stay in an internal box B above level k are held
    // by the owner q of B's first cell, and their List 1 takes every leaf under B and beside it
    // (traversal.py:470-550).  (ii) The List-3 walk of a target box with extents does not stop
    // at the first box that is not adjacent: where the stick-out regions may still meet it files
    // the box under "close" and goes on down (traversal.py:830-868), anywhere inside the
    // colleagues of the target box -- for a target box above level k, leaf or not, that is more
    // than `ring` cells around its cells.  q needs my non-empty cells under B and within ring_B
    // cells of it: the colleagues' width (ring * side), cut down to the shell in which a
    // separation criterion can fail -- a target's extent leaves its box by at most
    // stick_out_factor box radii, and a source box of a cell or less that fails the criterion is
    // within one cell of that (linf criteria), or within 3 source radii of the round region of
    // radius sqrt(d) (1 + stick_out_factor) box radii (static_l2).
    if (pl.has_stay) {
        for (int lev = 0; lev < k; ++lev) {
            const int64_t nb = (int64_t) 1 << (D * lev);
            const int side = 1 << (k - lev);
            const double sof = pl.stick_out_factor;
            const int shell_linf = (int) std::ceil(sof * side / 2) + 1;
            const int shell_l2 = (int) std::ceil(side / 2.0 * (std::sqrt((double) D) * (1 + sof) - 1) + 2);
            const int ring_b = std::min(ring * side, std::max(std::max(shell_linf, shell_l2), ring));
            for (int64_t pth = 0; pth < nb; ++pth) {
                if (!pl.exists[lev][pth]) continue;
                const bool internal = pl.split[lev][pth];
                const int64_t ntg = internal ? pl.stay[lev][pth] - pl.stay_src[lev][pth]
                                             : pl.counts[lev][pth] - pl.src_counts[lev][pth];
                if (ntg <= 0) continue;
                const int64_t first = pth << (D * (k - lev));
                const int q = pl.owner[(size_t) first];
                if (q == rank) continue;
                int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
                for (int ax = 0; ax < D; ++ax) {
                    const int v = xyz_of_cell[(size_t) first * 3 + ax];      // the block's low corner
                    lo[ax] = std::max(0, v - ring_b); hi[ax] = std::min(n - 1, v + side - 1 + ring_b);
                }
                int p[3];
                for (p[0] = lo[0]; p[0] <= hi[0]; ++p[0])
                    for (p[1] = lo[1]; p[1] <= hi[1]; ++p[1])
                        for (p[2] = lo[2]; p[2] <= hi[2]; ++p[2]) {
                            int64_t gi = 0;
                            for (int ax = 0; ax < D; ++ax) gi = gi * n + p[ax];
                            const int64_t c = cell_of_grid[(size_t) gi];
                            if (pl.owner[(size_t) c] != rank || pl.counts[k][(size_t) c] <= 0) continue;
                            bits[(size_t) c * nwords + (q >> 6)] |= 1ull << (q & 63);
                            any_for_peer[q] = 1;
                        }
            }
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
    c->self_loopback = lb && atoi(lb) != 0 && nranks == 1;    // (a switch for worlds of one rank)
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

int bt_mgpu_comm_shm(const char *name, int rank, int nranks, int64_t slot_bytes, double timeout_s, bt_mgpu_comm **out)
{
    if (!out) { set_error("bt_mgpu_comm_shm: invalid argument"); return BT_ERR_INVALID; }
    std::string err;
    bt::ShmGroup *g = bt::ShmGroup::open(name, rank, nranks, slot_bytes > 0 ? slot_bytes : (int64_t) 64 << 20,
                                         timeout_s > 0 ? timeout_s : 120.0, &err);
    if (!g) { set_error("bt_mgpu_comm_shm: %s", err.c_str()); return BT_ERR_INVALID; }
    bt_mgpu_comm *c = new bt_mgpu_comm();
    c->kind = 2; c->rank = rank; c->nranks = nranks; c->shm = g;
    *out = c;
    return BT_OK;
}

void bt_mgpu_comm_destroy(bt_mgpu_comm *c) { delete c; }

int bt_mgpu_use_rccl_library(const char *path)
{
    if (!path || !*path) { set_error("bt_mgpu_use_rccl_library: empty path"); return BT_ERR_INVALID; }
    // the entry points are bound once per process: naming another image afterwards cannot take
    // effect, and a communicator made by that image must not be driven through this one
    if (nccl_bound() && nccl_bound_path() != path) {
        set_error("bt_mgpu_use_rccl_library: the RCCL entry points are already bound (%s); %s cannot be used "
                  "in this process any more", nccl_bound_path().empty() ? "image found by search"
                                                                         : nccl_bound_path().c_str(), path);
        return BT_ERR_INVALID;
    }
    nccl_library_path() = path;
    return BT_OK;
}

int bt_mgpu_comm_set_self_loopback(bt_mgpu_comm *c, int on)
{
    // (the number of point-to-point rounds follows from the largest message of any rank; with the
    // own segments in play that maximum would have to include every rank's diagonal -- the switch
    // exists for worlds of one rank, where nothing else travels)
    if (c && on && c->nranks > 1) {
        set_error("bt_mgpu_comm_set_self_loopback: a test switch for communicators of one rank (this one has %d)",
                  c->nranks);
        return BT_ERR_INVALID;
    }
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
    static thread_local TopPlan pl;      // (its tables are reused from call to call)
    compute_plan(dims, top_level, max_particles_in_box, nranks, global_hist, nullptr, pl);
    std::copy(pl.owner.begin(), pl.owner.end(), owner_of_cell);
    if (cell_prefix) std::copy(pl.prefix.begin(), pl.prefix.end(), cell_prefix);
    return BT_OK;
}

// bt_mgpu_plan for particles with extents: stay_table (levels 0..top_level, box (level, path) at
// (C^level - 1) / (C - 1) + path) counts the particles that stay in a box; cell_hist counts such
// a particle at the first cell under its box.  box_arrive / box_split (same indexing, may be
// NULL): the particles that arrive in every box, and whether it exists and splits.
int bt_mgpu_plan_ext(int dims, int top_level, int64_t max_particles_in_box, int nranks,
                     const int64_t *cell_hist, const int64_t *stay_table, int32_t *owner_of_cell,
                     int64_t *cell_prefix, int64_t *box_arrive, uint8_t *box_split)
{
    if (dims < 1 || dims > 3 || top_level < 1 || dims * top_level > 30 || nranks < 1
            || !cell_hist || !owner_of_cell) {
        set_error("bt_mgpu_plan_ext: invalid argument");
        return BT_ERR_INVALID;
    }
    static thread_local TopPlan pl;
    compute_plan(dims, top_level, max_particles_in_box, nranks, cell_hist, stay_table, pl);
    std::copy(pl.owner.begin(), pl.owner.end(), owner_of_cell);
    if (cell_prefix) std::copy(pl.prefix.begin(), pl.prefix.end(), cell_prefix);
    for (int lev = 0; lev <= top_level; ++lev) {
        const int64_t off = top_table_offset(dims, lev), n = (int64_t) 1 << (dims * lev);
        if (box_arrive) std::copy(pl.counts[lev].begin(), pl.counts[lev].begin() + n, box_arrive + off);
        if (box_split)
            for (int64_t i = 0; i < n; ++i)
                box_split[off + i] = pl.valid ? (uint8_t) ((pl.exists[lev][(size_t) i] ? 1 : 0) | (pl.split[lev][(size_t) i] ? 2 : 0)) : 0;
    }
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
    if (ctx->mgpu) { ctx->mgpu->route[0].valid = false; ctx->mgpu->route[1].valid = false; }
    // refine weights (the limit is a parameter of the job; a NULL array means unit weights)
    const bool weighted = p->max_leaf_refine_weight > 0;
    const int32_t *wset[2] = {weighted ? p->source_refine_weights : nullptr,
                              weighted ? p->target_refine_weights : nullptr};
    const int64_t split_limit = weighted ? (int64_t) p->max_leaf_refine_weight : p->max_particles_in_box;
    // targets with extents (the same on every rank: extent_norm is a parameter of the job)
    const bool ext = p->extent_norm != BT_NORM_NONE;
    if (ext && ((p->extent_norm != BT_NORM_LINF && p->extent_norm != BT_NORM_L2)
                || (p->ntargets > 0 && !p->target_radii) || split_limit <= 0)) {
        set_error("bt_mgpu_exchange: extents need extent_norm linf or l2, target_radii on every rank "
                  "that has targets, and max_particles_in_box > 0");
        return BT_ERR_INVALID;
    }
    const void *rset[2] = {nullptr, ext ? p->target_radii : nullptr};
    hipStream_t stream = ctx->stream;

    MgpuState *ms = mgpu_state(ctx);
    const int k = p->top_level > 0 ? p->top_level : (D == 3 ? 5 : D == 2 ? 7 : 12);
    const int64_t ncells = (int64_t) 1 << (D * k);
    const int64_t row = 2 * (int64_t) nranks;
    // histogram words: source cells, target cells, and with extents the targets that stay in
    // a box of levels 0..k
    const int64_t ntop1 = top_table_offset(D, k + 1);
    const int64_t nh = 2 * ncells + (ext ? 2 * ntop1 : 0);       // + stayers: sources, targets
    // ... and 64-bit words for refine weights: per source cell, per target cell, per box of the
    // stayers (both kinds together)
    const int64_t nw = weighted ? 2 * ncells + (ext ? ntop1 : 0) : 0;
    if (!ms->ev[0]) {
        BT_HIP_CHECK(hipEventCreate(&ms->ev[0]));
        BT_HIP_CHECK(hipEventCreate(&ms->ev[1]));
        BT_HIP_CHECK(hipEventCreateWithFlags(&ms->ev_counts, hipEventDisableTiming));
        BT_HIP_CHECK(hipEventCreateWithFlags(&ms->ev_done, hipEventDisableTiming));
    }
    // the previous build on this context may still be reading the pinned blocks
    BT_CHECK(arena_reclaim(ms, true));
    // pinned block: [box: 16 doubles][local hist: 2 ncells i32][global hist: 2 ncells i64]
    // [owner: ncells i32][send counts: row i64][matrix: row * nranks i64][prefix: ncells + 1 i64]
    const size_t o_box = 0, o_local = 128, o_ghist = o_local + (((size_t) nh * 4 + 7) & ~(size_t) 7),
        o_owner = o_ghist + (size_t) (nh + nw) * 8, o_send = o_owner + (size_t) ncells * 4,
        o_matrix = o_send + (size_t) row * 8, o_prefix = o_matrix + (size_t) row * nranks * 8,
        o_tables = o_prefix + (size_t) (ncells + 1) * 8,
        pin_need = o_tables + ((ext || weighted) ? (size_t) ntop1 * 16 : 0);
    if (ms->pin_cap < pin_need) {
        if (ms->pin) { BT_HIP_CHECK(hipStreamSynchronize(stream)); (void) hipHostFree(ms->pin); ms->pin = nullptr; ms->pin_cap = 0; }
        BT_HIP_CHECK(hipHostMalloc((void **) &ms->pin, pin_need, hipHostMallocDefault));
        ms->pin_cap = pin_need;
    }
    double *h_box = (double *) (ms->pin + o_box);             // mm[2 D + 1], then the root box (T[8]) at +8 doubles
    int32_t *h_local = (int32_t *) (ms->pin + o_local);
    int64_t *h_ghist2 = (int64_t *) (ms->pin + o_ghist);
    int32_t *h_owner = (int32_t *) (ms->pin + o_owner);
    int64_t *h_send = (int64_t *) (ms->pin + o_send);
    int64_t *h_matrix = (int64_t *) (ms->pin + o_matrix);
    int64_t *h_prefix = (int64_t *) (ms->pin + o_prefix);
    int64_t *h_tables = (int64_t *) (ms->pin + o_tables);       // arrive [ntop1], stay [ntop1]

    // ---- 1. global bounding box -> root box, all on the device -----------------------------
    Buf<double> mm;
    Buf<unsigned char> rootbox_d;
    BT_CHECK(mm.alloc(ctx->pool, 2 * D + 1));
    BT_CHECK(rootbox_d.alloc(ctx->pool, 64));
    mm_init_kernel<<<1, 1, 0, stream>>>(mm.get(), D, p->ntargets > 0 ? -1.0 : 0.0);   // MIN: -1 if some rank has targets
    for (int s = 0; s < 2; ++s)
        BT_CHECK(bt::bbox_minmax_device(ctx, D, p->coord_kind, cset[s], rset[s], nset[s], mm.get()));
    BT_CHECK(comm_all_reduce(comm, stream, mm.get(), 2 * D + 1, RED_MIN_F64));
    if (f64) root_box_from_mm_kernel<double><<<1, 1, 0, stream>>>(mm.get(), D, (double *) rootbox_d.get());
    else root_box_from_mm_kernel<float><<<1, 1, 0, stream>>>(mm.get(), D, (float *) rootbox_d.get());
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::copy_to_pinned(ctx, h_box, mm.get(), sizeof(double) * (2 * D + 1)));
    BT_CHECK(bt::copy_to_pinned(ctx, h_box + 8, rootbox_d.get(), 8 * (size_t) es));

    // ---- 2. cell histograms (sources, targets), all-reduced -------------------------------
    // (always both: whether the job has separate targets is only known with the bounding box)
    Buf<uint32_t> cells[2];
    Buf<int32_t> hist32, owner_d;
    Buf<int64_t> hist64;
    BT_CHECK(hist32.alloc(ctx->pool, nh));
    BT_CHECK(hist64.alloc(ctx->pool, nh + nw));
    if (nw > 0) BT_HIP_CHECK(hipMemsetAsync(hist64.get() + nh, 0, (size_t) nw * 8, stream));
    BT_CHECK(owner_d.alloc(ctx->pool, ncells));
    BT_HIP_CHECK(hipMemsetAsync(hist32.get(), 0, (size_t) nh * 4, stream));
    for (int s = 0; s < 2; ++s) {
        if (nset[s] == 0) continue;
        BT_CHECK(cells[s].alloc(ctx->pool, nset[s]));
        if (ext)
            // (the sources too: points, but with a stick-out factor near zero the rounding of the
            // test lets points stay in boxes, in the build and therefore here)
            BT_CHECK(bt::morton_cells_ext_device(ctx, D, p->coord_kind, cset[s], rset[s], nset[s], rootbox_d.get(),
                                                 k, p->stick_out_factor, p->extent_norm, cells[s].get(),
                                                 hist32.get() + s * ncells, hist32.get() + 2 * ncells + s * ntop1,
                                                 wset[s], weighted ? hist64.get() + nh + 2 * ncells : nullptr));
        else
            BT_CHECK(bt::morton_cells_device(ctx, D, p->coord_kind, cset[s], nset[s], rootbox_d.get(), k,
                                             cells[s].get(), hist32.get() + s * ncells));
    }
    if (weighted)
        for (int s = 0; s < 2; ++s)
            BT_CHECK(bt::weight_hist_device(ctx, cells[s].get(), wset[s], nset[s], (int) ncells,
                                            hist64.get() + nh + s * ncells));
    widen_hist_kernel<<<(unsigned) div_up(nh, 256), 256, 0, stream>>>(nh, hist32.get(), hist64.get());
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(bt::copy_to_pinned(ctx, h_local, hist32.get(), (size_t) nh * 4));
    BT_CHECK(comm_all_reduce(comm, stream, hist64.get(), (size_t) (nh + nw), RED_SUM_I64));
    BT_CHECK(bt::copy_to_pinned(ctx, h_ghist2, hist64.get(), (size_t) (nh + nw) * 8));
    bt::host_trace("x:queued");
    BT_CHECK(bt::sync_stream(ctx));                       // the one wait the GPU idles through
    bt::host_trace("x:hist here");

    const bool sep = h_box[2 * D] < 0;
    const int nsets = sep ? 2 : 1;
    if (h_box[0] > -h_box[D]) {
        set_error("bt_mgpu_exchange: no rank has any particle");
        return BT_ERR_INVALID;
    }
    double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}, root_extent = 0;
    for (int ax = 0; ax < D; ++ax) {
        bmin[ax] = f64 ? h_box[8 + ax] : (double) ((const float *) (h_box + 8))[ax];
        bmax[ax] = f64 ? h_box[8 + 3 + ax] : (double) ((const float *) (h_box + 8))[3 + ax];
    }
    root_extent = f64 ? h_box[8 + 6] : (double) ((const float *) (h_box + 8))[6];

    // ---- 3. plan (host, identical on all ranks), counts --------------------------------------
    std::vector<int64_t> &ghist = ms->ghist;
    ghist.resize((size_t) ncells);
    for (int64_t c = 0; c < ncells; ++c) ghist[(size_t) c] = h_ghist2[c] + h_ghist2[ncells + c];
    TopPlan &pl = ms->plan;
    std::vector<int64_t> &stay_all = ms->stay_all;
    if (ext) {
        stay_all.resize((size_t) ntop1);
        for (int64_t i = 0; i < ntop1; ++i) stay_all[(size_t) i] = h_ghist2[2 * ncells + i] + h_ghist2[2 * ncells + ntop1 + i];
    }
    std::vector<int64_t> &wcell = ms->wcell;
    if (weighted) {
        wcell.resize((size_t) ncells);
        for (int64_t c = 0; c < ncells; ++c) wcell[(size_t) c] = h_ghist2[nh + c] + h_ghist2[nh + ncells + c];
    }
    compute_plan(D, k, split_limit, nranks, ghist.data(), ext ? stay_all.data() : nullptr, pl,
                 weighted ? wcell.data() : nullptr, (weighted && ext) ? h_ghist2 + nh + 2 * ncells : nullptr);
    if (ext && !sep) {
        set_error("bt_mgpu_exchange: extent_norm is set but no rank has targets");
        return BT_ERR_INVALID;
    }
    bt::host_trace("x:planned");
    memcpy(h_owner, pl.owner.data(), (size_t) ncells * 4);
    BT_HIP_CHECK(hipMemcpyAsync(owner_d.get(), h_owner, (size_t) ncells * 4, hipMemcpyHostToDevice, stream));
    int64_t send_counts[2 * BT_MGPU_MAX_RANKS], nrecv_of[2] = {0, 0};       // [set][owner]
    for (int64_t i = 0; i < row; ++i) send_counts[i] = 0;
    for (int s = 0; s < nsets; ++s)
        for (int64_t c = 0; c < ncells; ++c) {
            send_counts[(size_t) s * nranks + pl.owner[(size_t) c]] += h_local[(size_t) s * ncells + c];
            if (pl.owner[(size_t) c] == rank) nrecv_of[s] += h_ghist2[(size_t) s * ncells + c];
        }
    memcpy(h_send, send_counts, (size_t) row * 8);
    Buf<int64_t> counts_d, self_d;
    BT_CHECK(counts_d.alloc(ctx->pool, row * (nranks + 1)));
    BT_CHECK(self_d.alloc(ctx->pool, 4));
    BT_HIP_CHECK(hipMemcpyAsync(counts_d.get(), h_send, (size_t) row * 8, hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_gather(comm, stream, counts_d.get(), counts_d.get() + row, (size_t) row * 8));
    const bool loop_self = comm->kind == 0 && comm->self_loopback;
    int64_t s_off_self[2] = {0, 0};
    for (int s = 0; s < 2; ++s)
        for (int r = 0; r < rank; ++r) s_off_self[s] += send_counts[(size_t) s * nranks + r];
    self_offsets_kernel<<<1, 1, 0, stream>>>(counts_d.get() + row, nranks, rank, s_off_self[0], s_off_self[1],
                                             loop_self ? 1 : 0, self_d.get());
    BT_HIP_CHECK(hipGetLastError());
    // [sender][set][receiver]: on its way to the host while the partition sweeps run
    BT_CHECK(bt::copy_to_pinned(ctx, h_matrix, counts_d.get() + row, (size_t) row * nranks * 8));
    BT_HIP_CHECK(hipEventRecord(ms->ev_counts, stream));
    // values and bytes per particle record of a set (targets with extents carry their radius)
    const int vals_of[2] = {D + (weighted ? 1 : 0), D + (ext ? 1 : 0) + (weighted ? 1 : 0)};
    const int64_t rec_of[2] = {(int64_t) vals_of[0] * es, (int64_t) vals_of[1] * es};

    // ---- 4. payload: interleaved coordinates, one exchange per particle set -------------------
    // One sweep over the coordinates per set: stable partition by owner into the send buffer, the
    // segment this rank keeps straight into the receive buffer (bt_shard.hip) -- or, with the test
    // switch bt_mgpu_comm_set_self_loopback, into the send buffer like any other, to make the
    // trip through ncclSend / ncclRecv.
    Buf<unsigned char> send[2], wide_w[2];
    unsigned char *points_of[2] = {nullptr, nullptr};
    for (int s = 0; s < nsets; ++s) {
        const int64_t n = nset[s];
        BT_CHECK(send[s].alloc(ctx->pool, n * rec_of[s]));
        const int64_t points_bytes = std::max<int64_t>(nrecv_of[s], 1) * rec_of[s];
        Buf<unsigned char> &own = s == 0 ? ms->points : ms->tpoints;
        if (p->alloc) {
            points_of[s] = (unsigned char *) p->alloc(p->alloc_user, points_bytes);
            if (!points_of[s]) { set_error("bt_mgpu_exchange: the caller's allocator returned NULL"); return BT_ERR_ALLOC; }
            own.reset();
        } else {
            BT_CHECK(own.alloc(ctx->pool, points_bytes));
            points_of[s] = own.get();
        }
        const void *arrays[BT_MAX_DIMS + 2];
        int nv = 0;
        for (int ax = 0; ax < D; ++ax) arrays[nv++] = cset[s][ax];
        if (s == 1 && ext) arrays[nv++] = rset[s];
        if (weighted) {
            BT_CHECK(wide_w[s].alloc(ctx->pool, std::max<int64_t>(n, 1) * es));
            BT_CHECK(bt::widen_weights_device(ctx, wset[s], n, es, wide_w[s].get()));
            arrays[nv++] = wide_w[s].get();
        }
        BT_CHECK(bt::partition_pack_device(ctx, vals_of[s], es, arrays, cells[s].get(), n, owner_d.get(), (int) ncells,
                                           nranks, rank, self_d.get() + 2 * s, send[s].get(),
                                           loop_self ? send[s].get() : points_of[s], &ms->route[s].owners,
                                           &ms->route[s].offsets));
    }
    bt::host_trace("x:sweeps queued");
    // (host work the sweeps do not wait for: the GPU is busy meanwhile)
    // which top boxes hold sources / targets (flags of the shared top levels)
    pl.sep_targets = sep;
    pl.stick_out_factor = ext ? p->stick_out_factor : 0;
    pl.src_counts.resize((size_t) k + 1);
    pl.src_counts[k].assign(h_ghist2, h_ghist2 + ncells);
    pl.stay_src.resize(ext ? (size_t) k + 1 : 0);
    if (ext) {
        const int64_t *ss = h_ghist2 + 2 * ncells;
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev), off = top_table_offset(D, lev);
            pl.stay_src[lev].assign(ss + off, ss + off + n);
        }
        // (sources that stay above level k were counted at their box's first cell)
        for (int lev = 0; lev < k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev);
            const int sh = D * (k - lev);
            for (int64_t i = 0; i < n; ++i) pl.src_counts[k][(size_t) (i << sh)] -= pl.stay_src[lev][(size_t) i];
        }
    }
    for (int lev = k - 1; lev >= 0; --lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        pl.src_counts[lev].resize((size_t) n);
        const int64_t *below = pl.src_counts[lev + 1].data();
        for (int64_t i = 0; i < n; ++i) {
            int64_t sum = ext ? pl.stay_src[lev][(size_t) i] : 0;
            for (int m = 0; m < (1 << D); ++m) sum += below[i * (1 << D) + m];
            pl.src_counts[lev][(size_t) i] = sum;
        }
    }
    for (int ax = 0; ax < 3; ++ax) { pl.bbox_min[ax] = bmin[ax]; pl.bbox_max[ax] = bmax[ax]; }
    pl.root_extent = root_extent;
    BT_HIP_CHECK(hipEventSynchronize(ms->ev_counts));     // (the GPU is busy with the sweeps)
    bt::host_trace("x:counts here");
    int32_t rounds_total = 0;
    int64_t bytes_sent = 0;
    BT_HIP_CHECK(hipEventRecord(ms->ev[0], stream));
    for (int s = 0; s < nsets; ++s) {
        const int64_t rec = rec_of[s];
        std::vector<int64_t> s_cnt_b((size_t) nranks), r_cnt_b((size_t) nranks), s_off_b((size_t) nranks),
            r_off_b((size_t) nranks);
        int64_t biggest = 0, s_off = 0, r_off = 0;
        for (int r = 0; r < nranks; ++r) {
            const int64_t sc = send_counts[(size_t) s * nranks + r];
            const int64_t rc = h_matrix[(size_t) r * row + (size_t) s * nranks + rank];
            if (h_matrix[(size_t) rank * row + (size_t) s * nranks + r] != sc) {
                set_error("bt_mgpu_exchange: the gathered counts disagree with this rank's own");
                return BT_ERR_INTERNAL;
            }
            s_cnt_b[r] = sc * rec; r_cnt_b[r] = rc * rec;
            s_off_b[r] = s_off * rec; r_off_b[r] = r_off * rec;
            s_off += sc; r_off += rc;
            for (int q = 0; q < nranks; ++q)
                if (q != r) biggest = std::max(biggest, h_matrix[(size_t) r * row + (size_t) s * nranks + q] * rec);
        }
        if (r_off != nrecv_of[s]) {
            set_error("bt_mgpu_exchange: rank %d receives %lld particles, its cells hold %lld", rank,
                      (long long) r_off, (long long) nrecv_of[s]);
            return BT_ERR_INTERNAL;
        }
        // (self-loopback is refused on communicators of more than one rank: `biggest`, which fixes
        // the number of rounds, must be the same on every rank)
        int32_t rounds = 1;
        {
            // the plan bt_mgpu_route and bt_mgpu_global_ids work from
            RoutePlan &rp = ms->route[s];
            rp.rank = rank; rp.nranks = nranks; rp.loop_self = loop_self;
            rp.n = nset[s]; rp.nrecv = nrecv_of[s];
            rp.s_cnt.assign((size_t) nranks, 0); rp.r_cnt.assign((size_t) nranks, 0);
            rp.chunk_offset = 0; rp.total = 0;
            for (int q = 0; q < nranks; ++q) {
                rp.s_cnt[(size_t) q] = send_counts[(size_t) s * nranks + q];
                rp.r_cnt[(size_t) q] = h_matrix[(size_t) q * row + (size_t) s * nranks + rank];
                int64_t nq = 0;
                for (int r = 0; r < nranks; ++r) nq += h_matrix[(size_t) q * row + (size_t) s * nranks + r];
                if (q < rank) rp.chunk_offset += nq;
                rp.total += nq;
            }
            rp.biggest = (loop_self ? std::max(biggest, s_cnt_b[rank]) : biggest) / rec;
            rp.valid = true;
        }
        if (!loop_self) { s_cnt_b[rank] = 0; r_cnt_b[rank] = 0; }       // (own segment: packed in place)
        else biggest = std::max(biggest, s_cnt_b[rank]);
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send[s].get(), s_off_b.data(), s_cnt_b.data(),
                                   (char *) points_of[s], r_off_b.data(), r_cnt_b.data(), biggest, !loop_self, &rounds));
        rounds_total += rounds;
        bytes_sent += (nset[s] - (loop_self ? 0 : send_counts[(size_t) s * nranks + rank])) * rec;
    }
    BT_HIP_CHECK(hipEventRecord(ms->ev[1], stream));
    ms->a2a_pending = true;
    void *radii_out = nullptr;
    if (ext && nsets == 2) {
        // the received targets' radii as an array of their own (bt_tree_params.target_radii is dense)
        const int64_t nt = nrecv_of[1];
        if (p->alloc) {
            radii_out = p->alloc(p->alloc_user, std::max<int64_t>(nt, 1) * es);
            if (!radii_out) { set_error("bt_mgpu_exchange: the caller's allocator returned NULL"); return BT_ERR_ALLOC; }
        } else {
            BT_CHECK(ms->tradii.alloc(ctx->pool, std::max<int64_t>(nt, 1) * es));
            radii_out = ms->tradii.get();
        }
        if (nt > 0) {
            if (f64) record_column_kernel<uint64_t><<<(unsigned) div_up(nt, 256), 256, 0, stream>>>(
                nt, vals_of[1], D, (const uint64_t *) points_of[1], (uint64_t *) radii_out);
            else record_column_kernel<uint32_t><<<(unsigned) div_up(nt, 256), 256, 0, stream>>>(
                nt, vals_of[1], D, (const uint32_t *) points_of[1], (uint32_t *) radii_out);
            BT_HIP_CHECK(hipGetLastError());
        }
    }
    BT_CHECK(ms->cell_prefix.alloc(ctx->pool, ncells + 1));
    memcpy(h_prefix, pl.prefix.data(), (size_t) (ncells + 1) * 8);
    BT_HIP_CHECK(hipMemcpyAsync(ms->cell_prefix.get(), h_prefix, (size_t) (ncells + 1) * 8,
                                hipMemcpyHostToDevice, stream));
    if (weighted) {
        // the received particles' weights as the dense int32 array bt_tree_params.refine_weights is
        const int64_t n0 = nrecv_of[0], n1 = nsets == 2 ? nrecv_of[1] : 0;
        BT_CHECK(ms->weights.alloc(ctx->pool, std::max<int64_t>(n0 + n1, 1)));
        for (int s = 0; s < nsets; ++s) {
            const int64_t n = s == 0 ? n0 : n1;
            if (n == 0) continue;
            int32_t *dst = ms->weights.get() + (s == 0 ? 0 : n0);
            if (f64) record_column_i32_kernel<uint64_t><<<(unsigned) div_up(n, 256), 256, 0, stream>>>(
                n, vals_of[s], vals_of[s] - 1, (const uint64_t *) points_of[s], dst);
            else record_column_i32_kernel<uint32_t><<<(unsigned) div_up(n, 256), 256, 0, stream>>>(
                n, vals_of[s], vals_of[s] - 1, (const uint32_t *) points_of[s], dst);
        }
        BT_HIP_CHECK(hipGetLastError());
        out->refine_weights = ms->weights.get();
    }
    if (ext || weighted) {
        // the top of the global tree for bt_tree_build: arrivals and stayers per top box (their
        // refine weights in a weighted job)
        BT_CHECK(ms->top_tables.alloc(ctx->pool, 2 * ntop1));
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (D * lev), off = top_table_offset(D, lev);
            if (weighted) {
                memcpy(h_tables + off, pl.wcounts[lev].data(), (size_t) n * 8);
                memcpy(h_tables + ntop1 + off, pl.wstay[lev].data(), (size_t) n * 8);
            } else {
                memcpy(h_tables + off, pl.counts[lev].data(), (size_t) n * 8);
                memcpy(h_tables + ntop1 + off, pl.stay[lev].data(), (size_t) n * 8);
            }
        }
        BT_HIP_CHECK(hipMemcpyAsync(ms->top_tables.get(), h_tables, (size_t) ntop1 * 16, hipMemcpyHostToDevice, stream));
        out->top_box_arrive = ms->top_tables.get();
        out->top_box_stay = ms->top_tables.get() + ntop1;
    }
    BT_HIP_CHECK(hipEventRecord(ms->ev_done, stream));
    ms->done_pending = true;

    out->target_record_len = vals_of[1];
    out->source_record_len = vals_of[0];
    out->target_radii = radii_out;
    out->n_owned = nrecv_of[0];
    out->points = points_of[0];
    out->n_owned_targets = nrecv_of[1];
    out->target_points = points_of[1];
    for (int ax = 0; ax < D; ++ax) { out->bbox_min[ax] = bmin[ax]; out->bbox_max[ax] = bmax[ax]; }
    out->root_extent = root_extent;
    out->top_level = k;
    out->top_cell_prefix = split_limit > 0 ? ms->cell_prefix.get() : nullptr;
    out->bytes_sent = bytes_sent;
    out->rounds = rounds_total;
    out->sep_targets = sep ? 1 : 0;
    out->source_chunk_offset = ms->route[0].chunk_offset;
    out->n_global_sources = ms->route[0].total;
    out->n_sent_sources = nset[0] - (loop_self ? 0 : send_counts[(size_t) rank]);
    if (nsets == 2) {
        out->target_chunk_offset = ms->route[1].chunk_offset;
        out->n_global_targets = ms->route[1].total;
        out->n_sent_targets = nset[1] - (loop_self ? 0 : send_counts[(size_t) nranks + rank]);
    }
    out->a2a_ms = -1.f;
    // a stream-ordered context returns with the payload exchange queued (bt_mgpu_exchange_time
    // waits for it); any other waits here
    BT_CHECK(bt::finish_call(ctx));
    if (!ctx->stream_ordered) {
        (void) hipEventElapsedTime(&ms->a2a_ms, ms->ev[0], ms->ev[1]);
        ms->a2a_pending = false;
        out->a2a_ms = ms->a2a_ms;
    }
    return BT_OK;
}


// ---- particle identity: per-particle arrays over the plan of the last exchange ------------------

static int bt_mgpu_route_body(bt_context *ctx, bt_mgpu_comm *comm, int set, int direction, int elem_size,
                              const void *in, void *out, bool ids)
{
    bt::CallScope bt_call_scope_(ctx);
    const char *who = ids ? "bt_mgpu_global_ids" : "bt_mgpu_route";
    if (!ctx || !comm || set < 0 || set > 1 || (elem_size != 4 && elem_size != 8)
            || (direction != BT_ROUTE_TO_OWNERS && direction != BT_ROUTE_TO_CALLERS)) {
        set_error("%s: invalid argument", who);
        return BT_ERR_INVALID;
    }
    MgpuState *ms = ctx->mgpu;
    if (!ms || !ms->route[set].valid) {
        set_error("%s: no exchange of %s on this context", who, set == 0 ? "sources" : "separate targets");
        return BT_ERR_INVALID;
    }
    RoutePlan &rp = ms->route[set];
    if (rp.rank != comm->rank || rp.nranks != comm->nranks) {
        set_error("%s: the communicator (rank %d of %d) is not the exchange's (rank %d of %d)", who, comm->rank,
                  comm->nranks, rp.rank, rp.nranks);
        return BT_ERR_INVALID;
    }
    const bool reverse = direction == BT_ROUTE_TO_CALLERS;
    const int64_t n_in = reverse ? rp.nrecv : rp.n, n_out = reverse ? rp.n : rp.nrecv;
    if ((n_in > 0 && !in && !ids) || (n_out > 0 && !out)) {
        set_error("%s: NULL array", who);
        return BT_ERR_INVALID;
    }
    if (ids && elem_size == 4 && rp.total > (int64_t) INT32_MAX) {
        set_error("bt_mgpu_global_ids: %lld particles do not fit 32-bit ids; ask for 8-byte ids", (long long) rp.total);
        return BT_ERR_UNSUPPORTED;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    BT_CHECK(reset_status(ctx));
    hipStream_t stream = ctx->stream;
    const int rank = rp.rank, nranks = rp.nranks;
    const int64_t es = elem_size;
    // byte offsets and counts of the two layouts: the chunk's send layout (owner-major) and the
    // owner order (senders in rank order)
    std::vector<int64_t> lay_off((size_t) nranks), lay_cnt((size_t) nranks), own_off((size_t) nranks), own_cnt((size_t) nranks);
    int64_t a = 0, b = 0;
    for (int r = 0; r < nranks; ++r) {
        lay_off[(size_t) r] = a * es; lay_cnt[(size_t) r] = rp.s_cnt[(size_t) r] * es; a += rp.s_cnt[(size_t) r];
        own_off[(size_t) r] = b * es; own_cnt[(size_t) r] = rp.r_cnt[(size_t) r] * es; b += rp.r_cnt[(size_t) r];
    }
    const int64_t self_delta = (own_off[(size_t) rank] - lay_off[(size_t) rank]) / es;
    if (!rp.loop_self) { lay_cnt[(size_t) rank] = 0; own_cnt[(size_t) rank] = 0; }     // (own segment: moved by the kernel)
    Buf<unsigned char> lay;
    BT_CHECK(lay.alloc(ctx->pool, std::max<int64_t>(rp.n, 1) * es));
    int32_t rounds = 1;
    if (!reverse) {
        BT_CHECK(bt::route_device(ctx, elem_size, false, ids ? nullptr : in, rp.chunk_offset, lay.get(),
                                  rp.loop_self ? nullptr : out, nullptr, rp.owners.get(), rp.offsets.get(), rp.n,
                                  nranks, rank, self_delta));
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) lay.get(), lay_off.data(), lay_cnt.data(), (char *) out,
                                   own_off.data(), own_cnt.data(), rp.biggest * es, !rp.loop_self, &rounds));
    } else {
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) in, own_off.data(), own_cnt.data(), (char *) lay.get(),
                                   lay_off.data(), lay_cnt.data(), rp.biggest * es, !rp.loop_self, &rounds));
        BT_CHECK(bt::route_device(ctx, elem_size, true, nullptr, 0, lay.get(), rp.loop_self ? nullptr : (void *) in,
                                  out, rp.owners.get(), rp.offsets.get(), rp.n, nranks, rank, self_delta));
    }
    return bt::finish_call(ctx);
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
    BT_CHECK(reset_status(ctx));
    memset(out, 0, sizeof(*out));
    hipStream_t stream = ctx->stream;
    const int rank = comm->rank, nranks = comm->nranks, k = pl.k, nlev = tree->nlevels;
    constexpr int W = BT_MAX_LEVELS + 2;
    BT_CHECK(arena_reclaim(ms, false));
    BT_ARENA(mine, int64_t, ms, W);
    BT_ARENA(all, int64_t, ms, (int64_t) W * nranks);
    for (int l = 0; l < W; ++l) mine[l] = 0;
    for (int l = 0; l < nlev; ++l) mine[l] = tree->level_start_box_nrs[l + 1] - tree->level_start_box_nrs[l];
    mine[BT_MAX_LEVELS] = tree->nsources;
    mine[BT_MAX_LEVELS + 1] = tree->ntargets;
    Buf<int64_t> gath;
    BT_CHECK(gath.alloc(ctx->pool, (int64_t) W * (nranks + 1)));
    BT_HIP_CHECK(hipMemcpyAsync(gath.get(), mine, (size_t) W * 8, hipMemcpyHostToDevice, stream));
    BT_CHECK(comm_all_gather(comm, stream, gath.get(), gath.get() + W, (size_t) W * 8));
    BT_CHECK(bt::copy_to_pinned(ctx, all, gath.get() + W, (size_t) W * nranks * 8));
    BT_CHECK(bt::sync_stream(ctx));
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
    const int ntop_levels = std::min(k + 1, nlev);
    int64_t nindex = 0;
    for (int l = 0; l < ntop_levels; ++l) nindex += (int64_t) pl.index[l].size();
    BT_ARENA(index, int32_t, ms, nindex);
    nindex = 0;
    for (int l = 0; l < ntop_levels; ++l) {
        na.toff[l] = (int32_t) nindex;
        na.gstart[l] = out->level_start_box_nrs[l];
        memcpy(index + nindex, pl.index[l].data(), pl.index[l].size() * 4);
        nindex += (int64_t) pl.index[l].size();
    }
    Buf<int32_t> index_d;
    BT_CHECK(index_d.alloc(ctx->pool, nindex));
    BT_HIP_CHECK(hipMemcpyAsync(index_d.get(), index, (size_t) nindex * 4, hipMemcpyHostToDevice, stream));
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
    BT_CHECK(arena_mark(ctx, ms));
    return bt::finish_call(ctx);
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

    BT_CHECK(reset_status(ctx));
    BT_CHECK(arena_reclaim(ms, false));
    // targets with extents: the traversal also reads the target bounding box and the cumulative
    // source count of every box of the LET
    const bool ext = pl.has_stay;
    if (ext && nb > 0 && (!tree->box_target_bounding_box_min || !tree->box_target_bounding_box_max
                          || !tree->box_source_counts_cumul)) {
        set_error("bt_mgpu_let_build: the exchange had targets with extents; the local tree must come "
                  "with box_target_bounding_box_min / _max and box_source_counts_cumul");
        return BT_ERR_INVALID;
    }
    const size_t es = tree->coord_kind == BT_F64 ? 8 : 4;
    // -- Morton paths of my boxes ----------------------------------------------------------------
    Buf<uint64_t> paths;
    BT_CHECK(paths.alloc(ctx->pool, nb));
    BT_CHECK(bt::box_paths_device(ctx, D, tree->coord_kind, nb, tree->aligned_nboxes, tree->box_centers,
                                  tree->box_levels, pl.bbox_min, pl.root_extent, paths.get()));
    // my deep boxes (levels > k) are the tail of the level-major local tree
    const int64_t b0 = nlev_local > k + 1 ? tree->level_start_box_nrs[k + 1] : nb;
    const int64_t n_mine = nb - b0;

    // -- halo: my deep boxes in the cells other ranks' lists can reach ------------------------------
    const int nwords = (nranks + 63) / 64;
    std::vector<uint64_t> need_bits;
    std::vector<char> any_for_peer;
    cells_needed_by(pl, rank, well_sep_is_n_away, nwords, need_bits, any_for_peer);
    BT_ARENA(h_need, uint64_t, ms, need_bits.size());
    memcpy(h_need, need_bits.data(), need_bits.size() * 8);
    Buf<uint64_t> need_d;
    BT_CHECK(need_d.alloc(ctx->pool, (int64_t) need_bits.size()));
    BT_HIP_CHECK(hipMemcpyAsync(need_d.get(), h_need, need_bits.size() * 8, hipMemcpyHostToDevice, stream));
    std::vector<int> peers;
    for (int q = 0; q < nranks; ++q)
        if (q != rank && any_for_peer[q] && n_mine > 0) peers.push_back(q);
    Buf<int32_t> pos;                 // [peers][n_mine + 1]
    BT_CHECK(pos.alloc(ctx->pool, (int64_t) peers.size() * (n_mine + 1)));
    // row[q][l]: how many of my level-l boxes rank q gets; every rank learns every row -- the
    // whole (sender, receiver, level) table: message sizes, the largest message of the job, and
    // where every box of the LET goes (LetBlocks)
    const int LW = nlev;
    const int64_t rowlen = (int64_t) nranks * LW;
    Buf<int32_t> rows;
    BT_CHECK(rows.alloc(ctx->pool, rowlen * (nranks + 1)));
    BT_HIP_CHECK(hipMemsetAsync(rows.get(), 0, (size_t) rowlen * 4, stream));
    LocalLevels lls{};
    lls.nlevels = nlev_local;
    for (int l = 0; l <= nlev_local; ++l) lls.start[l] = tree->level_start_box_nrs[l];
    // (slot [receiver = this rank][level 0] of a rank's row counts nothing -- no rank sends to
    // itself, level 0 is shared --: it says whether the rank came with subtree sizes)
    if (tree->box_subtree_sizes || nb == 0)
        let_flag_kernel<<<1, 1, 0, stream>>>(rows.get() + (int64_t) rank * LW);
    NeedPred pr{paths.get(), tree->box_levels, need_d.get(), b0, k, D, nwords, 0};
    for (size_t i = 0; i < peers.size(); ++i) {
        pr.q = peers[i];
        int32_t *pp = pos.get() + (int64_t) i * (n_mine + 1);
        BT_CHECK((device_exclusive_scan<int32_t, int32_t>(ctx, pr, n_mine, pp, (int32_t *) nullptr, true)));
        let_level_counts_kernel<<<1, BT_MAX_LEVELS, 0, stream>>>(pp, b0, k, lls, rows.get() + (int64_t) peers[i] * LW);
    }
    BT_HIP_CHECK(hipGetLastError());
    BT_CHECK(comm_all_gather(comm, stream, rows.get(), rows.get() + rowlen, (size_t) rowlen * 4));
    BT_ARENA(table, int32_t, ms, rowlen * nranks);          // [sender][receiver][level]
    BT_CHECK(bt::copy_to_pinned(ctx, table, rows.get() + rowlen, (size_t) rowlen * nranks * 4));
    BT_CHECK(bt::sync_stream(ctx));                         // the one wait of this call
    auto sent = [&](int from, int to) {
        int64_t c = 0;
        for (int l = ntop_levels; l < LW; ++l) c += table[((int64_t) from * nranks + to) * LW + l];
        return c;
    };
    bool with_sizes = true;
    for (int q = 0; q < nranks; ++q) with_sizes = with_sizes && table[((int64_t) q * nranks + q) * LW] != 0;
    std::vector<int64_t> s_cnt((size_t) nranks, 0), s_off((size_t) nranks, 0), r_cnt((size_t) nranks, 0),
        r_off((size_t) nranks, 0);
    int64_t nsend = 0, nrecv = 0, biggest = 0;
    for (int q = 0; q < nranks; ++q) {
        s_cnt[q] = sent(rank, q); s_off[q] = nsend; nsend += s_cnt[q];
        r_cnt[q] = sent(q, rank); r_off[q] = nrecv; nrecv += r_cnt[q];
        for (int t = 0; t < nranks; ++t)
            if (t != q) biggest = std::max(biggest, sent(q, t) * 16);
    }
    Buf<uint64_t> send_rec, halo_rec;
    BT_CHECK(send_rec.alloc(ctx->pool, 2 * std::max<int64_t>(nsend, 1)));
    BT_CHECK(halo_rec.alloc(ctx->pool, 2 * std::max<int64_t>(nrecv, 1)));
    for (size_t i = 0; i < peers.size(); ++i) {
        if (s_cnt[peers[i]] == 0) continue;
        pr.q = peers[i];
        let_pack_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
            n_mine, pr, pos.get() + (int64_t) i * (n_mine + 1), tree->box_flags, box_ids,
            send_rec.get() + 2 * s_off[peers[i]]);
    }
    BT_HIP_CHECK(hipGetLastError());
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
        // (the host arrays are read while the call queues its sends and receives; ranks that are
        // threads wait inside)
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send_rec.get(), sob.data(), scb.data(),
                                   (char *) halo_rec.get(), rob.data(), rcb.data(), biggest, false, nullptr));
    }
    // the extras of the same boxes in the same order: per peer [count][2 D] coordinates, then
    // [count] source counts
    Buf<unsigned char> send_ex, halo_ex;
    // (a multiple of the coordinate width, so that every peer's block starts aligned: the counts
    // are int32 at the head of a region of one coordinate width per box)
    const int64_t exrec = ext ? (int64_t) (2 * D * es + es) : 0;
    if (ext) {
        BT_CHECK(send_ex.alloc(ctx->pool, std::max<int64_t>(nsend, 1) * exrec));
        BT_CHECK(halo_ex.alloc(ctx->pool, std::max<int64_t>(nrecv, 1) * exrec));
        for (size_t i = 0; i < peers.size(); ++i) {
            const int q = peers[i];
            if (s_cnt[q] == 0) continue;
            pr.q = q;
            unsigned char *blk0 = send_ex.get() + s_off[q] * exrec;
            const int32_t *pp = pos.get() + (int64_t) i * (n_mine + 1);
            if (es == 8) {
                LetExtras<double> ex{(const double *) tree->box_target_bounding_box_min,
                                     (const double *) tree->box_target_bounding_box_max,
                                     tree->box_source_counts_cumul, tree->aligned_nboxes, D};
                let_pack_extras_kernel<double><<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
                    n_mine, pr, pp, ex, (double *) blk0, (int32_t *) (blk0 + s_cnt[q] * 2 * D * es));
            } else {
                LetExtras<float> ex{(const float *) tree->box_target_bounding_box_min,
                                    (const float *) tree->box_target_bounding_box_max,
                                    tree->box_source_counts_cumul, tree->aligned_nboxes, D};
                let_pack_extras_kernel<float><<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
                    n_mine, pr, pp, ex, (float *) blk0, (int32_t *) (blk0 + s_cnt[q] * 2 * D * es));
            }
        }
        BT_HIP_CHECK(hipGetLastError());
        std::vector<int64_t> sob((size_t) nranks), scb((size_t) nranks), rob((size_t) nranks), rcb((size_t) nranks);
        for (int q = 0; q < nranks; ++q) {
            sob[q] = s_off[q] * exrec; scb[q] = s_cnt[q] * exrec; rob[q] = r_off[q] * exrec; rcb[q] = r_cnt[q] * exrec;
        }
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send_ex.get(), sob.data(), scb.data(),
                                   (char *) halo_ex.get(), rob.data(), rcb.data(), biggest / 16 * exrec, false, nullptr));
    }

    // subtree sizes of the halo boxes, from their owners
    Buf<int32_t> send_sz, halo_sz;
    if (with_sizes) {
        BT_CHECK(send_sz.alloc(ctx->pool, std::max<int64_t>(nsend, 1)));
        BT_CHECK(halo_sz.alloc(ctx->pool, std::max<int64_t>(nrecv, 1)));
        for (size_t i = 0; i < peers.size(); ++i) {
            const int q = peers[i];
            if (s_cnt[q] == 0) continue;
            pr.q = q;
            let_pack_sizes_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
                n_mine, pr, pos.get() + (int64_t) i * (n_mine + 1), tree->box_subtree_sizes, send_sz.get() + s_off[q]);
        }
        BT_HIP_CHECK(hipGetLastError());
        std::vector<int64_t> sob((size_t) nranks), scb((size_t) nranks), rob((size_t) nranks), rcb((size_t) nranks);
        for (int q = 0; q < nranks; ++q) {
            sob[q] = s_off[q] * 4; scb[q] = s_cnt[q] * 4; rob[q] = r_off[q] * 4; rcb[q] = r_cnt[q] * 4;
        }
        BT_CHECK(comm_all_to_all_v(comm, stream, (const char *) send_sz.get(), sob.data(), scb.data(),
                                   (char *) halo_sz.get(), rob.data(), rcb.data(), biggest / 4, false, nullptr));
    }

    // -- the box set: top levels from the plan, my deep boxes, the halo ----------------------------
    int64_t ntop = 0;
    for (int lev = 0; lev < ntop_levels; ++lev) ntop += pl.nboxes[lev];
    const int64_t nd = n_mine + nrecv;
    const int64_t B = ntop + nd;
    if (B > 0x7fffffff) { set_error("bt_mgpu_let_build: more than 2^31-1 boxes"); return BT_ERR_UNSUPPORTED; }
    BT_ARENA(t_paths, uint64_t, ms, ntop);
    BT_ARENA(t_meta, int32_t, ms, ntop);
    BT_ARENA(t_gid, int32_t, ms, ntop);
    BT_ARENA(t_mine, int8_t, ms, ntop);
    BT_ARENA(t_srccum, int32_t, ms, ext ? ntop : 1);
    std::vector<int32_t> level_starts(1, 0);
    int64_t nt = 0;
    for (int lev = 0; lev < ntop_levels; ++lev) {
        const int64_t n = (int64_t) 1 << (D * lev);
        for (int64_t pth = 0; pth < n; ++pth) {
            if (!pl.exists[lev][pth]) continue;
            const bool internal = pl.split[lev][pth];
            // tree.py:109-145 with sources = targets: children on both sides, or a leaf that is both.
            // With separate targets a box with children STILL carries both child flags, whatever
            // lies below it: tree_build_kernels.py:1254 sets HAS_SOURCE_OR_TARGET_CHILD_BOXES (=
            // both bits, tree.py:138-139) before the per-side tests of :1262-1272 -- the single-GPU
            // build does the same (test_multi_rank_native_entries_disjoint_clouds).
            int32_t flags = internal ? (BT_BOX_HAS_SOURCE_CHILD_BOXES | BT_BOX_HAS_TARGET_CHILD_BOXES)
                                     : (BT_BOX_IS_SOURCE_BOX | BT_BOX_IS_TARGET_BOX);
            if (!internal && pl.sep_targets) {
                // a leaf is a source box iff it holds sources, a target box iff targets (tbk:1258-1262)
                const int64_t ns = pl.src_counts[lev][pth], ntg = pl.counts[lev][pth] - ns;
                flags = (ns > 0 ? BT_BOX_IS_SOURCE_BOX : 0) | (ntg > 0 ? BT_BOX_IS_TARGET_BOX : 0);
            }
            // a box with children whose own (staying) particles are targets is a target box too
            // (tbk:1252-1256; likewise sources, which only stay when the stick-out factor is ~0)
            const int64_t stay_s = (internal && ext) ? pl.stay_src[lev][pth] : 0;
            const bool own_targets = internal && ext && pl.stay[lev][pth] - stay_s > 0;
            if (own_targets) flags |= BT_BOX_IS_TARGET_BOX;
            if (stay_s > 0) flags |= BT_BOX_IS_SOURCE_BOX;
            // lists of the shared internal boxes are built by every rank, those of a top LEAF
            // only by the rank that owns its cells -- and the lists an internal box has as a
            // target box (its own targets', extents only) by the rank that holds those targets
            const int64_t first_cell = pth << (D * (k - lev));
            t_paths[nt] = (uint64_t) pth;
            t_meta[nt] = lev | (flags << 8);
            t_gid[nt] = num->level_start_box_nrs[lev] + pl.index[lev][pth];
            t_mine[nt] = internal ? ((own_targets && pl.owner[first_cell] != rank) ? 2 : 1)
                                  : (pl.owner[first_cell] == rank ? 1 : 0);
            if (ext) t_srccum[nt] = (int32_t) std::min<int64_t>(pl.src_counts[lev][pth], 0x7fffffff);
            ++nt;
        }
        level_starts.push_back((int32_t) nt);
    }
    if (nt != ntop) { set_error("bt_mgpu_let_build: the plan's box counts are inconsistent"); return BT_ERR_INTERNAL; }
    BT_CHECK(ms->let_paths.alloc(ctx->pool, B));
    BT_CHECK(ms->let_meta.alloc(ctx->pool, B));
    BT_CHECK(ms->let_gid.alloc(ctx->pool, B));
    BT_CHECK(ms->let_mask.alloc(ctx->pool, B));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_paths.get(), t_paths, (size_t) ntop * 8, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_meta.get(), t_meta, (size_t) ntop * 4, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_gid.get(), t_gid, (size_t) ntop * 4, hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipMemcpyAsync(ms->let_mask.get(), t_mine, (size_t) ntop, hipMemcpyHostToDevice, stream));

    // -- deep levels: per level the blocks of the ranks in rank order (LetBlocks) ------------------
    std::vector<LetBlocks> blk((size_t) nranks);
    std::vector<int64_t> seq((size_t) nranks, 0);           // position within a sender's sequence
    int64_t cursor = ntop;
    for (int lev = 0; lev < nlev; ++lev) {
        if (lev < ntop_levels) {
            out->active_level_ranges[lev][0] = level_starts[(size_t) lev];
            out->active_level_ranges[lev][1] = level_starts[(size_t) lev + 1];
            continue;
        }
        for (int q = 0; q < nranks; ++q) {
            const int64_t c = q == rank
                ? (lev < nlev_local ? (int64_t) tree->level_start_box_nrs[lev + 1] - tree->level_start_box_nrs[lev] : 0)
                : (int64_t) table[((int64_t) q * nranks + rank) * LW + lev];
            blk[(size_t) q].src_start[lev] = q == rank ? (lev < nlev_local ? tree->level_start_box_nrs[lev] : (int32_t) nb)
                                                       : (int32_t) seq[(size_t) q];
            blk[(size_t) q].dst_start[lev] = (int32_t) cursor;
            if (q == rank) {
                out->active_level_ranges[lev][0] = (int32_t) cursor;
                out->active_level_ranges[lev][1] = (int32_t) (cursor + c);
            }
            seq[(size_t) q] += c;
            cursor += c;
        }
        level_starts.push_back((int32_t) cursor);
    }
    if (cursor != B) {
        set_error("bt_mgpu_let_build: level counts (%lld) do not add up to the box count (%lld)",
                  (long long) cursor, (long long) B);
        return BT_ERR_INTERNAL;
    }
    if (n_mine > 0)
        let_scatter_own_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
            n_mine, b0, blk[(size_t) rank], paths.get(), tree->box_levels, tree->box_flags, box_ids,
            ms->let_paths.get(), ms->let_meta.get(), ms->let_gid.get(), ms->let_mask.get());
    for (int q = 0; q < nranks; ++q) {
        if (q == rank || r_cnt[q] == 0) continue;
        let_scatter_halo_kernel<<<(unsigned) div_up(r_cnt[q], 256), 256, 0, stream>>>(
            r_cnt[q], blk[(size_t) q], halo_rec.get() + 2 * r_off[q], ms->let_paths.get(), ms->let_meta.get(),
            ms->let_gid.get(), ms->let_mask.get(), ctx->d_status);
    }
    BT_HIP_CHECK(hipGetLastError());
    ms->let_has_sizes = with_sizes;
    out->has_subtree_sizes = with_sizes ? 1 : 0;
    if (with_sizes) {
        BT_CHECK(ms->let_sizes.alloc(ctx->pool, std::max<int64_t>(B, 1)));
        if (n_mine > 0)
            let_scatter_own_sizes_kernel<<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(
                n_mine, b0, blk[(size_t) rank], tree->box_levels, tree->box_subtree_sizes, ms->let_sizes.get());
        for (int q = 0; q < nranks; ++q) {
            if (q == rank || r_cnt[q] == 0) continue;
            let_scatter_halo_sizes_kernel<<<(unsigned) div_up(r_cnt[q], 256), 256, 0, stream>>>(
                r_cnt[q], blk[(size_t) q], halo_rec.get() + 2 * r_off[q], halo_sz.get() + r_off[q], ms->let_sizes.get());
        }
        BT_HIP_CHECK(hipGetLastError());
    }
    ms->let_ext = ext;
    if (ext) {
        BT_CHECK(ms->let_tbb.alloc(ctx->pool, 2 * (int64_t) D * B * (int64_t) es));
        BT_CHECK(ms->let_srccum.alloc(ctx->pool, B));
        // the shared top boxes: union of the ranks' versions
        Buf<double> top_mm;
        Buf<int32_t> top_srccum_d;
        BT_CHECK(top_mm.alloc(ctx->pool, ntop * 2 * D));
        BT_CHECK(top_srccum_d.alloc(ctx->pool, ntop));
        fill_f64_kernel<<<(unsigned) div_up(ntop * 2 * D, 256), 256, 0, stream>>>(ntop * 2 * D, 1.7976931348623158e+308,
                                                                               top_mm.get());
        BT_HIP_CHECK(hipMemcpyAsync(top_srccum_d.get(), t_srccum, (size_t) ntop * 4, hipMemcpyHostToDevice, stream));
        const int64_t ntop_local = b0;          // the local tree's boxes of levels <= k come first
#define BT_LET_EXT(T)                                                                                         \
        {                                                                                                     \
            LetExtras<T> ex{(const T *) tree->box_target_bounding_box_min,                                    \
                            (const T *) tree->box_target_bounding_box_max, tree->box_source_counts_cumul,     \
                            tree->aligned_nboxes, D};                                                         \
            if (ntop_local > 0)                                                                               \
                let_top_tbb_gather_kernel<T><<<(unsigned) div_up(ntop_local, 256), 256, 0, stream>>>(        \
                    ntop_local, box_ids, ex, top_mm.get());                                                   \
            BT_CHECK(comm_all_reduce(comm, stream, top_mm.get(), (size_t) (ntop * 2 * D), RED_MIN_F64));       \
            let_top_tbb_place_kernel<T><<<(unsigned) div_up(ntop, 256), 256, 0, stream>>>(                    \
                ntop, D, top_mm.get(), top_srccum_d.get(), B, (T *) ms->let_tbb.get(), ms->let_srccum.get()); \
            if (n_mine > 0)                                                                                   \
                let_scatter_own_extras_kernel<T><<<(unsigned) div_up(n_mine, 256), 256, 0, stream>>>(         \
                    n_mine, b0, blk[(size_t) rank], tree->box_levels, ex, B, (T *) ms->let_tbb.get(),         \
                    ms->let_srccum.get());                                                                    \
            for (int q = 0; q < nranks; ++q) {                                                                \
                if (q == rank || r_cnt[q] == 0) continue;                                                     \
                const unsigned char *blk0 = halo_ex.get() + r_off[q] * exrec;                                 \
                let_scatter_halo_extras_kernel<T><<<(unsigned) div_up(r_cnt[q], 256), 256, 0, stream>>>(      \
                    r_cnt[q], blk[(size_t) q], halo_rec.get() + 2 * r_off[q], (const T *) blk0,              \
                    (const int32_t *) (blk0 + r_cnt[q] * 2 * D * es), D, B, (T *) ms->let_tbb.get(),          \
                    ms->let_srccum.get());                                                                    \
            }                                                                                                 \
        }
        if (es == 8) BT_LET_EXT(double) else BT_LET_EXT(float)
#undef BT_LET_EXT
        BT_HIP_CHECK(hipGetLastError());
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
    BT_CHECK(arena_mark(ctx, ms));
    return bt::finish_call(ctx);
}

// A rank that leaves a collective entry with an error tells the local group, so that its peers
// (threads, or processes of a shared-memory group, waiting at a barrier of the same collective)
// return an error instead of waiting.
static int peer_result(bt_mgpu_comm *comm, int status)
{
    if (status != BT_OK && comm && comm->kind == 1 && comm->group) comm->group->fail();
    if (status != BT_OK && comm && comm->kind == 2 && comm->shm) comm->shm->fail();
    return status;
}

int bt_mgpu_exchange(bt_context *ctx, bt_mgpu_comm *comm, const bt_mgpu_params *p, bt_mgpu_shard *out)
{
    return peer_result(comm, bt_mgpu_exchange_body(ctx, comm, p, out));
}

int bt_mgpu_exchange_time(bt_context *ctx, float *a2a_ms)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !a2a_ms) { set_error("bt_mgpu_exchange_time: invalid argument"); return BT_ERR_INVALID; }
    MgpuState *ms = ctx->mgpu;
    if (!ms || !ms->ev[1]) { set_error("bt_mgpu_exchange_time: no exchange on this context"); return BT_ERR_INVALID; }
    if (ms->a2a_pending) {
        BT_HIP_CHECK(hipEventSynchronize(ms->ev[1]));
        BT_HIP_CHECK(hipEventElapsedTime(&ms->a2a_ms, ms->ev[0], ms->ev[1]));
        ms->a2a_pending = false;
    }
    *a2a_ms = ms->a2a_ms;
    return BT_OK;
}

int bt_mgpu_route(bt_context *ctx, bt_mgpu_comm *comm, int particle_set, int direction, int elem_size,
                  const void *in, void *out)
{
    return peer_result(comm, bt_mgpu_route_body(ctx, comm, particle_set, direction, elem_size, in, out, false));
}

int bt_mgpu_global_ids(bt_context *ctx, bt_mgpu_comm *comm, int particle_set, int id_size, void *ids)
{
    return peer_result(comm, bt_mgpu_route_body(ctx, comm, particle_set, BT_ROUTE_TO_OWNERS, id_size, nullptr, ids, true));
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
    BT_CHECK(reset_status(ctx));
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
    // (every box gets its parent and its centre from the link kernel; only the padding of the
    // centre rows is cleared)
    const size_t cs = ms->let_kind == BT_F64 ? 8 : 4;
    if (aligned > B)
        for (int ax = 0; ax < ms->let_dims; ++ax)
            BT_HIP_CHECK(hipMemsetAsync((char *) o->box_centers + ((size_t) ax * (size_t) aligned + (size_t) B) * cs, 0,
                                        (size_t) (aligned - B) * cs, stream));
    BT_CHECK(bt::let_link_device(ctx, ms->let_dims, ms->let_kind, ms->let_nlevels, ms->let_level_starts.data(),
                                 ms->let_paths.get(), aligned, pl.bbox_min, pl.bbox_max, pl.root_extent,
                                 o->box_parent_ids, o->box_child_ids, o->box_centers));
    if (ms->let_ext) {
        if (!o->box_target_bounding_box_min || !o->box_target_bounding_box_max || !o->box_source_counts_cumul) {
            set_error("bt_mgpu_let_export: the LET has targets with extents; box_target_bounding_box_min / "
                      "_max and box_source_counts_cumul are required");
            return BT_ERR_INVALID;
        }
        if (cs == 8)
            let_tbb_export_kernel<double><<<(unsigned) div_up(aligned, 256), 256, 0, stream>>>(
                B, aligned, ms->let_dims, (const double *) ms->let_tbb.get(), (double *) o->box_target_bounding_box_min,
                (double *) o->box_target_bounding_box_max);
        else
            let_tbb_export_kernel<float><<<(unsigned) div_up(aligned, 256), 256, 0, stream>>>(
                B, aligned, ms->let_dims, (const float *) ms->let_tbb.get(), (float *) o->box_target_bounding_box_min,
                (float *) o->box_target_bounding_box_max);
        BT_HIP_CHECK(hipGetLastError());
        BT_HIP_CHECK(hipMemcpyAsync(o->box_source_counts_cumul, ms->let_srccum.get(), (size_t) B * 4,
                                    hipMemcpyDeviceToDevice, stream));
        ms->let_tbb.reset(); ms->let_srccum.reset();
    }
    if (ms->let_has_sizes && o->box_subtree_sizes) {
        // the shared top levels: a box and what the LET holds below it
        const int ntop_levels = std::min(pl.k + 1, ms->let_nlevels);
        for (int lev = ntop_levels - 1; lev >= 0; --lev) {
            const int32_t b0 = ms->let_level_starts[(size_t) lev], nbl = ms->let_level_starts[(size_t) lev + 1] - b0;
            if (nbl > 0)
                let_top_sizes_kernel<<<(unsigned) div_up(nbl, 256), 256, 0, stream>>>(
                    b0, nbl, 1 << ms->let_dims, aligned, o->box_child_ids, ms->let_sizes.get());
        }
        BT_HIP_CHECK(hipGetLastError());
        BT_HIP_CHECK(hipMemcpyAsync(o->box_subtree_sizes, ms->let_sizes.get(), (size_t) B * 4,
                                    hipMemcpyDeviceToDevice, stream));
    }
    ms->let_sizes.reset();
    ms->let_paths.reset(); ms->let_meta.reset(); ms->let_gid.reset(); ms->let_mask.reset();
    ms->let_nlevels = 0;
    return bt::finish_call(ctx);
}

}  // extern "C"
