// Multi-GPU entry of the C ABI: the particle exchange of a sharded tree build over an
// RCCL communicator the caller owns (one process per GPU).  The
// reference builds its tree on one rank and broadcasts it
// (boxtree/distributed/__init__.py:183-199); nothing here has a counterpart upstream.
//
//   1. bounding box: local min/max, ncclAllReduce(min) over (min, -max)
//      -> the same root box on every rank (host arithmetic of tree_build.py:462-476);
//   2. level-k Morton-cell histogram, ncclAllReduce(sum) -> every rank derives the same
//      top of the global tree and the same owner of every cell (bt_mgpu_plan);
//   3. stable partition by owner that carries the coordinates (interleaved), one grouped
//      ncclSend/ncclRecv round per 512 MiB of the largest peer message (a rank's own
//      segment is a device copy): the all-to-all-v over the point-to-point xGMI links;
//   4. the caller builds its subtrees with bt_tree_build on the returned shard
//      (sources = points + axis, source_stride = dims, bbox_*, top_level,
//      top_cell_prefix).
//
// RCCL is bound at run time (dlopen of the librccl.so.1 already in the process, or the
// system one), so the library itself has no link-time dependency on it.
#include "bt_common.hpp"
#include "bt_prims.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
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

Nccl &nccl()
{
    static Nccl n = [] {
        Nccl r;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
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

constexpr int64_t MESSAGE_LIMIT_BYTES = (int64_t) 512 << 20;   // see DESIGN.md (RCCL, > 1 GB)

__global__ __launch_bounds__(256) void widen_hist_kernel(int64_t n, const int32_t *in, int64_t *out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[i];
}

}  // namespace

struct MgpuState {
    Buf<unsigned char> points;       // received particles, interleaved [n_owned][dims]
    Buf<int64_t> cell_prefix;        // [C^top_level + 1]
};

void bt_free_mgpu_state(bt_context *ctx)
{
    bt::CallScope bt_call_scope_(ctx);
    delete ctx->mgpu;
    ctx->mgpu = nullptr;
}

extern "C" {

// Host part, a pure function of the all-reduced histogram (identical on every rank):
// owner rank of every level-`top_level` cell, and the exclusive prefix sums of the
// histogram.  kind "adaptive", point particles, unit weights: a box of the global top
// tree splits iff it holds more than max_particles_in_box particles
// (tree_build_kernels.py:577-591); all cells below a leaf of that top tree go to one
// rank, so no global leaf straddles ranks.  max_particles_in_box <= 0: cells are
// assigned individually (no top-tree plan).
int bt_mgpu_plan(int dims, int top_level, int64_t max_particles_in_box, int nranks,
                 const int64_t *global_hist, int32_t *owner_of_cell, int64_t *cell_prefix)
{
    if (dims < 1 || dims > 3 || top_level < 1 || dims * top_level > 30 || nranks < 1
            || !global_hist || !owner_of_cell) {
        set_error("bt_mgpu_plan: invalid argument");
        return BT_ERR_INVALID;
    }
    const int C = 1 << dims, k = top_level;
    const int64_t ncells = (int64_t) 1 << (dims * k);
    // counts per level, paths in Morton order
    std::vector<std::vector<int64_t>> counts((size_t) k + 1);
    counts[k].assign(global_hist, global_hist + ncells);
    for (int lev = k - 1; lev >= 0; --lev) {
        const int64_t n = (int64_t) 1 << (dims * lev);
        counts[lev].assign((size_t) n, 0);
        for (int64_t i = 0; i < n; ++i)
            for (int m = 0; m < C; ++m) counts[lev][i] += counts[lev + 1][i * C + m];
    }
    // frontier: the first cell of the top-tree leaf a cell lies in
    std::vector<int64_t> unit_start((size_t) ncells);
    if (max_particles_in_box > 0) {
        std::vector<std::vector<char>> split((size_t) k + 1);
        std::vector<char> exists(1, 1);
        for (int lev = 0; lev <= k; ++lev) {
            const int64_t n = (int64_t) 1 << (dims * lev);
            split[lev].assign((size_t) n, 0);
            for (int64_t i = 0; i < n; ++i)
                split[lev][i] = exists[i] && counts[lev][i] > max_particles_in_box;
            if (lev < k) {
                std::vector<char> next((size_t) n * C, 0);
                for (int64_t i = 0; i < n * C; ++i)
                    next[i] = split[lev][i / C] && counts[lev + 1][i] > 0;
                exists.swap(next);
            }
        }
        for (int64_t c = 0; c < ncells; ++c) {
            int leaf_level = k;
            for (int lev = k - 1; lev >= 0; --lev)
                if (!split[lev][c >> (dims * (k - lev))]) leaf_level = lev;
            const int sh = dims * (k - leaf_level);
            unit_start[c] = (c >> sh) << sh;
        }
    } else {
        for (int64_t c = 0; c < ncells; ++c) unit_start[c] = c;
    }
    // contiguous Morton ranges balanced by particle count: a cell goes to the rank whose
    // ideal range contains its first particle
    const int64_t total = counts[0][0];
    std::vector<int64_t> cum((size_t) ncells + 1, 0);
    for (int64_t c = 0; c < ncells; ++c) cum[c + 1] = cum[c] + global_hist[c];
    for (int64_t c = 0; c < ncells; ++c) {
        const int64_t u = unit_start[c];
        const int64_t o = (int64_t) (((__int128) cum[u] * nranks) / std::max<int64_t>(total, 1));
        owner_of_cell[c] = (int32_t) std::min<int64_t>(o, nranks - 1);
    }
    if (cell_prefix)
        for (int64_t c = 0; c <= ncells; ++c) cell_prefix[c] = cum[c];
    return BT_OK;
}

int bt_mgpu_exchange(bt_context *ctx, void *rccl_comm, int rank, int nranks,
                     const bt_mgpu_params *p, bt_mgpu_shard *out)
{
    bt::CallScope bt_call_scope_(ctx);
    if (!ctx || !rccl_comm || !p || !out || rank < 0 || rank >= nranks) {
        set_error("bt_mgpu_exchange: invalid argument");
        return BT_ERR_INVALID;
    }
    if (p->dims < 1 || p->dims > 3 || (p->coord_kind != BT_F32 && p->coord_kind != BT_F64) || p->n < 0) {
        set_error("bt_mgpu_exchange: bad dims / coord_kind / n");
        return BT_ERR_INVALID;
    }
    if (nranks > BT_MGPU_MAX_RANKS) {
        set_error("bt_mgpu_exchange: %d ranks; the one-sweep partition supports at most %d owners",
                  nranks, BT_MGPU_MAX_RANKS);
        return BT_ERR_UNSUPPORTED;
    }
    if (!nccl().ok) {
        set_error("bt_mgpu_exchange: librccl.so could not be loaded");
        return BT_ERR_UNSUPPORTED;
    }
    BT_HIP_CHECK(hipSetDevice(ctx->device));
    memset(out, 0, sizeof(*out));
    const int D = p->dims;
    const bool f64 = p->coord_kind == BT_F64;
    const int es = f64 ? 8 : 4;
    const int64_t n = p->n;
    Nccl &nc = nccl();
    nccl_comm_t comm = (nccl_comm_t) rccl_comm;
    hipStream_t stream = ctx->stream;

    // ---- 1. global bounding box -> root box --------------------------------------------
    double lmin[3], lmax[3];
    BT_CHECK(bt_bbox(ctx, D, p->coord_kind, p->coords, nullptr, n, lmin, lmax));
    Buf<double> mm;
    BT_CHECK(mm.alloc(ctx->pool, 2 * D));
    double h_mm[6];
    for (int ax = 0; ax < D; ++ax) { h_mm[ax] = lmin[ax]; h_mm[D + ax] = -lmax[ax]; }
    BT_HIP_CHECK(hipMemcpyAsync(mm.get(), h_mm, sizeof(double) * 2 * D, hipMemcpyHostToDevice, stream));
    BT_NCCL_CHECK(nc.AllReduce(mm.get(), mm.get(), 2 * D, NCCL_FLOAT64, NCCL_MIN, comm, stream));
    BT_HIP_CHECK(hipMemcpyAsync(h_mm, mm.get(), sizeof(double) * 2 * D, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
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

    // ---- 2. cell histogram, all-reduced ----------------------------------------------------
    const int k = p->top_level > 0 ? p->top_level : (D == 3 ? 5 : D == 2 ? 7 : 12);
    const int64_t ncells = (int64_t) 1 << (D * k);
    Buf<uint32_t> cells;
    Buf<int32_t> hist32, owner_d;
    Buf<int64_t> hist64;
    BT_CHECK(cells.alloc(ctx->pool, n));
    BT_CHECK(hist32.alloc(ctx->pool, ncells));
    BT_CHECK(hist64.alloc(ctx->pool, ncells));
    BT_CHECK(owner_d.alloc(ctx->pool, ncells));
    BT_HIP_CHECK(hipMemsetAsync(hist32.get(), 0, (size_t) ncells * 4, stream));
    BT_CHECK(bt_morton_cells(ctx, D, p->coord_kind, p->coords, n, bmin, bmax, k, cells.get(), hist32.get()));
    widen_hist_kernel<<<(unsigned) div_up(ncells, 256), 256, 0, stream>>>(ncells, hist32.get(), hist64.get());
    std::vector<int32_t> h_local((size_t) ncells);
    BT_HIP_CHECK(hipMemcpyAsync(h_local.data(), hist32.get(), (size_t) ncells * 4, hipMemcpyDeviceToHost, stream));
    BT_NCCL_CHECK(nc.AllReduce(hist64.get(), hist64.get(), (size_t) ncells, NCCL_INT64, NCCL_SUM, comm, stream));
    std::vector<int64_t> ghist((size_t) ncells);
    BT_HIP_CHECK(hipMemcpyAsync(ghist.data(), hist64.get(), (size_t) ncells * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));

    // ---- 3. plan (host, identical on all ranks), bucketing, counts ---------------------------
    std::vector<int32_t> owner((size_t) ncells);
    std::vector<int64_t> prefix((size_t) ncells + 1);
    BT_CHECK(bt_mgpu_plan(D, k, p->max_particles_in_box, nranks, ghist.data(), owner.data(), prefix.data()));
    BT_HIP_CHECK(hipMemcpyAsync(owner_d.get(), owner.data(), (size_t) ncells * 4, hipMemcpyHostToDevice, stream));
    std::vector<int64_t> send_counts((size_t) nranks, 0);
    for (int64_t c = 0; c < ncells; ++c) send_counts[owner[c]] += h_local[c];
    Buf<int64_t> counts_d;
    BT_CHECK(counts_d.alloc(ctx->pool, (int64_t) nranks * (nranks + 1)));
    BT_HIP_CHECK(hipMemcpyAsync(counts_d.get(), send_counts.data(), (size_t) nranks * 8, hipMemcpyHostToDevice, stream));
    BT_NCCL_CHECK(nc.AllGather(counts_d.get(), counts_d.get() + nranks, (size_t) nranks, NCCL_INT64, comm, stream));
    std::vector<int64_t> matrix((size_t) nranks * nranks);      // [sender][receiver]
    BT_HIP_CHECK(hipMemcpyAsync(matrix.data(), counts_d.get() + nranks, matrix.size() * 8, hipMemcpyDeviceToHost, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));
    std::vector<int64_t> recv_counts((size_t) nranks), s_off((size_t) nranks + 1, 0), r_off((size_t) nranks + 1, 0);
    int64_t biggest = 0;
    for (int r = 0; r < nranks; ++r) {
        recv_counts[r] = matrix[(size_t) r * nranks + rank];
        s_off[r + 1] = s_off[r] + send_counts[r];
        r_off[r + 1] = r_off[r] + recv_counts[r];
        for (int q = 0; q < nranks; ++q)
            if (q != r) biggest = std::max(biggest, matrix[(size_t) r * nranks + q]);
    }
    const int64_t nrecv = r_off[nranks];

    // ---- 4. payload: interleaved coordinates, grouped point-to-point rounds -------------------
    MgpuState *ms = ctx->mgpu;
    if (!ms) { ms = new MgpuState(); ctx->mgpu = ms; }
    Buf<unsigned char> send;
    BT_CHECK(send.alloc(ctx->pool, n * D * es));
    BT_CHECK(ms->points.alloc(ctx->pool, std::max<int64_t>(nrecv, 1) * D * es));
    // one sweep over the coordinates: stable partition by owner into the send buffer, the
    // segment this rank keeps straight into the receive buffer (bt_shard.hip)
    BT_CHECK(bt_partition_pack(ctx, D, es, p->coords, cells.get(), n, owner_d.get(), nranks, rank,
                               s_off[rank], r_off[rank], send.get(), ms->points.get()));
    const int64_t rec = (int64_t) D * es;                       // bytes per particle
    const int64_t rounds = std::max<int64_t>(1, div_up(biggest * rec, MESSAGE_LIMIT_BYTES));
    auto cut = [&](int64_t c, int64_t j) { return (j * c) / rounds; };
    if (biggest > 0) {
        for (int64_t j = 0; j < rounds; ++j) {
            BT_NCCL_CHECK(nc.GroupStart());
            // (a failure inside the group still closes it: RCCL keeps an open group per thread)
            int group_status = BT_OK;
            auto in_group = [&]() -> int {
                for (int peer = 0; peer < nranks; ++peer) {
                    if (peer == rank) continue;
                    const int64_t s0 = cut(send_counts[peer], j), s1 = cut(send_counts[peer], j + 1);
                    const int64_t r0 = cut(recv_counts[peer], j), r1 = cut(recv_counts[peer], j + 1);
                    if (s1 > s0)
                        BT_NCCL_CHECK(nc.Send(send.get() + (s_off[peer] + s0) * rec, (size_t) ((s1 - s0) * rec),
                                              NCCL_UINT8, peer, comm, stream));
                    if (r1 > r0)
                        BT_NCCL_CHECK(nc.Recv(ms->points.get() + (r_off[peer] + r0) * rec,
                                              (size_t) ((r1 - r0) * rec), NCCL_UINT8, peer, comm, stream));
                }
                return BT_OK;
            };
            group_status = in_group();
            if (group_status != BT_OK) { (void) nc.GroupEnd(); return group_status; }
            BT_NCCL_CHECK(nc.GroupEnd());
        }
    }
    BT_CHECK(ms->cell_prefix.alloc(ctx->pool, ncells + 1));
    BT_HIP_CHECK(hipMemcpyAsync(ms->cell_prefix.get(), prefix.data(), (size_t) (ncells + 1) * 8,
                                hipMemcpyHostToDevice, stream));
    BT_HIP_CHECK(hipStreamSynchronize(stream));     // host vectors above go out of scope

    out->n_owned = nrecv;
    out->points = ms->points.get();
    for (int ax = 0; ax < D; ++ax) { out->bbox_min[ax] = bmin[ax]; out->bbox_max[ax] = bmax[ax]; }
    out->root_extent = root_extent;
    out->top_level = k;
    out->top_cell_prefix = p->max_particles_in_box > 0 ? ms->cell_prefix.get() : nullptr;
    out->bytes_sent = (n - send_counts[rank]) * rec;
    out->rounds = (int32_t) rounds;
    return BT_OK;
}

}  // extern "C"
