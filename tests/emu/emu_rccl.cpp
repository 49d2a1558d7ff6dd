// TEST INFRASTRUCTURE -- the few RCCL entry points boxtree_amd/csrc/bt_mgpu.hip binds, for a world of ONE
// rank on the emulator (tests/emu/README.md): all-reduce and all-gather are copies, a grouped
// ncclSend / ncclRecv to oneself is matched in order at ncclGroupEnd.  With it the tests that drive the
// library's RCCL branch on a one-rank communicator (tests/test_gpu_mgpu.py: root box, ownership,
// grouped point-to-point rounds with the self-loopback switch) run their logic without a GPU.  More
// than one rank is refused: there is no transport here.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {
struct Comm { int nranks, rank; };
struct Msg { void *buf; size_t bytes; };
thread_local int g_depth = 0;
thread_local std::vector<Msg> g_sends, g_recvs;

size_t type_size(int t)
{
    switch (t) {                      // ncclDataType_t
    case 0: case 1: return 1;         // int8, uint8
    case 2: case 3: case 7: return 4; // int32, uint32, float32
    case 4: case 5: case 8: return 8; // int64, uint64, float64
    case 6: case 9: return 2;         // float16, bfloat16
    default: return 0;
    }
}

int flush()
{
    if (g_sends.size() != g_recvs.size()) {
        fprintf(stderr, "emu rccl: %zu sends and %zu receives to oneself in one group\n", g_sends.size(), g_recvs.size());
        g_sends.clear(); g_recvs.clear();
        return 5;                     // ncclInvalidUsage
    }
    int rc = 0;
    for (size_t i = 0; i < g_sends.size(); ++i) {
        if (g_sends[i].bytes != g_recvs[i].bytes) { rc = 5; continue; }
        memmove(g_recvs[i].buf, g_sends[i].buf, g_sends[i].bytes);
    }
    g_sends.clear(); g_recvs.clear();
    return rc;
}
}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) { memset(id, 0x5A, sizeof(*id)); return 0; }

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId, int rank)
{
    if (nranks != 1 || rank != 0) {
        fprintf(stderr, "emu rccl: a world of %d ranks (the emulator has no transport: one rank only)\n", nranks);
        return 5;
    }
    *comm = new Comm{nranks, rank};
    return 0;
}

int ncclCommDestroy(void *comm) { delete (Comm *) comm; return 0; }

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int /*op*/, void *, void *)
{
    if (send != recv) memmove(recv, send, count * type_size(dtype));
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *, void *)
{
    if (send != recv) memmove(recv, send, count * type_size(dtype));
    return 0;
}

int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() { if (g_depth > 0) --g_depth; return g_depth == 0 ? flush() : 0; }

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *, void *)
{
    if (peer != 0) return 5;
    g_sends.push_back({const_cast<void *>(buf), count * type_size(dtype)});
    return g_depth == 0 ? flush() : 0;
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *, void *)
{
    if (peer != 0) return 5;
    g_recvs.push_back({buf, count * type_size(dtype)});
    return g_depth == 0 ? flush() : 0;
}

// (a test copies the receive buffer with the HIP runtime's own hipMemcpy through ctypes: "device"
// memory is host memory here)
int hipMemcpy(void *dst, const void *src, size_t bytes, int /*kind*/) { if (bytes) memmove(dst, src, bytes); return 0; }

const char *ncclGetErrorString(int code) { return code == 0 ? "no error" : "emulated RCCL: invalid usage"; }

}  // extern "C"
