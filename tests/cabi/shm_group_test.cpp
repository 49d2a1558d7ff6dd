// CPU unit test of boxtree_amd/csrc/bt_shm_group.hpp: N forked processes run the three
// collectives of the sharded build over a shared-memory segment, with memcpy standing in for the
// device copies (the header is free of HIP).  Slots are made small on purpose so that every
// collective runs in several rounds.  Exit code 0 = all ranks passed.
//
//   shm_group_test [nranks=3] [slot_bytes=8192]
#include <cstdlib>
#include <random>
#include <vector>

#include <sys/wait.h>

#include "../../boxtree_amd/csrc/bt_shm_group.hpp"

static int cp(void *, void *dst, const void *src, size_t n) { memcpy(dst, src, n); return 0; }

#define REQUIRE(cond)                                                                         \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            fprintf(stderr, "rank %d: %s:%d: %s failed (%s)\n", rank, __FILE__, __LINE__, #cond, \
                    g ? g->last_error() : "");                                                \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// what rank q sends to rank d in test `t`: a size and a byte pattern both sides can compute
static int64_t msg_bytes(int t, int q, int d, int n)
{
    std::mt19937_64 r((uint64_t) (1000 * t + 31 * q + d));
    const int kind = (int) (r() % 4);
    if (kind == 0) return 0;                                   // empty messages are legal
    if (kind == 1) return (int64_t) (r() % 64) + 1;
    return (int64_t) (r() % 40000) + 1;                        // several rounds with 8-kB slots
}
static unsigned char msg_byte(int t, int q, int d, int64_t i) { return (unsigned char) (t * 7 + q * 31 + d * 13 + i * 3 + (i >> 8)); }

static int run_rank(const char *name, int rank, int n, int64_t slot)
{
    std::string err;
    bt::ShmGroup *g = bt::ShmGroup::open(name, rank, n, slot, 20.0, &err);
    if (!g) { fprintf(stderr, "rank %d: open: %s\n", rank, err.c_str()); return 1; }
    bt::ShmMover mv{nullptr, cp, cp};

    // all-reduce: sums of int64, minima of doubles
    {
        std::vector<int64_t> v(5);
        for (int i = 0; i < 5; ++i) v[i] = (int64_t) (rank + 1) * (i + 1) - (i == 4 ? ((int64_t) 1 << 62) : 0);
        REQUIRE(g->all_reduce(mv, v.data(), v.size(), false));
        for (int i = 0; i < 5; ++i) {
            int64_t want = 0;
            for (int q = 0; q < n; ++q) want += (int64_t) ((uint64_t) ((int64_t) (q + 1) * (i + 1)) - (i == 4 ? ((uint64_t) 1 << 62) : 0));
            REQUIRE(v[i] == want);
        }
        std::vector<double> m = {1.5 * rank, -2.0 - rank, 1e300, (double) ((rank * 7) % n)};
        REQUIRE(g->all_reduce(mv, m.data(), m.size(), true));
        REQUIRE(m[0] == 0.0 && m[1] == -2.0 - (n - 1) && m[2] == 1e300 && m[3] == 0.0);
    }
    // all-gather, one round and many
    for (size_t bytes : {(size_t) 24, (size_t) (3 * slot + 100)}) {
        std::vector<unsigned char> send(bytes), recv(bytes * n, 0xEE);
        for (size_t i = 0; i < bytes; ++i) send[i] = msg_byte(1, rank, 0, (int64_t) i);
        REQUIRE(g->all_gather(mv, send.data(), recv.data(), bytes));
        for (int q = 0; q < n; ++q)
            for (size_t i = 0; i < bytes; ++i) REQUIRE(recv[q * bytes + i] == msg_byte(1, q, 0, (int64_t) i));
    }
    // all-to-all-v: random sizes (zero included), with and without the rank's own segment
    for (int t = 0; t < 6; ++t) {
        const bool skip_self = t & 1;
        std::vector<int64_t> s_off(n), s_cnt(n), r_off(n), r_cnt(n);
        int64_t st = 0, rt = 0;
        for (int d = 0; d < n; ++d) {
            s_cnt[d] = msg_bytes(t, rank, d, n); s_off[d] = st; st += s_cnt[d] + 5;      // (gaps between segments)
            r_cnt[d] = msg_bytes(t, d, rank, n); r_off[d] = rt; rt += r_cnt[d] + 3;
        }
        std::vector<unsigned char> send((size_t) st + 1, 0xAA), recv((size_t) rt + 1, 0xBB);
        for (int d = 0; d < n; ++d)
            for (int64_t i = 0; i < s_cnt[d]; ++i) send[(size_t) (s_off[d] + i)] = msg_byte(t, rank, d, i);
        int rounds = -1;
        REQUIRE(g->all_to_all_v(mv, (const char *) send.data(), s_off.data(), s_cnt.data(), (char *) recv.data(),
                                r_off.data(), r_cnt.data(), skip_self, &rounds));
        REQUIRE(rounds >= 0);
        for (int q = 0; q < n; ++q) {
            for (int64_t i = 0; i < r_cnt[q]; ++i) {
                const unsigned char got = recv[(size_t) (r_off[q] + i)];
                if (skip_self && q == rank) REQUIRE(got == 0xBB);            // untouched
                else REQUIRE(got == msg_byte(t, q, rank, i));
            }
            for (int k = 0; k < 3 && r_off[q] + r_cnt[q] + k < rt; ++k) REQUIRE(recv[(size_t) (r_off[q] + r_cnt[q] + k)] == 0xBB);
        }
    }
    // a receiver that expects another size fails everybody, nobody hangs
    {
        std::vector<int64_t> off(n, 0), s_cnt(n, 8), r_cnt(n, 8);
        if (rank == n - 1) r_cnt[0] = 16;
        std::vector<char> send(8, 1), recv(64, 0);
        const bool ok = g->all_to_all_v(mv, send.data(), off.data(), s_cnt.data(), recv.data(), off.data(), r_cnt.data(),
                                        false, nullptr);
        if (n > 1) REQUIRE(!ok);
        // and the group stays failed
        if (n > 1) REQUIRE(!g->barrier());
    }
    delete g;
    return 0;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 3;
    const int64_t slot = argc > 2 ? atoll(argv[2]) : 8192;
    char name[64];
    snprintf(name, sizeof name, "/bt_shm_test_%d", (int) getpid());
    std::vector<pid_t> kids;
    for (int r = 0; r < n; ++r) {
        const pid_t p = fork();
        if (p == 0) _exit(run_rank(name, r, n, slot));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) {
        int st = 0;
        waitpid(p, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) ++bad;
    }
    shm_unlink(name);           // (the last rank removed it; a failed run may not have)
    if (bad) { fprintf(stderr, "%d of %d ranks failed\n", bad, n); return 1; }
    printf("shm_group_test: %d ranks ok (slots of %lld bytes)\n", n, (long long) slot);
    return 0;
}
