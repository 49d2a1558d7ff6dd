// Ranks as PROCESSES that share one GPU (or none): the three collectives of the sharded build
// (bt_mgpu.hip: all-reduce of a few 8-byte words, all-gather, all-to-all-v) staged through one POSIX
// shared-memory segment.  RCCL refuses two ranks on one device, and the thread ranks of
// bt_mgpu_comm_local share an address space; this is the third communicator kind, so that the code
// bench.py times on N GPUs also runs as N real processes on a box with one (tests, and
// `bench.py --gpus 2` there).  A correctness vehicle, not a transport: every byte crosses the host.
//
// Host-only and free of HIP: the caller hands in how device bytes get to a host buffer and back
// (Mover), so the collectives themselves are unit-tested with memcpy movers by plain processes
// (tests/cabi/shm_group_test.cpp).
//
// Layout of the segment: Header | rank 0's slot | rank 1's slot | ...   A slot is written by its
// owner only, between two barriers; a collective that moves more than a slot holds runs in rounds.
#pragma once

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace bt {

struct ShmHeader {
    std::atomic<uint32_t> magic;       // set last by the creator
    int32_t nranks;
    int64_t slot_bytes;
    std::atomic<int32_t> arrived;
    std::atomic<uint32_t> gen;
    std::atomic<int32_t> failed;
    std::atomic<int32_t> attached, detached;
    char pad[64];
};

constexpr uint32_t SHM_MAGIC = 0xB7C0FFEEu;

struct ShmMover {                      // device <-> host, synchronous at return
    void *user;
    int (*to_host)(void *user, void *host, const void *dev, size_t bytes);
    int (*to_dev)(void *user, void *dev, const void *host, size_t bytes);
};

class ShmGroup {
public:
    // Every rank opens the same name with the same nranks / slot_bytes; the first to arrive creates
    // and initialises the segment, the others wait for its magic word.  timeout_s bounds every wait
    // (a peer that died must not hang the rest).
    static ShmGroup *open(const char *name, int rank, int nranks, int64_t slot_bytes, double timeout_s,
                          std::string *err)
    {
        if (!name || name[0] != '/' || nranks < 1 || rank < 0 || rank >= nranks || slot_bytes < 4096) {
            *err = "invalid argument (the name must start with '/', slots hold at least 4096 bytes)";
            return nullptr;
        }
        slot_bytes = (slot_bytes + 4095) / 4096 * 4096;
        const size_t total = header_bytes() + (size_t) nranks * (size_t) slot_bytes;
        bool creator = true;
        int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 && errno == EEXIST) {
            creator = false;
            fd = shm_open(name, O_RDWR, 0600);
        }
        if (fd < 0) { *err = std::string("shm_open: ") + strerror(errno); return nullptr; }
        if (creator && ftruncate(fd, (off_t) total) != 0) {
            *err = std::string("ftruncate: ") + strerror(errno);
            close(fd); shm_unlink(name);
            return nullptr;
        }
        const double t0 = now();
        if (!creator) {                // the creator may not have sized the segment yet
            struct stat sb;
            while (fstat(fd, &sb) == 0 && (size_t) sb.st_size < total) {
                if (now() - t0 > timeout_s) { *err = "timed out waiting for the segment to be sized"; close(fd); return nullptr; }
                sched_yield();
            }
        }
        void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { *err = std::string("mmap: ") + strerror(errno); return nullptr; }
        ShmHeader *h = (ShmHeader *) p;
        if (creator) {
            h->nranks = nranks;
            h->slot_bytes = slot_bytes;
            h->arrived.store(0); h->gen.store(0); h->failed.store(0);
            h->attached.store(0); h->detached.store(0);
            h->magic.store(SHM_MAGIC, std::memory_order_release);
        } else {
            while (h->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
                if (now() - t0 > timeout_s) { *err = "timed out waiting for the segment's creator"; munmap(p, total); return nullptr; }
                sched_yield();
            }
            if (h->nranks != nranks || h->slot_bytes != slot_bytes) {
                *err = "the segment was created with another rank count or slot size";
                munmap(p, total);
                return nullptr;
            }
        }
        h->attached.fetch_add(1);
        ShmGroup *g = new ShmGroup();
        g->name_ = name; g->h_ = h; g->base_ = (char *) p; g->total_ = total;
        g->rank_ = rank; g->n_ = nranks; g->slot_ = slot_bytes; g->timeout_ = timeout_s;
        return g;
    }

    ~ShmGroup()
    {
        if (!h_) return;
        // the last rank to leave removes the name (a rank that never arrived leaves it behind:
        // callers use a fresh name per job)
        const bool last = h_->detached.fetch_add(1) + 1 == n_;
        munmap(base_, total_);
        if (last) shm_unlink(name_.c_str());
    }

    int rank() const { return rank_; }
    int nranks() const { return n_; }
    int64_t slot_bytes() const { return slot_; }
    char *slot(int r) const { return base_ + header_bytes() + (size_t) r * (size_t) slot_; }
    const char *last_error() const { return err_.c_str(); }

    // false: a peer failed or did not arrive in time (this rank should fail too)
    bool barrier()
    {
        if (h_->failed.load()) { err_ = "a peer rank of the shared-memory group failed"; return false; }
        const uint32_t g = h_->gen.load(std::memory_order_acquire);
        if (h_->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            h_->arrived.store(0, std::memory_order_relaxed);
            h_->gen.fetch_add(1, std::memory_order_release);
            return true;
        }
        const double t0 = now();
        unsigned spins = 0;
        while (h_->gen.load(std::memory_order_acquire) == g) {
            if (h_->failed.load()) { err_ = "a peer rank of the shared-memory group failed"; return false; }
            if ((++spins & 1023u) == 0 && now() - t0 > timeout_) {
                err_ = "timed out in a barrier of the shared-memory group";
                h_->failed.store(1);
                return false;
            }
            sched_yield();
        }
        return true;
    }
    void fail() { h_->failed.store(1); }

    // in-place all-reduce of `count` 8-byte words (sum of int64, or min of double)
    bool all_reduce(const ShmMover &mv, void *dev, size_t count, bool min_f64)
    {
        if ((int64_t) (count * 8) > slot_) { err_ = "all-reduce larger than a slot"; fail(); return false; }
        if (mv.to_host(mv.user, slot(rank_), dev, count * 8) != 0) { err_ = "copy to host failed"; fail(); return false; }
        if (!barrier()) return false;
        // (results are put together in the rank's own memory: the slots stay as published until
        // every rank has read them)
        std::string res(count * 8, '\0');
        for (size_t i = 0; i < count; ++i) {
            if (!min_f64) {
                int64_t s = 0;
                for (int q = 0; q < n_; ++q) { int64_t v; memcpy(&v, slot(q) + i * 8, 8); s += v; }
                memcpy(&res[i * 8], &s, 8);
            } else {
                double s; memcpy(&s, slot(0) + i * 8, 8);
                for (int q = 1; q < n_; ++q) { double v; memcpy(&v, slot(q) + i * 8, 8); s = v < s ? v : s; }
                memcpy(&res[i * 8], &s, 8);
            }
        }
        if (!barrier()) return false;
        if (mv.to_dev(mv.user, dev, res.data(), count * 8) != 0) { err_ = "copy to device failed"; fail(); return false; }
        return true;
    }

    // recv[q * bytes ...] = rank q's send[0 .. bytes)
    bool all_gather(const ShmMover &mv, const void *send, void *recv, size_t bytes)
    {
        for (size_t done = 0; done < bytes; done += (size_t) slot_) {
            const size_t piece = bytes - done < (size_t) slot_ ? bytes - done : (size_t) slot_;
            if (mv.to_host(mv.user, slot(rank_), (const char *) send + done, piece) != 0) {
                err_ = "copy to host failed"; fail(); return false;
            }
            if (!barrier()) return false;
            for (int q = 0; q < n_; ++q)
                if (mv.to_dev(mv.user, (char *) recv + (size_t) q * bytes + done, slot(q), piece) != 0) {
                    err_ = "copy to device failed"; fail(); return false;
                }
            if (!barrier()) return false;
        }
        return true;
    }

    // all-to-all-v in bytes; offsets and counts are host arrays [nranks] (the rank's own segment
    // included unless skip_self).  Every rank's slot is cut into nranks equal mailboxes, one per
    // destination; a round moves one mailbox-full of every message.  *rounds_out: rounds run.
    bool all_to_all_v(const ShmMover &mv, const char *send, const int64_t *s_off, const int64_t *s_cnt,
                      char *recv, const int64_t *r_off, const int64_t *r_cnt, bool skip_self, int *rounds_out)
    {
        // the counts first: every receiver checks what it is about to be sent, and every rank
        // learns the largest message (the number of rounds must be the same everywhere)
        const int64_t box = ((slot_ - (int64_t) n_ * 8) / n_) & ~(int64_t) 15;
        if (box < 16) { err_ = "slots too small for this many ranks"; fail(); return false; }
        memcpy(slot(rank_), s_cnt, (size_t) n_ * 8);
        if (!barrier()) return false;
        int64_t biggest = 0;
        bool ok = true;
        for (int q = 0; q < n_; ++q) {
            const int64_t *row = (const int64_t *) slot(q);
            for (int d = 0; d < n_; ++d)
                if (!(skip_self && q == d) && row[d] > biggest) biggest = row[d];
            if (!(skip_self && q == rank_) && row[rank_] != r_cnt[q]) {
                char msg[160];
                snprintf(msg, sizeof msg, "all-to-all-v: rank %d sends %lld bytes to rank %d, which expects %lld",
                         q, (long long) row[rank_], rank_, (long long) r_cnt[q]);
                err_ = msg;
                ok = false;
            }
        }
        if (!ok) { fail(); return false; }
        if (!barrier()) return false;
        const int64_t rounds = biggest > 0 ? (biggest + box - 1) / box : 0;
        if (rounds_out) *rounds_out = (int) rounds;
        char *mail = slot(rank_) + (size_t) n_ * 8;
        for (int64_t j = 0; j < rounds; ++j) {
            for (int d = 0; d < n_; ++d) {
                if (skip_self && d == rank_) continue;
                const int64_t lo = j * box, hi = (j + 1) * box < s_cnt[d] ? (j + 1) * box : s_cnt[d];
                if (hi > lo && mv.to_host(mv.user, mail + (size_t) d * (size_t) box, send + s_off[d] + lo,
                                          (size_t) (hi - lo)) != 0) {
                    err_ = "copy to host failed"; fail(); return false;
                }
            }
            if (!barrier()) return false;
            for (int q = 0; q < n_; ++q) {
                if (skip_self && q == rank_) continue;
                const int64_t lo = j * box, hi = (j + 1) * box < r_cnt[q] ? (j + 1) * box : r_cnt[q];
                const char *from = slot(q) + (size_t) n_ * 8 + (size_t) rank_ * (size_t) box;
                if (hi > lo && mv.to_dev(mv.user, recv + r_off[q] + lo, from, (size_t) (hi - lo)) != 0) {
                    err_ = "copy to device failed"; fail(); return false;
                }
            }
            if (!barrier()) return false;
        }
        return true;
    }

private:
    ShmGroup() = default;
    static size_t header_bytes() { return (sizeof(ShmHeader) + 4095) / 4096 * 4096; }
    static double now()
    {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
    }
    std::string name_, err_;
    ShmHeader *h_ = nullptr;
    char *base_ = nullptr;
    size_t total_ = 0;
    int rank_ = 0, n_ = 1;
    int64_t slot_ = 0;
    double timeout_ = 60.0;
};

}  // namespace bt
