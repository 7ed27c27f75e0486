// tests/_fakerccl/fakerccl.cpp — TEST-ONLY stand-in for librccl.so, loaded through the library's $TLAMC_RCCL hook
// (tla_rust_amd/csrc/shard_rccl.cpp).  RCCL refuses two ranks on one GPU, and the GPU boxes this repository is developed
// on have ONE: this file implements the nine nccl* entry points the hip-rccl back-end uses — between PROCESSES that may all
// sit on the same device — by staging through POSIX shared memory, so that mc_comm_* / mc_shard_run / mc_shard_trace, `mc
// X.tla -gpus P` and `bench.py --gpus N` execute with P = 2 / 4 / 8 exactly as they would over xGMI (same calls, same
// order, same sizes; a size mismatch between a send and its receive is an error here, not a hang).  It is not a
// transport of the product and measures nothing: every collective synchronises its stream and copies through the host.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {
constexpr int MAXR = 8;
constexpr size_t SEG = 4u << 20;  // bytes per (source, destination) pair and pass
struct Shared {
    std::atomic<int> ready;
    std::atomic<int> failed;
    pthread_barrier_t bar;
    uint64_t send_bytes[MAXR][MAXR];
};
struct Op { bool send; char *buf; size_t bytes; int peer; };
struct Comm {
    int rank, n;
    Shared *sh;
    char *box;  // n * n segments: box + (src * n + dst) * SEG
    size_t map_bytes;
    char name[64];
    std::vector<Op> ops;
    bool grouped = false;
    hipStream_t stream = nullptr;
};
thread_local Comm *g_group = nullptr;  // communicator of the open group (the back-end uses one per process)
thread_local int g_depth = 0;

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
void barrier(Comm *c) { pthread_barrier_wait(&c->sh->bar); }

// the exchange every collective reduces to: sends and receives of one rank, matched against the peers' by size
ncclResult_t exchange(Comm *c) {
    if (c->stream && hipStreamSynchronize(c->stream) != hipSuccess) c->sh->failed = 1;
    for (int p = 0; p < c->n; ++p) c->sh->send_bytes[c->rank][p] = 0;
    for (const Op &o : c->ops)
        if (o.send) c->sh->send_bytes[c->rank][o.peer] += o.bytes;
    barrier(c);
    size_t longest = 0;
    for (int s = 0; s < c->n; ++s)
        for (int d = 0; d < c->n; ++d) longest = std::max<size_t>(longest, c->sh->send_bytes[s][d]);
    for (const Op &o : c->ops)  // what I expect from a peer must be what it sends me
        if (!o.send) {
            size_t want = 0;
            for (const Op &q : c->ops) if (!q.send && q.peer == o.peer) want += q.bytes;
            if (want != c->sh->send_bytes[o.peer][c->rank]) c->sh->failed = 1;
        }
    std::vector<size_t> sent(c->n, 0), got(c->n, 0);  // bytes of the pair's stream already moved
    for (size_t off = 0; off < longest; off += SEG) {
        for (int p = 0; p < c->n; ++p) {  // my data for p, bytes [off, off + SEG) of the concatenation of my sends to p
            size_t pos = 0, filled = 0;
            char *seg = c->box + ((size_t)c->rank * c->n + p) * SEG;
            for (const Op &o : c->ops) {
                if (!o.send || o.peer != p) continue;
                const size_t lo = std::max(pos, off), hi = std::min(pos + o.bytes, off + SEG);
                if (hi > lo) {
                    if (hipMemcpy(seg + (lo - off), o.buf + (lo - pos), hi - lo, hipMemcpyDeviceToHost) != hipSuccess) c->sh->failed = 1;
                    filled += hi - lo;
                }
                pos += o.bytes;
            }
            (void)filled;
        }
        barrier(c);
        for (int p = 0; p < c->n; ++p) {
            size_t pos = 0;
            const char *seg = c->box + ((size_t)p * c->n + c->rank) * SEG;
            for (const Op &o : c->ops) {
                if (o.send || o.peer != p) continue;
                const size_t lo = std::max(pos, off), hi = std::min(pos + o.bytes, off + SEG);
                if (hi > lo && hipMemcpy(o.buf + (lo - pos), seg + (lo - off), hi - lo, hipMemcpyHostToDevice) != hipSuccess) c->sh->failed = 1;
                pos += o.bytes;
            }
        }
        barrier(c);
    }
    c->ops.clear();
    barrier(c);
    return c->sh->failed ? ncclInternalError : ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/fakerccl_%d_%ld_%ld", (int)getpid(), (long)t.tv_sec, (long)t.tv_nsec);
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank;
    c->n = nranks;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = ((sizeof(Shared) + 4095) & ~(size_t)4095) + (size_t)nranks * nranks * SEG;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared *)m;
    c->box = (char *)m + ((sizeof(Shared) + 4095) & ~(size_t)4095);
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->bar, &a, (unsigned)nranks);
        c->sh->failed = 0;
        c->sh->ready = 1;
    } else {
        for (int spin = 0; !c->sh->ready; ++spin) {
            if (spin > 60000) { munmap(m, c->map_bytes); delete c; return ncclSystemError; }
            usleep(1000);
        }
    }
    barrier(c);
    if (rank == 0) shm_unlink(c->name);  // every rank has it mapped: the name can go
    *out = (ncclComm_t)c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    munmap((void *)c->sh, c->map_bytes);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    Comm *c = g_group;
    g_group = nullptr;
    return c ? exchange(c) : ncclSuccess;
}
static ncclResult_t queue(Comm *c, bool send, void *buf, size_t bytes, int peer, hipStream_t s) {
    if (!c || peer < 0 || peer >= c->n) return ncclInvalidArgument;
    c->stream = s;
    c->ops.push_back(Op{send, (char *)buf, bytes, peer});
    if (g_depth) { g_group = c; return ncclSuccess; }
    return exchange(c);
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return queue((Comm *)comm, true, (void *)buf, count * dtype_size(t), peer, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    return queue((Comm *)comm, false, buf, count * dtype_size(t), peer, s);
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    if (!c || g_depth) return ncclInvalidUsage;
    const size_t bytes = count * dtype_size(t);
    c->stream = s;
    for (int p = 0; p < c->n; ++p) {
        c->ops.push_back(Op{true, (char *)send, bytes, p});
        c->ops.push_back(Op{false, (char *)recv + (size_t)p * bytes, bytes, p});
    }
    return exchange(c);
}
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fakerccl: exchange failed (HIP copy or a send / receive size mismatch)"; }
}
