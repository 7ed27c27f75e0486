// shard_rccl.cpp — the hip-rccl back-end behind the C ABI: the level loop of the fingerprint-sharded
// search in C++, its collectives issued with RCCL (ncclSend / ncclRecv groups = all-to-all, ncclAllGather, ncclAllReduce over
// xGMI) on one HIP stream per rank — the stream the engine's step kernels are ordered on (mc_shard_set_stream).  One process
// per GPU; a host in any language creates the communicator from 128 bytes it ships between its ranks itself
// (mc_comm_unique_id on rank 0, mc_comm_create everywhere) and calls mc_shard_run; `mc X.tla -gpus P` does exactly that with
// forked ranks and a file (mc_main.cpp).  tla_rust_amd/sharded.py is the same loop over torch.distributed, which the CPU
// tests drive with gloo; this file has no Python in it.
//
// Per level (after the replicated prefix, mc_shard_begin_replicated): ONE host synchronisation — the all-gather of the ranks'
// frontier sizes and verdicts — then rounds of  expand -> pack -> all-to-all (fingerprints, fixed capacity, counts in band) ->
// probe -> all-to-all (answers) -> keep,  all enqueued without waiting (include/tlamc.h mc_shard_*_pack).  States stay on the
// rank that generated them ("stay" form); the prefix hands every rank the states of its last level it OWNS, a uniform sample.
#include "tlamc.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

extern "C" void mc_set_error_internal(const char *msg);

struct mc_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint32_t rank = 0, world = 1;
    int device = 0;
};

namespace {
// librccl.so is opened on first use, privately (RTLD_LOCAL): libtlamc.so loads on a machine without RCCL, and a host process
// that carries its own copy of the library (PyTorch does) keeps its symbols to itself.  $TLAMC_RCCL names another file.
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load() {
        if (h) return true;
        const char *env = getenv("TLAMC_RCCL");
        const char *names[] = {env && *env ? env : "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!h) { mc_set_error_internal((std::string("librccl.so cannot be loaded: ") + dlerror()).c_str()); return false; }
#define SYM(f) f = (decltype(f))dlsym(h, "nccl" #f); if (!f) { mc_set_error_internal("librccl.so lacks nccl" #f); h = nullptr; return false; }
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(GroupStart) SYM(GroupEnd) SYM(Send) SYM(Recv) SYM(AllGather) SYM(GetErrorString)
#undef SYM
        return true;
    }
} R;
int fail_hip(hipError_t e, const char *what) {
    mc_set_error_internal((std::string(what) + ": " + hipGetErrorString(e)).c_str());
    return MC_EHIP;
}
int fail_nccl(ncclResult_t r, const char *what) {
    mc_set_error_internal((std::string(what) + ": " + R.GetErrorString(r)).c_str());
    return MC_ERCCL;
}
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail_hip(e_, #x); } while (0)
#define NCCLCK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail_nccl(r_, #x); } while (0)

// equal-split all-to-all of `count` elements per peer
int all_to_all(mc_comm *c, const void *send, void *recv, size_t count, ncclDataType_t t, size_t elem) {
    NCCLCK(R.GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {
        NCCLCK(R.Send((const char *)send + (size_t)p * count * elem, count, t, (int)p, c->comm, c->stream));
        NCCLCK(R.Recv((char *)recv + (size_t)p * count * elem, count, t, (int)p, c->comm, c->stream));
    }
    NCCLCK(R.GroupEnd());
    return MC_OK;
}
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int need(size_t n) {
        if (n <= bytes) return MC_OK;
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
        HIPCK(hipMalloc(&p, n));
        bytes = n;
        return MC_OK;
    }
    ~DevBuf() { if (p) hipFree(p); }
};
}  // namespace

extern "C" {

int mc_comm_unique_id(uint8_t id_out[MC_COMM_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= MC_COMM_ID_BYTES, "ncclUniqueId does not fit MC_COMM_ID_BYTES");
    if (!id_out) return MC_EBADCFG;
    ncclUniqueId id;
    if (!R.load()) return MC_ERCCL;
    NCCLCK(R.GetUniqueId(&id));
    memset(id_out, 0, MC_COMM_ID_BYTES);
    memcpy(id_out, &id, sizeof id);
    return MC_OK;
}

int mc_comm_create(const uint8_t id[MC_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, mc_comm **out) {
    if (!id || !out || !world || rank >= world || world > 8) { mc_set_error_internal("mc_comm_create: rank / world (1..8 ranks)"); return MC_EBADCFG; }
    if (!R.load()) return MC_ERCCL;
    HIPCK(hipSetDevice(device));
    mc_comm *c = new mc_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t r = R.CommInitRank(&c->comm, (int)world, uid, (int)rank);
    if (r != ncclSuccess) { delete c; return fail_nccl(r, "ncclCommInitRank"); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { R.CommDestroy(c->comm); delete c; return fail_hip(e, "hipStreamCreate"); }
    *out = c;
    return MC_OK;
}

void mc_comm_destroy(mc_comm *c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    if (c->comm && R.h) R.CommDestroy(c->comm);
    delete c;
}

int mc_shard_run(mc_engine *e, mc_comm *c, const mc_shard_opts *o, mc_result *out) {
    if (!e || !c || !o || !out) return MC_EBADCFG;
    const uint32_t P = c->world;
    const uint64_t chunk = o->chunk_states ? o->chunk_states : (1ull << 19);
    const uint64_t fan = o->packed_fanout ? o->packed_fanout : 16;
    HIPCK(hipSetDevice(c->device));
    memset(out, 0, sizeof *out);
    out->violated_invariant = -1;
    int rc;
    if ((rc = mc_shard_set_stream(e, (void *)c->stream, 1))) return rc;
    struct Restore { mc_engine *e; ~Restore() { mc_shard_set_stream(e, nullptr, 0); } } restore{e};

    // ---- the small levels: the same fused BFS on every rank, then each rank keeps the states it owns
    std::vector<uint64_t> levels(MC_MAX_LEVELS);
    uint32_t nlev = MC_MAX_LEVELS;
    const uint64_t until = (o->replicate_until ? o->replicate_until : (1ull << 15)) * P;
    if ((rc = mc_shard_begin_replicated(e, until, o->max_distinct, o->max_levels, levels.data(), &nlev))) return rc;
    levels.resize(nlev);

    DevBuf d_info, d_all, d_send[2], d_recv[2], d_ans[2], d_back[2];
    if ((rc = d_info.need(2 * sizeof(uint64_t))) || (rc = d_all.need((size_t)P * 2 * sizeof(uint64_t)))) return rc;
    std::vector<uint64_t> all(2 * (size_t)P), sizes(P);
    // ONE collective and one host wait per level: every rank learns every rank's frontier size and verdict
    auto level_info = [&](uint64_t local_n, int32_t verdict, uint64_t &frontier, int32_t &worst) -> int {
        const uint64_t mine[2] = {local_n, (uint64_t)verdict};
        HIPCK(hipMemcpyAsync(d_info.p, mine, sizeof mine, hipMemcpyHostToDevice, c->stream));
        NCCLCK(R.AllGather(d_info.p, d_all.p, 2, ncclUint64, c->comm, c->stream));
        HIPCK(hipMemcpyAsync(all.data(), d_all.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        frontier = 0;
        worst = 0;
        for (uint32_t p = 0; p < P; ++p) {
            sizes[p] = all[2 * p];
            frontier += sizes[p];
            worst = std::max(worst, (int32_t)all[2 * p + 1]);
        }
        return MC_OK;
    };
    uint64_t local_n = 0, gen = 0, dl = 0, frontier = 0;
    int32_t verdict = 0, worst = 0;
    if ((rc = mc_shard_level_size(e, &local_n)) || (rc = mc_shard_counters(e, &gen, &dl, &verdict))) return rc;
    if ((rc = level_info(local_n, verdict, frontier, worst))) return rc;
    uint64_t cum = 0;
    for (uint64_t v : levels) cum += v;
    if (worst != 0) frontier = levels.empty() ? 0 : levels.back();
    bool budget = false;

    while (frontier > 0 && worst == 0) {
        if ((o->max_levels && levels.size() >= o->max_levels) || (o->max_distinct && cum >= o->max_distinct)) { budget = true; break; }
        uint64_t max_n = 0;
        for (uint32_t p = 0; p < P; ++p) max_n = std::max(max_n, sizes[p]);
        const uint64_t rounds = (max_n + chunk - 1) / chunk;
        const uint64_t mine = sizes[c->rank];
        auto launch = [&](uint64_t r) -> int {
            const uint64_t first = std::min(r * chunk, mine), n = std::min(chunk, mine - first);
            // a rank routes (P - 1) / P of its candidates over P owners; the capacity every rank derives is that of the level's
            // largest chunk (the same number everywhere: the exchanges are equal-split)
            return mc_shard_expand_launch(e, (uint32_t)(r & 1), first, n, P * (std::min(chunk, max_n) * fan / P + 4096));
        };
        if (rounds && (rc = launch(0))) return rc;
        for (uint64_t r = 0; r < rounds; ++r) {
            const uint32_t slot = (uint32_t)(r & 1);
            uint64_t n_round = 0;
            for (uint32_t p = 0; p < P; ++p) n_round = std::max(n_round, std::min(chunk, sizes[p] > r * chunk ? sizes[p] - r * chunk : 0));
            const uint64_t cap = std::min(n_round * fan * (P - 1) / ((uint64_t)P * P) + 1024, std::min(chunk, max_n) * fan / P + 4096);
            const size_t total = (size_t)P * cap;
            if ((rc = d_send[slot].need(total * 8)) || (rc = d_recv[slot].need(total * 8)) || (rc = d_ans[slot].need(total)) ||
                (rc = d_back[slot].need(total)))
                return rc;
            if ((rc = mc_shard_expand_pack(e, slot, (uint64_t *)d_send[slot].p, cap))) return rc;  // behind expand r, no host wait
            if (r + 1 < rounds && (rc = launch(r + 1))) return rc;                                   // overlaps the exchange below
            if ((rc = all_to_all(c, d_send[slot].p, d_recv[slot].p, cap, ncclUint64, 8))) return rc;
            if ((rc = mc_shard_probe_pack(e, (const uint64_t *)d_recv[slot].p, cap, (uint8_t *)d_ans[slot].p))) return rc;
            if ((rc = mc_shard_wait_keep(e, slot))) return rc;  // the slot's previous keep has read d_back[slot]
            if ((rc = all_to_all(c, d_ans[slot].p, d_back[slot].p, cap, ncclUint8, 1))) return rc;
            if ((rc = mc_shard_keep_pack(e, slot, (const uint8_t *)d_back[slot].p, cap))) return rc;
        }
        uint64_t new_local = 0;
        if ((rc = mc_shard_end_level(e, &new_local))) return rc;  // waits for the engine's streams; device errors surface here
        if ((rc = mc_shard_counters(e, &gen, &dl, &verdict))) return rc;
        if ((rc = level_info(new_local, verdict, frontier, worst))) return rc;
        if (frontier > 0) {
            if (levels.size() >= MC_MAX_LEVELS) { mc_set_error_internal("more BFS levels than MC_MAX_LEVELS"); return MC_EBADCFG; }
            levels.push_back(frontier);
            cum += frontier;
        }
    }
    if (frontier > 0 && worst == 0 && (rc = mc_shard_check_frontier(e))) return rc;  // a budget stop leaves a level unexpanded
    if ((rc = mc_shard_counters(e, &gen, &dl, &verdict))) return rc;
    {   // global counters: sum of generated, worst verdict
        const uint64_t mine2[2] = {gen, (uint64_t)verdict};
        HIPCK(hipMemcpyAsync(d_info.p, mine2, sizeof mine2, hipMemcpyHostToDevice, c->stream));
        NCCLCK(R.AllGather(d_info.p, d_all.p, 2, ncclUint64, c->comm, c->stream));
        HIPCK(hipMemcpyAsync(all.data(), d_all.p, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
        gen = 0;
        worst = 0;
        for (uint32_t p = 0; p < P; ++p) { gen += all[2 * p]; worst = std::max(worst, (int32_t)all[2 * p + 1]); }
    }
    out->distinct = cum;
    out->generated = gen;
    out->queue_left = frontier;
    out->depth = (uint32_t)levels.size();
    out->levels = (uint32_t)levels.size();
    for (size_t k = 0; k < levels.size(); ++k) out->level_distinct[k] = levels[k];
    out->verdict = worst != 0 ? worst : budget ? MC_V_BUDGET : MC_V_OK;
    return MC_OK;
}

}  // extern "C"
