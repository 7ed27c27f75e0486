// shard_rccl.cpp — the hip-rccl back-end behind the C ABI: the collectives of the fingerprint-sharded
// search issued with RCCL (ncclSend / ncclRecv groups = all-to-all, ncclAllGather over xGMI) on one HIP stream per rank, and
// the binding of the ONE level loop (shard_loop.h) to the engine's step calls (mc_shard_* of include/tlamc.h).  One process per
// GPU; a host in any language creates the communicator from 128 bytes it ships between its ranks itself (mc_comm_unique_id on
// rank 0, mc_comm_create everywhere) and calls mc_shard_run; `mc X.tla -gpus P` does exactly that with forked ranks and a file
// (mc_main.cpp), `bench.py --gpus N` with spawned ranks.  A host that owns its own collectives (torch.distributed: tla_rust_amd/
// sharded.py) hands them over as an mc_transport and runs the same loop (mc_shard_run_transport).  No Python in this file.
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

#include "shard_loop.h"

struct mc_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint32_t rank = 0, world = 1;
    int device = 0;
    void *d_scratch = nullptr;  // staging of the host all-gather
    size_t scratch_bytes = 0;
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

// ---- the RCCL transport (mc_transport of include/tlamc.h over one communicator)
void *rccl_alloc(void *user, size_t bytes) {
    mc_comm *c = (mc_comm *)user;
    void *p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void rccl_release(void *user, void *p) {
    (void)user;
    if (p) hipFree(p);  // (synchronises with the device: no kernel or collective still uses the buffer afterwards)
}
int rccl_all_to_all(void *user, const void *send, void *recv, uint64_t bytes) {
    mc_comm *c = (mc_comm *)user;
    NCCLCK(R.GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {
        NCCLCK(R.Send((const char *)send + (size_t)p * bytes, bytes, ncclUint8, (int)p, c->comm, c->stream));
        NCCLCK(R.Recv((char *)recv + (size_t)p * bytes, bytes, ncclUint8, (int)p, c->comm, c->stream));
    }
    NCCLCK(R.GroupEnd());
    return MC_OK;
}
int rccl_all_to_all_others(void *user, const void *send, void *recv, uint64_t bytes) {  // every block but the rank's own
    mc_comm *c = (mc_comm *)user;
    NCCLCK(R.GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        NCCLCK(R.Send((const char *)send + (size_t)p * bytes, bytes, ncclUint8, (int)p, c->comm, c->stream));
        NCCLCK(R.Recv((char *)recv + (size_t)p * bytes, bytes, ncclUint8, (int)p, c->comm, c->stream));
    }
    NCCLCK(R.GroupEnd());
    return MC_OK;
}
int rccl_all_to_all_v(void *user, const void *send, const uint64_t *so, const uint64_t *sb, void *recv, const uint64_t *ro, const uint64_t *rb) {
    mc_comm *c = (mc_comm *)user;
    NCCLCK(R.GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {  // both ends know a pair's size: an empty pair is skipped on both
        if (sb[p]) NCCLCK(R.Send((const char *)send + so[p], sb[p], ncclUint8, (int)p, c->comm, c->stream));
        if (rb[p]) NCCLCK(R.Recv((char *)recv + ro[p], rb[p], ncclUint8, (int)p, c->comm, c->stream));
    }
    NCCLCK(R.GroupEnd());
    return MC_OK;
}
int rccl_all_gather(void *user, const void *mine, void *all_out, uint64_t bytes) {
    mc_comm *c = (mc_comm *)user;
    HIPCK(hipSetDevice(c->device));
    const size_t need = (size_t)bytes * (c->world + 1);
    if (need > c->scratch_bytes) {
        if (c->d_scratch) hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        HIPCK(hipMalloc(&c->d_scratch, need + 4096));
        c->scratch_bytes = need + 4096;
    }
    char *d_mine = (char *)c->d_scratch, *d_all = d_mine + bytes;
    HIPCK(hipMemcpyAsync(d_mine, mine, bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCK(R.AllGather(d_mine, d_all, bytes, ncclUint8, c->comm, c->stream));
    HIPCK(hipMemcpyAsync(all_out, d_all, (size_t)bytes * c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCK(hipStreamSynchronize(c->stream));
    return MC_OK;
}

// ---- the engine's step calls as the loop sees them, and the three streams of a stay level
struct AbiOps {
    mc_engine *eng;
    int device = 0;
    hipStream_t comm = nullptr, work = nullptr, main_s = nullptr;
    bool sync_comm = false;  // transport without a stream: its collectives are synchronous host calls
    hipEvent_t ev[mc_shard::EV_COUNT] = {};
    bool on_host[mc_shard::EV_COUNT] = {};  // recorded "on" a synchronous transport: complete when record() returns
    uint64_t chunk = 0;
    int32_t is_traced = 0;
    mc_spec_desc desc;
    size_t W = 0;
    int init(mc_engine *e, const mc_transport *t) {
        eng = e;
        void *ms = nullptr;
        int rc = mc_shard_info(e, &ms, &chunk, &is_traced);
        if (rc) return rc;
        main_s = (hipStream_t)ms;
        comm = (hipStream_t)t->hip_stream;
        h_world = t->world;
        sync_comm = t->hip_stream == nullptr;
        HIPCK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
        for (auto &x : ev) HIPCK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        return mc_shard_set_stream(e, (void *)work, 1);
    }
    ~AbiOps() {
        if (eng) mc_shard_set_stream(eng, nullptr, 0);
        for (auto &x : ev) if (x) hipEventDestroy(x);
        if (h_counts) { if (main_s) hipStreamSynchronize(main_s); hipHostFree(h_counts); }
        if (work) { hipStreamSynchronize(work); hipStreamDestroy(work); }
    }
    hipStream_t stream(int s) const { return s == mc_shard::S_COMM ? comm : s == mc_shard::S_WORK ? work : main_s; }
    void record(int e, int s) {
        on_host[e] = s == mc_shard::S_COMM && sync_comm;
        if (!on_host[e]) hipEventRecord(ev[e], stream(s));
    }
    void wait(int s, int e) {
        if (on_host[e]) return;
        if (s == mc_shard::S_COMM && sync_comm) hipEventSynchronize(ev[e]);  // the host issues the collective: the host waits
        else hipStreamWaitEvent(stream(s), ev[e], 0);
    }
    void clear_counts(uint64_t *buf, uint32_t P, uint64_t cap) {
        for (uint32_t t = 0; t < P; ++t) hipMemsetAsync(buf + (uint64_t)t * cap, 0, sizeof(uint64_t), main_s);
    }
    void clear_bytes(void *p, size_t n, int s) { hipMemsetAsync(p, 0, n, stream(s)); }
    uint64_t chunk_limit() const { return chunk; }
    size_t state_bytes() const { return W; }
    bool traced() const { return is_traced != 0; }
    int resume(uint64_t *lv, uint32_t *n) { return mc_shard_resume(eng, lv, n); }
    int note_levels(const uint64_t *lv, uint32_t n, int32_t verdict) { return mc_shard_note_levels(eng, lv, n, verdict); }
    int begin() { return mc_shard_begin(eng); }
    int begin_replicated(uint64_t mf, uint64_t md, uint64_t ml, uint64_t *lv, uint32_t *n) { return mc_shard_begin_replicated(eng, mf, md, ml, lv, n); }
    int level_size(uint64_t *n) { return mc_shard_level_size(eng, n); }
    int expand_launch(uint32_t slot, uint64_t first, uint64_t count, uint64_t cap) { return mc_shard_expand_launch(eng, slot, first, count, cap); }
    int expand_finish(uint32_t slot, uint64_t *fp, uint64_t cap, uint64_t *counts) {
        const int rc = mc_shard_expand_finish(eng, slot, fp, cap, counts);  // (MC_EROUTE itself when the slot's pending list is what overflowed)
        return rc;
    }
    // Fixed-capacity rounds: the count word of every bucket is read back behind the compaction — asynchronously, on the expand stream,
    // into pinned memory — and folded when the level ends: the fullest bucket of the level and the entries of all buckets of the run
    // are what the loop sizes the next level's buckets from and reports as the volume the candidates needed (shard_loop.h `fill`).
    static constexpr size_t FILL_ROUNDS = 1024;
    uint64_t *h_counts = nullptr;
    size_t h_rounds = 0;
    uint32_t h_world = 0;
    uint64_t fill_max = 0, fill_lvl = 0, fill_sum = 0;
    void fold_counts() {
        for (size_t i = 0; i < h_rounds * h_world; ++i) {
            fill_max = std::max(fill_max, h_counts[i]);
            fill_sum += h_counts[i];
        }
        h_rounds = 0;
    }
    int expand_pack(uint32_t slot, uint64_t *fp, uint64_t cap) {
        int rc = mc_shard_expand_pack(eng, slot, fp, cap);
        if (rc) return rc;
        if (!h_counts) HIPCK(hipHostMalloc((void **)&h_counts, FILL_ROUNDS * 8 * sizeof(uint64_t)));
        if (h_rounds == FILL_ROUNDS) {  // (a level of more than a thousand rounds: fold what has arrived)
            HIPCK(hipStreamSynchronize(main_s));
            fold_counts();
        }
        for (uint32_t t = 0; t < h_world; ++t)
            HIPCK(hipMemcpyAsync(h_counts + h_rounds * h_world + t, fp + (uint64_t)t * cap, sizeof(uint64_t), hipMemcpyDeviceToHost, main_s));
        ++h_rounds;
        return MC_OK;
    }
    int probe(const uint64_t *fp, uint64_t n, uint8_t *ans) { return mc_shard_probe(eng, fp, n, ans); }
    int probe_pack(const uint64_t *fp, uint64_t cap, uint8_t *ans) { return mc_shard_probe_pack(eng, fp, cap, ans); }
    int keep_pack(uint32_t slot, const uint8_t *back, uint64_t cap) { return mc_shard_keep_pack(eng, slot, back, cap); }
    int wait_keep(uint32_t slot) { return mc_shard_wait_keep(eng, slot); }
    int keep(uint32_t slot, const uint8_t *back) { return mc_shard_keep_slot(eng, slot, back, nullptr); }
    int materialise_slot(uint32_t slot, const uint8_t *back, uint8_t *st, uint64_t cap, uint64_t *counts) {
        return mc_shard_materialise_slot(eng, slot, back, st, cap, counts);
    }
    int materialise_parents(uint32_t slot, uint64_t *out) { return mc_shard_materialise_parents(eng, slot, out); }
    int ingest(const uint8_t *st, uint64_t n) { return mc_shard_ingest(eng, st, n); }
    int ingest_parents(const uint64_t *pp, uint64_t n, uint32_t src) { return mc_shard_ingest_parents(eng, pp, n, src); }
    int end_level(uint64_t *n) {
        int rc = mc_shard_end_level(eng, n);  // waits for the engine's streams: the count words of the level's rounds have landed
        if (h_rounds) {
            if (rc) HIPCK(hipStreamSynchronize(main_s));  // (a failed level may have returned before its wait)
            fold_counts();
        }
        fill_lvl = fill_max;
        fill_max = 0;
        return rc;
    }
    int route_fill(uint64_t *mx, uint64_t *sum) { *mx = fill_lvl; *sum = fill_sum; return MC_OK; }
    int counters(uint64_t *g, uint64_t *d, int32_t *v) { return mc_shard_counters(eng, g, d, v); }
    int check_frontier() { return mc_shard_check_frontier(eng); }
    int violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *v, int32_t *inv) { return mc_shard_violation(eng, found, idx, slot, v, inv); }
    int fetch(uint64_t idx, uint8_t *st, uint32_t *pr, uint64_t *pi, uint32_t *ps) { return mc_shard_fetch(eng, idx, st, pr, pi, ps); }
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
    if (c->d_scratch) hipFree(c->d_scratch);
    if (c->comm && R.h) R.CommDestroy(c->comm);
    delete c;
}

int mc_comm_transport(mc_comm *c, mc_transport *out) {
    if (!c || !out) return MC_EBADCFG;
    memset(out, 0, sizeof *out);
    out->user = c;
    out->rank = c->rank;
    out->world = c->world;
    out->hip_stream = (void *)c->stream;
    out->alloc = rccl_alloc;
    out->release = rccl_release;
    out->all_to_all = rccl_all_to_all;
    out->all_to_all_v = rccl_all_to_all_v;
    out->all_gather = rccl_all_gather;
    out->all_to_all_others = rccl_all_to_all_others;
    return MC_OK;
}

int mc_comm_all_gather(mc_comm *c, const void *mine, void *all_out, uint64_t bytes) {
    if (!c || !mine || !all_out || !bytes) return MC_EBADCFG;
    return rccl_all_gather(c, mine, all_out, bytes);
}

extern "C" size_t mc_engine_state_bytes_internal(mc_engine *e);

int mc_shard_run_transport(mc_engine *e, const mc_transport *t, const mc_shard_opts *o, mc_result *out) {
    if (!e || !t || !o || !out || !t->all_to_all || !t->all_to_all_v || !t->all_gather || !t->alloc || !t->release) return MC_EBADCFG;
    AbiOps ops;
    ops.eng = nullptr;
    int rc = ops.init(e, t);
    if (rc) return rc;
    ops.W = mc_engine_state_bytes_internal(e);
    return mc_shard::run_restarting(ops, *t, *o, out);
}

int mc_shard_trace_transport(mc_engine *e, const mc_transport *t, uint8_t *states_out, int32_t *slots_out, size_t *n_inout, int32_t *final_slot) {
    if (!e || !t || !states_out || !slots_out || !n_inout || !t->all_gather) return MC_EBADCFG;
    AbiOps ops;
    ops.eng = nullptr;
    int rc = ops.init(e, t);
    if (rc) return rc;
    ops.W = mc_engine_state_bytes_internal(e);
    mc_shard::Loop<AbiOps> loop(ops, *t);
    return loop.trace(states_out, slots_out, n_inout, final_slot);
}

int mc_shard_run(mc_engine *e, mc_comm *c, const mc_shard_opts *o, mc_result *out) {
    mc_transport t;
    int rc = mc_comm_transport(c, &t);
    if (rc) return rc;
    HIPCK(hipSetDevice(c->device));
    return mc_shard_run_transport(e, &t, o, out);
}

int mc_shard_trace(mc_engine *e, mc_comm *c, uint8_t *states_out, int32_t *slots_out, size_t *n_inout, int32_t *final_slot) {
    mc_transport t;
    int rc = mc_comm_transport(c, &t);
    if (rc) return rc;
    HIPCK(hipSetDevice(c->device));
    return mc_shard_trace_transport(e, &t, states_out, slots_out, n_inout, final_slot);
}

}  // extern "C"
