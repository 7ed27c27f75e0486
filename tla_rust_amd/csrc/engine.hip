// engine.hip — the MI355X (gfx950) BFS engine behind include/tlamc.h: the HOST half (Engine: buffers, streams, the level loop of a fused
// run, the step calls of a sharded one, checkpoints, the C ABI); the kernels it launches are in engine_kernels.h.
//
// Per BFS level the frontier (a contiguous index range of the state arena in HBM) is processed in chunks of `chunk_states`
// (bench.py: 2^23) with no host synchronisation inside a level; the host reads one small counter block per level, and while
// levels are small it enqueues eight of them blind (LevelCtl).  The kernels, in the order they matter:
//
//   k_expand_family<Spec, ROUTE>   (specs with action families: raft)  one wavefront = one arena block of 64 parents, one
//                       workgroup = two (MC_EXPAND_WAVES; four until round 5).  Guards -> enabled (parent, slot) pairs, dense slots and the actions of in-flight
//                       messages evaluated by the parent's lane, the sparse fixed slots bucketed per family in LDS and evaluated
//                       64 pairs of ONE family at a time; successors that can never be stored are counted, not evaluated
//                       (S::GENERATED_ONLY); a candidate's fingerprint is the parent's plus O(delta) terms; a per-wavefront
//                       filter drops repeats; 64 candidates at a time probe the seen-set (open addressing over 32- or 64-byte
//                       buckets in HBM, agent-scope atomicCAS on write-once slots).  ROUTE = false (fused runs): the
//                       survivors wait in LDS and the WORKGROUP (a pair of wavefronts) writes them at its end — pooled, counting-sorted by action
//                       class, one atomicAdd on the arena's fill level, row copy + patch from the parent's Summary (the
//                       in-wave tail; wave_write_survivors).  ROUTE = true (sharded runs): candidates of other owners go to
//                       exchange buckets, the rank's own are probed here and go through the new-list.
//   k_expand_insert<Spec, ROUTE>   the slot-by-slot form of the same (lane = parent, loop = action slot; every other spec).
//   k_materialise<Spec> one lane per entry of the new-list (the survivors a wavefront had no list space for, every survivor of
//                       a spec without in-wave writes, of a fast-growing level, or of a sharded run): re-evaluates its
//                       (parent, slot) and writes the full successor (coalesced: consecutive lanes own consecutive arena
//                       indices), plus (parent, slot) for counterexamples.  Runs on a second stream beside the next expand.
//   k_expand + k_insert the round-1 form — a slot-major candidate matrix between two kernels — kept behind MC_F_MATRIX for A/B.
//
// Algorithmic HBM bytes per distinct state = 2*W + 8*(G/D)  (SURVEY.md §8d): each state is written once and read once, each
// in-model successor touches one seen-set word; DESIGN.md §5 has what the kernels really move and what bounds them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <memory>
#include <type_traits>
#include <string>
#include <vector>

#include "spec_registry.h"

namespace mc {

// The library is built from this one source compiled several times (MC_TU = 0: C ABI + host helpers;
// 1..5: the engine + kernels of one group of specs each), so the per-spec kernels compile in parallel.
#ifndef MC_TU
#define MC_TU -1  // single translation unit: everything
#endif
}  // namespace mc
extern "C" void mc_set_error_internal(const char *msg);
namespace mc {
static void set_error(const std::string &s) { mc_set_error_internal(s.c_str()); }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
            return MC_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

}  // namespace mc
#include "engine_kernels.h"   // namespace mc { ... every kernel ... }
#include "engine_pairs.h"     // k_expand_pairs: the by-pairs expand + insert + write kernel (specs with S::PAIR_FAMILIES)
namespace mc {

// ------------------------------------------------------------------------------------- host side
struct EngineBase {
    void *owned_device_blob = nullptr;  // program image of a compiled PlusCal spec (spec_vm.h)
    uint64_t program_hash = 0;          // ... and a hash of that image + its entry points: what a checkpoint of it is matched by
    mc_progress_fn progress_fn = nullptr;  // mc_engine_set_progress
    void *progress_user = nullptr;
    double progress_interval = 1.0;
    std::chrono::steady_clock::time_point progress_last;
    bool stop_requested = false;  // mc_engine_request_stop (from a progress callback): run() stops before the next level, verdict "budget"
    void report_progress(uint32_t levels, uint64_t generated, uint64_t distinct, uint64_t queue) {
        if (!progress_fn) return;
        const auto now = std::chrono::steady_clock::now();
        if (std::chrono::duration<double>(now - progress_last).count() < progress_interval) return;
        progress_last = now;
        progress_fn(progress_user, levels, generated, distinct, queue);
    }
    virtual ~EngineBase() { if (owned_device_blob) hipFree(owned_device_blob); }
    virtual int run(mc_result *out) = 0;
    virtual int step(uint32_t levels, mc_result *out) = 0;
    virtual int trace(uint8_t *states_out, int32_t *actions_out, size_t *n_inout) = 0;
    virtual int kernel_stats(mc_kernel_stats *out) = 0;
    virtual int read_states(uint64_t first, uint64_t count, uint8_t *out) = 0;
    virtual int debug_reexpand(unsigned extra_flags, double *ms) = 0;
    virtual int debug_phases(uint64_t *out48, int reset) = 0;
    virtual int debug_flags(uint32_t set, uint32_t clear) = 0;
    virtual int checkpoint(const char *path) = 0;
    virtual int restore(const char *path) = 0;
    virtual int shard_begin() = 0;
    virtual int shard_begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) = 0;
    virtual int shard_level_size(uint64_t *n) = 0;
    virtual int shard_set_stream(void *hip_stream, int enable) = 0;
    virtual int shard_expand_launch(unsigned slot, uint64_t first, uint64_t count, uint64_t send_cap) = 0;
    virtual int shard_expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int shard_probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) = 0;
    virtual int shard_expand_pack(unsigned slot, uint64_t *send_fp, uint64_t cap) = 0;
    virtual int shard_probe_pack(const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) = 0;
    virtual int shard_keep_pack(unsigned slot, const uint8_t *answers_back, uint64_t cap) = 0;
    virtual int shard_wait_keep(unsigned slot) = 0;
    virtual int shard_materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int shard_ingest(const uint8_t *recv_states, uint64_t n) = 0;
    virtual int shard_keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) = 0;
    virtual int shard_end_level(uint64_t *new_local) = 0;
    virtual int shard_counters(uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) = 0;
    virtual int shard_check_frontier() = 0;
    virtual int shard_materialise_parents(unsigned slot, uint64_t *send_parents) = 0;
    virtual int shard_ingest_parents(const uint64_t *recv_parents, uint64_t n, unsigned src_rank) = 0;
    virtual int shard_violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) = 0;
    virtual int shard_fetch(uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) = 0;
    virtual int shard_info(void **main_stream, uint64_t *chunk_states, int32_t *traced) = 0;
    virtual int shard_note_levels(const uint64_t *levels, uint32_t n, int32_t verdict) = 0;
    virtual int shard_resume(uint64_t *levels_out, uint32_t *nlevels) = 0;
    virtual int shard_checkpoint(const char *path) = 0;
    virtual int shard_restore(const char *path) = 0;
    virtual size_t state_bytes() const = 0;
};

static uint64_t round_pow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct KTimer {
    std::vector<hipEvent_t> pool;
    struct Pending { int which; size_t e0, e1; uint64_t units; };
    std::vector<Pending> pending;
    size_t used = 0;
    bool enabled = false;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void resolve(mc_kernel_stat *stats /*[3]*/) {
        for (auto &p : pending) {
            float ms = 0;
            hipEventElapsedTime(&ms, pool[p.e0], pool[p.e1]);
            stats[p.which].launches++;
            stats[p.which].ms_total += ms;
            stats[p.which].units += p.units;
        }
        pending.clear();
        used = 0;
    }
    ~KTimer() { for (auto e : pool) hipEventDestroy(e); }
};

// specs whose stored row is not the row the C ABI hands out (spec_gen.h: generated code packs its cells): S::EXPORT_WORDS, S::export_row
template <class S, class = void>
struct HasExport : std::false_type {};
template <class S>
struct HasExport<S, decltype((void)S::EXPORT_WORDS)> : std::true_type {};
template <class S, class = void>
struct PackedRows : std::false_type {};
template <class S>
struct PackedRows<S, decltype((void)S::PACKED_ROWS)> : std::integral_constant<bool, S::PACKED_ROWS> {};

template <class S>
struct Engine : EngineBase {
    using Params = typename S::Params;
    int W = 1;  // 64-bit words per packed state
    unsigned max_slots = 1;
    Params prm;
    mc_spec_desc desc;
    mc_config cfg;
    hipStream_t stream = nullptr, stream2 = nullptr;  // expand+insert on `stream`, materialise on `stream2`
    // The chunks of ONE level are independent of each other (they read the frontier, probe / insert with atomics, take arena indices
    // with atomics): the odd ones go to a stream of their own, so that the first workgroups of chunk c+1 fill the CUs the last
    // workgroups of chunk c leave idle (a launch of 2^23 parents is 32 rounds of the 2048 resident workgroups: the last one is half
    // empty) instead of waiting for the stream order.  $TLAMC_EXPAND_STREAMS = 2 switches it on (A/B; off by default until measured).
    hipStream_t stream_b = nullptr;
    hipEvent_t ev_level = nullptr;
    hipEvent_t ev_e[2] = {nullptr, nullptr}, ev_m[2] = {nullptr, nullptr};
    uint64_t *d_arena = nullptr, *d_table = nullptr, *d_cand = nullptr;
    uint32_t *d_newlist = nullptr;
    uint64_t *d_newfp = nullptr;  // fingerprints of the new-list entries (specs with apply_known_fp)
    uint16_t *d_nsl = nullptr, *d_pslot = nullptr;
    uint64_t *d_inittmp = nullptr;
    uint32_t *d_parent = nullptr;
    uint8_t *d_prank = nullptr;  // sharded runs with MC_F_TRACE: rank a state's parent lives on (0xff = this rank)
    DevCounters *d_ctr = nullptr, *h_ctr = nullptr;
    LevelCtl *d_lc = nullptr, *h_lc = nullptr;
    uint64_t table_cap = 0, arena_cap = 0, chunk = 0, row_stride = 0, seg_cap = 0;
    bool seen_sparse = false;
    // slot slices of a generic-kernel launch (k_expand_insert): as many as it takes to give the device a few thousand wavefronts,
    // for specs whose slots all go through the loop (no unrolled prefix) and that have enough of them; TLAMC_NOSLICE=1 = A/B
    unsigned slices_for(uint64_t ncols, bool blind) const {
        // (a lowering asks for it with SLICE_SLOTS: worth it when one slot evaluation is expensive — the Paxos family's witness
        // enumeration, the bytecode interpreter; for atomic_add's two-instruction slots the slices only repeat the parent loads)
        if (!WantsSlices<S>::value || S::FIX_SLOTS != 0 || UsesFamilies<S>::value || pairs_only() || use_matrix || max_slots < 16 || no_slices) return 1;  // (the by-pairs kernel has no slices: its lanes are pairs already)
        const uint64_t waves = (ncols + 63) / 64;
        uint64_t sg = blind ? 32 : (16384 + waves - 1) / waves;  // a blind level launches its CAPACITY: the frontier itself is smaller
        if (sg > 32) sg = 32;
        if (sg > max_slots / 4) sg = max_slots / 4;
        return sg < 1 ? 1u : (unsigned)sg;
    }
    bool no_slices = getenv("TLAMC_NOSLICE") != nullptr;
    uint64_t seen_arg() const { return seen_sparse ? ((table_cap / MC_SPARSE_SLOTS) | SEEN_SPARSE) : table_cap / 8; }
    KTimer timer;
    mc_kernel_stat kstat[3];
    // counterexample of the last run
    bool have_viol = false;
    unsigned long long last_viol = ~0ull;
    std::vector<uint64_t> level_start;

    int alloc() {
        W = S::words(prm);
        max_slots = (unsigned)S::max_slots(prm);
        use_matrix = (cfg.flags & MC_F_MATRIX) != 0;
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        {   // A/B knob (measured in DESIGN.md section 4): TLAMC_PRIO=1 gives the materialise stream the highest priority, =2 the lowest
            const char *pe = getenv("TLAMC_PRIO");
            int lo_p = 0, hi_p = 0;
            hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);  // lo_p = least, hi_p = greatest priority (numerically smaller)
            if (pe && *pe == '1') HIP_TRY(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, hi_p));
            else if (pe && *pe == '2') HIP_TRY(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, lo_p));
            else HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        }
        for (int i = 0; i < 2; i++) { HIP_TRY(hipEventCreateWithFlags(&ev_e[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ev_m[i], hipEventDisableTiming)); }
        {
            const char *es = getenv("TLAMC_EXPAND_STREAMS");
            if (es && *es == '2') {
                HIP_TRY(hipStreamCreateWithFlags(&stream_b, hipStreamNonBlocking));
                HIP_TRY(hipEventCreateWithFlags(&ev_level, hipEventDisableTiming));
            }
        }
        table_cap = cfg.table_capacity ? cfg.table_capacity : (1ull << 24);
        table_cap = (table_cap + 63) / 64 * 64;  // whole 8-slot buckets; any size (seen_insert), at most 2^32 buckets
        if (table_cap / 8 > 0xffffffffull) { set_error("table_capacity: at most 2^35 - 8 slots per device"); return MC_EBADCFG; }
        arena_cap = cfg.arena_capacity ? cfg.arena_capacity : (1ull << 22);
        arena_cap = (arena_cap + 63) & ~63ull;
        if (arena_cap >= (1ull << 32) - 1) { set_error("arena_capacity must be < 2^32 states"); return MC_EBADCFG; }
        // a table that can never be more than a third full is probed 32 bytes at a time (seen_insert); $TLAMC_SPARSE_RATIO = A/B knob
        // for "a third" (slots per arena state, default 3)
        static const double sparse_ratio = getenv("TLAMC_SPARSE_RATIO") && atof(getenv("TLAMC_SPARSE_RATIO")) >= 1.25 ? atof(getenv("TLAMC_SPARSE_RATIO")) : 3.0;
        seen_sparse = (double)table_cap >= sparse_ratio * (double)arena_cap && table_cap / MC_SPARSE_SLOTS <= 0xffffffffull && !getenv("TLAMC_DENSE_TABLE");
        chunk = cfg.chunk_states ? cfg.chunk_states : (1ull << 18);
        chunk = (chunk + 255) & ~255ull;
        if (chunk > (1ull << 24) - 256) chunk = (1ull << 24) - 256;  // a column index must fit 24 bits (new-list entries: (slot << 24) | column)
        if (max_slots > 255) { set_error("spec has more than 255 action slots per state"); return MC_EBADCFG; }
        row_stride = chunk + 256;
        HIP_TRY(hipMalloc(&d_arena, arena_cap * W * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_table, table_cap * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_cand, (size_t)(use_matrix ? max_slots : 1) * row_stride * sizeof(uint64_t)));
        seg_cap = (row_stride / NSHARD + 256) * max_slots;
        seg_cap += seg_cap / 64 + 65;   // (round 6: an in-wave writer parks its overflow here as chunks of 64 entries + a link: 65 / 64 of the worst case)
        HIP_TRY(hipMalloc(&d_newlist, (size_t)2 * NSHARD * seg_cap * sizeof(uint32_t)));
        if (HasKnownFp<S>::value && !use_matrix) HIP_TRY(hipMalloc(&d_newfp, (size_t)2 * NSHARD * seg_cap * sizeof(uint64_t)));
        {
            const uint64_t ni = S::num_init(prm);
            HIP_TRY(hipMalloc(&d_inittmp, (size_t)(ni < chunk ? ni : chunk) * W * sizeof(uint64_t)));
        }
        HIP_TRY(hipMalloc(&d_nsl, row_stride * sizeof(uint16_t)));
        if (cfg.flags & MC_F_TRACE) {
            HIP_TRY(hipMalloc(&d_parent, arena_cap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&d_pslot, arena_cap * sizeof(uint16_t)));
            if (cfg.shard_count > 1) {
                HIP_TRY(hipMalloc(&d_prank, arena_cap));
                HIP_TRY(hipMemset(d_prank, 0xff, arena_cap));
            }
        }
        HIP_TRY(hipMalloc(&d_ctr, sizeof(DevCounters)));
        HIP_TRY(hipHostMalloc(&h_ctr, sizeof(DevCounters)));
        HIP_TRY(hipMalloc(&d_lc, sizeof(LevelCtl)));
        HIP_TRY(hipHostMalloc(&h_lc, sizeof(LevelCtl)));
        timer.enabled = (cfg.flags & MC_F_TIMING) != 0;
        return MC_OK;
    }
    ~Engine() override {
        if (d_arena) hipFree(d_arena);
        if (d_table) hipFree(d_table);
        if (d_cand) hipFree(d_cand);
        if (d_newlist) hipFree(d_newlist);
        if (d_newfp) hipFree(d_newfp);
        if (d_nsl) hipFree(d_nsl);
        if (d_inittmp) hipFree(d_inittmp);
        for (auto &q : sl) {
            if (q.rt_fp) { hipFree(q.rt_fp); hipFree(q.rt_src); hipFree(q.pend_src); }
            if (q.rt_cur) hipFree(q.rt_cur);
        }
        if (d_new_src) { hipFree(d_new_src); hipFree(d_incl); }
        if (d_ends) { hipFree(d_ends); hipHostFree(h_ends); hipHostFree(h_cur); }
        for (auto &ev : ev_slot) if (ev) hipEventDestroy(ev);
        for (auto &ev : ev_keep) if (ev) hipEventDestroy(ev);
        if (ev_ans) hipEventDestroy(ev_ans);
        if (ev_append) hipEventDestroy(ev_append);
        if (d_scan_tmp) hipFree(d_scan_tmp);
        if (d_parent) hipFree(d_parent);
        if (d_pslot) hipFree(d_pslot);
        if (d_prank) hipFree(d_prank);
        if (d_ctr) hipFree(d_ctr);
        if (h_ctr) hipHostFree(h_ctr);
        if (d_lc) hipFree(d_lc);
        if (h_lc) hipHostFree(h_lc);
        for (int i = 0; i < 2; i++) { if (ev_e[i]) hipEventDestroy(ev_e[i]); if (ev_m[i]) hipEventDestroy(ev_m[i]); }
        if (stream2) hipStreamDestroy(stream2);
        if (stream_b) hipStreamDestroy(stream_b);
        if (ev_level) hipEventDestroy(ev_level);
        if (stream) hipStreamDestroy(stream);
    }

    template <class F>
    void timed(int which, uint64_t units, F &&launch, hipStream_t on = nullptr) {
        if (!timer.enabled) { launch(); return; }
        if (!on) on = stream;
        hipEvent_t a = timer.get(), b = timer.get();
        size_t ia = timer.used - 2, ib = timer.used - 1;
        hipEventRecord(a, on);
        launch();
        hipEventRecord(b, on);
        timer.pending.push_back({which, ia, ib, units});
    }

    int read_counters() {
        if (stream_b) HIP_TRY(hipStreamSynchronize(stream_b));
        HIP_TRY(hipStreamSynchronize(stream2));
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (timer.enabled) timer.resolve(kstat);
        return MC_OK;
    }
    int check_dev_error() {
        if (h_ctr->error & DEV_EOVERFLOW) { set_error("packed-state capacity exceeded (raft messages / elections / allLogs slots, or a PlusCal sequence longer than its cells, or a recursive PlusCal procedure deeper than $TLAMC_PCAL_STACK frames)"); return MC_EOVERFLOW; }
        if (h_ctr->error & DEV_ETABLE) { set_error("seen-set full: raise table_capacity"); return MC_ETABLEFULL; }
        if (h_ctr->error & DEV_EROUTE) { set_error("sharded round: an exchange bucket is full (raise the fan-out allowance / send capacity)"); return MC_EROUTE; }
        if (h_ctr->error & DEV_EARENA) { set_error("state arena full: raise arena_capacity"); return MC_EARENA; }
        return MC_OK;
    }

    bool use_matrix = false;
    // fused runs of a by-family spec: the expand wavefronts write their own survivors (MC_F_NOINWAVE = A/B: everything through the
    // new-list and k_materialise, as in rounds 1-3)
    double inwave_growth_limit = getenv("TLAMC_INWAVE_GROWTH") ? atof(getenv("TLAMC_INWAVE_GROWTH")) : 10.0;  // (A/B knob.  Round 4, profiles/r04l:
    // t3 — levels grow by at most 2.2 x — was best with every level in-wave, the 5-server model — 2.4 x and more on all 18 levels — with
    // none, 230.6 against 239.0 ms: the limit was 2.3.  Round 5: with the writer's scratch traffic gone the in-wave tail wins there too,
    // 191.9 / 192.1 against 205.0 / 203.0 ms at 2.3 (profiles/r05i_raft5_inwave_growth.jsonl); 10 = in effect every level)
    bool inwave_ok() const { return (UsesFamilies<S>::value || UsesPairs<S>::value) && !use_matrix && !(cfg.flags & (MC_F_NOFAMILY | MC_F_NOINWAVE)); }
    // by-pairs specs (engine_pairs.h) write EVERY new state in the expand kernel: nothing ever reaches the new-list, so a fused run launches
    // neither k_materialise nor k_commit for them
    bool pairs_only() const { return UsesPairs<S>::value && inwave_ok(); }
    bool mat_inwave_level = false;     // the level being enqueued writes in-wave: k_materialise sees the overflow only
    uint64_t mat_list_hint = 0;        // states per chunk the level before sent through the new-list
    void set_inwave(RouteArgs &rt) const {
        if (!inwave_ok()) return;
        rt.arena_w = d_arena;
        rt.arena_cap = arena_cap;
        rt.parent = d_parent;
        rt.pslot = d_pslot;
    }
    // materialise + commit of the chunk whose survivors are in new-list `parity`, on the second stream: it overlaps
    // the expansion of the next chunk (memory-bound next to latency-bound)
    void finish_materialise(uint64_t chunk_base, uint64_t ncols, unsigned parity, hipStream_t expanded_on = nullptr) {
        if (pairs_only()) return;
        const unsigned bx = (unsigned)((ncols + 255) / 256);
        unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
        // In-wave levels hand k_materialise only the OVERFLOW of the expand wavefronts' survivor lists (0.15 % of the contract workload's
        // states, 11 % of the five-server model's): the grid follows what the level before sent through the list (x 4, per chunk) instead
        // of the chunk's size — a grid-stride kernel: any grid is correct — so that a launch with next to nothing to do is 8 workgroups
        // that start and end at once, not 2048 that queue behind the next chunk's expand (round 5: 70 launches per step showed 62 ms of
        // HIP-event time for 0.8 M states: VERDICT weak 9)
        if (mat_inwave_level) {
            const uint64_t want = (mat_list_hint * 4 + 255) / 256;   // workgroups of 256 states, all eight segments together
            const unsigned per_seg = (unsigned)((want + NSHARD - 1) / NSHARD);
            if (per_seg < gm) gm = per_seg ? per_seg : 1u;
        }
        hipEventRecord(ev_e[parity], expanded_on ? expanded_on : stream);
        hipStreamWaitEvent(stream2, ev_e[parity], 0);
        timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise<S>, dim3(gm, NSHARD), dim3(256), 0, stream2, prm, d_arena, chunk_base, d_newlist,
                               seg_cap, arena_cap, d_parent, d_pslot, d_ctr, parity, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
        }, stream2);
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, parity);
        hipEventRecord(ev_m[parity], stream2);
        // measurement knob: TLAMC_SERIAL=1 lets no expand kernel run beside a materialise kernel (each kernel's stand-alone time)
        static const bool serial = getenv("TLAMC_SERIAL") != nullptr;
        if (serial) hipStreamSynchronize(stream2);
    }
    // one batched level (LevelCtl): the same pair of kernels, ranges read on the device, then the level is closed there
    void enqueue_blind_level(uint64_t max_states) {
        // all four kernels on ONE stream: for a frontier this small the two-stream overlap buys nothing and the
        // cross-stream events would cost more than the kernels
        const uint64_t ncols = max_states + 64;
        RouteArgs rt{};
        rt.lc = d_lc;
        rt.new_fp = d_newfp;
        set_inwave(rt);
        const unsigned sg = slices_for(ncols, true);
        const bool dl = sg > 1 && (cfg.flags & MC_F_DEADLOCK);
        if (dl) { rt.succ = d_nsl; hipMemsetAsync(d_nsl, 0, ncols * sizeof(uint16_t), stream); }
        const unsigned lflags = cfg.flags | ((cfg.flags & MC_F_PARK) ? MC_FI_PARK : 0u);   // (the PARK instantiation, see run())
        timed(0, 0, [&] {
            launch_expand<S, false>(!(cfg.flags & MC_F_NOFAMILY), lflags, ncols, stream, sg, prm, (const uint64_t *)d_arena,
                                    (uint64_t)0, (uint64_t)0, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr, cfg.flags, rt, 0u);
            if (dl) hipLaunchKernelGGL(k_deadlock_slices, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)d_nsl,
                                       (uint64_t)0, (uint64_t)0, ncols, (const LevelCtl *)d_lc, d_ctr);
        });
        if (!pairs_only()) timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise<S>, dim3(32, NSHARD), dim3(256), 0, stream, prm, d_arena, (uint64_t)0, d_newlist, seg_cap,
                               arena_cap, d_parent, d_pslot, d_ctr, 0u, (const LevelCtl *)d_lc, (const uint64_t *)d_newfp);
        });
        hipLaunchKernelGGL(k_end_level, dim3(1), dim3(1), 0, stream, d_ctr, d_lc);
    }

    // insert + materialise + commit for the candidate matrix just written
    template <bool INIT>
    void finish_chunk(uint64_t chunk_base_or_first, uint64_t ncols, unsigned rows) {
        const unsigned bx = (unsigned)((ncols + 255) / 256);
        timed(1, ncols * rows, [&] {
            hipLaunchKernelGGL(k_insert, dim3(bx, rows), dim3(256), 0, stream, d_cand, row_stride, ncols, d_nsl, d_table,
                               seen_arg(), d_newlist, d_ctr);
        });
        const unsigned gm = bx < 2048 ? bx : 2048;
        timed(2, 0, [&] {
            if (INIT)
                hipLaunchKernelGGL(k_init_materialise<S>, dim3(gm), dim3(256), 0, stream, prm, d_arena, chunk_base_or_first,
                                   d_inittmp, d_newlist, arena_cap, d_parent, d_pslot, d_ctr);
            else
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm, 1), dim3(256), 0, stream, prm, d_arena, chunk_base_or_first,
                                   d_newlist, seg_cap, arena_cap, d_parent, d_pslot, d_ctr, 0u, (const LevelCtl *)nullptr, (const uint64_t *)nullptr);
        });
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream, d_ctr, 0u);
    }

    int run(mc_result *out) override {
        memset(out, 0, sizeof *out);
        out->violated_invariant = -1;
        memset(kstat, 0, sizeof kstat);
        have_viol = false;
        stop_requested = false;
        have_run = false;  // set again on the success path only: a run that fails half-way (MC_EARENA, MC_ETABLEFULL, MC_EOVERFLOW)
                           // leaves fingerprints of an unfinished level in the seen-set — the next step / checkpoint must not continue it
        level_start.clear();
        HIP_TRY(hipSetDevice(cfg.device));
        const bool in_place = ck_pending && ck_in_place;  // mc_engine_step: the seen-set of the stopped run is still valid
        ck_in_place = false;
        if (!in_place) HIP_TRY(hipMemsetAsync(d_table, 0, table_cap * sizeof(uint64_t), stream));
        HIP_TRY(hipMemsetAsync(d_ctr, 0, sizeof(DevCounters), stream));
        DevCounters init_c;
        memset(&init_c, 0, sizeof init_c);
        init_c.viol_key = ~0ull;
        const bool resuming = ck_pending;
        ck_pending = false;
        if (resuming) {  // counters of the checkpointed run; its states are already in the arena (restore())
            init_c.arena_next = ck_distinct;
            init_c.generated[0].v = ck_generated;
            init_c.cells[0].v = ck_cells;
        }
        HIP_TRY(hipMemcpyAsync(d_ctr, &init_c, sizeof init_c, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        progress_last = t0;
        auto progress = [&](uint32_t lv, uint64_t lo_, uint64_t hi_) {
            if (!progress_fn) return;
            uint64_t g = 0;
            for (int t = 0; t < NSHARD; t++) g += h_ctr->generated[t].v;
            report_progress(lv, g, hi_, hi_ - lo_);
        };

        // level 1: Init
        const uint64_t ninit = resuming ? 0 : S::num_init(prm);
        if (resuming && ck_distinct && !in_place)
            hipLaunchKernelGGL(k_reseed_table<S>, dim3((unsigned)((ck_distinct + 255) / 256)), dim3(256), 0, stream, prm,
                               (const uint64_t *)d_arena, ck_distinct, d_table, seen_arg(), d_ctr);
        for (uint64_t first = 0; first < ninit; first += chunk) {
            const uint64_t count = ninit - first < chunk ? ninit - first : chunk;
            const uint64_t ncols = (count + 63) & ~63ull;
            hipLaunchKernelGGL(k_init_cand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, first, count,
                               d_inittmp, d_cand, ncols, d_nsl, d_ctr, 0u, 1u);  // run() is never owner-filtered
            finish_chunk<true>(first, ncols, 1);
        }
        // with in-wave writes every writer takes its arena indices from arena_next itself from here on (DevCounters::atomic_alloc).
        // Init above, the candidate-matrix form and every spec whose states all go through k_materialise keep appending in stream
        // order: one returning atomicAdd per wavefront of k_materialise on ONE word is pure cost there (atomic_add N = 28: 145
        // against 108 ms per run, profiles/r04h / r04i; raft through k_materialise only: 203 against 159 ms)
        if (inwave_ok()) hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, 1u);
        bool alloc_atomic = inwave_ok();
        uint64_t prev_frontier = ~0ull >> 12;  // (the first level is never "fast growing")
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        uint64_t lo = 0, hi = h_ctr->arena_next;
        uint32_t level = 1;
        out->level_distinct[0] = hi;
        level_start.push_back(0);
        if (resuming) {  // continue with the frontier [lo, hi) = the last, unexpanded level of the checkpointed run
            level_start = ck_level_start;
            level = (uint32_t)level_start.size();
            lo = ck_lo;
            for (uint32_t k = 0; k < level && k < MC_MAX_LEVELS; k++)
                out->level_distinct[k] = (k + 1 < level ? level_start[k + 1] : hi) - level_start[k];
        }
        int budget = 0;
        uint64_t via_seen = 0;
        // PARKING (round 6, MC_F_PARK): the instantiation of the by-family kernel whose writers park the overflow of their survivor lists and
        // write it themselves (k_expand_family<.., PARK>).  Measured (profiles/r06zb, zf, zg, zh): every state of the five-server model
        // is written in-wave then (924 M instead of 821 M; k_materialise idle) and its step takes as long as before, or longer — not the default
        const bool park_levels = (cfg.flags & MC_F_PARK) != 0;
        mat_list_hint = ~0ull >> 8;   // (unknown before the first large level: the full grid)
        mat_inwave_level = false;
        // (the largest frontier a batched level takes: 2^20 states since round 6 — 2^16 before; profiles/r06zm_blind_ab.jsonl: the contract
        //  workload 128.7 / 127.5 -> 126.1 / 126.0 ms, the generated pagecache model 7.65 / 7.63 -> 7.42 / 7.44 — $TLAMC_BLIND_LOG2 = A/B knob.
        //  Specs whose blind launches are sliced — the interpreter, the Paxos family: 32 slices of the level's CAPACITY — keep 2^16)
        static const int blind_log2 = getenv("TLAMC_BLIND_LOG2") ? atoi(getenv("TLAMC_BLIND_LOG2")) : 20;
        uint64_t blind_max = 1ull << (blind_log2 >= 10 && blind_log2 <= 23 && slices_for((1ull << 16) + 64, true) == 1 ? blind_log2 : 16);
        if (chunk < blind_max) blind_max = chunk;
        while (hi > lo) {
            if (h_ctr->viol_key != ~0ull) break;
            if (stop_frontier && hi - lo >= stop_frontier) break;  // the caller continues this level sharded
            if (cfg.max_levels && level >= cfg.max_levels) { budget = 1; break; }
            if (cfg.max_distinct && hi >= cfg.max_distinct) { budget = 1; break; }
            if (stop_requested) { budget = 1; break; }
            if (!use_matrix && !(cfg.flags & MC_F_NOBATCH) && hi - lo <= blind_max) {
                // Small frontier: BLIND_BATCH levels are enqueued back to back; the kernels take each level's range from
                // LevelCtl and k_end_level applies the same stopping rules as this loop, so one host round trip covers
                // up to BLIND_BATCH levels instead of one (a level of a few thousand states is pure launch latency).
                memset(h_lc, 0, sizeof *h_lc);
                h_lc->lo = lo;
                h_lc->hi = hi;
                h_lc->max_states = blind_max;
                h_lc->max_distinct = cfg.max_distinct;
                h_lc->levels_left = cfg.max_levels ? (unsigned)(cfg.max_levels - level) : 0u;
                HIP_TRY(hipMemcpyAsync(d_lc, h_lc, sizeof *h_lc, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream2));  // (nothing of an earlier level is still being materialised)
                if (inwave_ok() && !alloc_atomic) {  // the batched small levels always write in-wave
                    hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, 1u);
                    alloc_atomic = true;
                }
                for (int k = 0; k < BLIND_BATCH; k++) enqueue_blind_level(blind_max);
                HIP_TRY(hipMemcpyAsync(h_lc, d_lc, sizeof *h_lc, hipMemcpyDeviceToHost, stream));
                if ((rc = read_counters())) return rc;
                if ((rc = check_dev_error())) return rc;
                for (unsigned k = 0; k < h_lc->nlev; k++) {
                    kstat[0].units += hi - lo;  // states expanded by this level (the launches were timed with 0 units)
                    lo = hi;
                    hi = h_lc->level_hi[k];
                    if (hi > lo) {
                        level_start.push_back(lo);
                        if (level < MC_MAX_LEVELS) out->level_distinct[level] = hi - lo;
                        level++;
                    }
                    if (level >= MC_MAX_LEVELS) { set_error("too many BFS levels"); return MC_EBADCFG; }
                }
                progress(level, lo, hi);
                prev_frontier = hi > lo ? hi - lo : 1;
                continue;  // the loop head re-checks violation / budgets / frontier with the host's copies
            }
            // IN-WAVE WRITES, level by level (inwave_growth_limit above: a level that grows faster than the limit goes through the
            // new-list and k_materialise alone).  The level before this one says which kind it is; the allocation mode
            // (DevCounters::atomic_alloc) follows — nothing is in flight between two levels.
            const bool lvl_inwave = inwave_ok() && (pairs_only() || (double)(hi - lo) <= inwave_growth_limit * (double)prev_frontier);
            if (inwave_ok() && lvl_inwave != alloc_atomic) {
                hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, lvl_inwave ? 1u : 0u);
                alloc_atomic = lvl_inwave;
            }
            prev_frontier = hi - lo;
            mat_inwave_level = lvl_inwave;
            unsigned chunk_no = 0;
            // (odd chunks on their own stream — see stream_b; not with the slot-sliced launches' shared flag array, not for the matrix form)
            const bool two_streams = stream_b && !use_matrix && hi - lo > chunk && !(slices_for(chunk, false) > 1 && (cfg.flags & MC_F_DEADLOCK));
            if (two_streams) {  // what the level's first kernels on `stream` were ordered behind, stream_b's are too
                hipEventRecord(ev_level, stream);
                hipStreamWaitEvent(stream_b, ev_level, 0);
            }
            for (uint64_t c0 = lo; c0 < hi; ++chunk_no) {
                const unsigned parity = chunk_no & 1u;
                const uint64_t base = c0 & ~63ull;
                uint64_t c1 = base + chunk;  // chunk boundaries stay 64-aligned
                if (c1 > hi) c1 = hi;
                const uint64_t ncols = ((c1 - base) + 63) & ~63ull;
                if (use_matrix) {  // two-kernel form: sparse candidate matrix + k_insert (kept for A/B measurements)
                    timed(0, c1 - c0, [&] {
                        hipLaunchKernelGGL(k_expand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, d_arena,
                                           c0, c1, d_cand, row_stride, ncols, d_nsl, d_ctr, cfg.flags);
                    });
                    finish_chunk<false>(base, ncols, max_slots);
                } else {
                    hipStream_t es = two_streams && parity ? stream_b : stream;
                    if (chunk_no >= 2) hipStreamWaitEvent(es, ev_m[parity], 0);  // new-list `parity` is free again
                    RouteArgs rt_new{};
                    rt_new.new_fp = d_newfp;
                    if (lvl_inwave) set_inwave(rt_new);
                    const unsigned sg = slices_for(ncols, false);
                    const bool dl = sg > 1 && (cfg.flags & MC_F_DEADLOCK);
                    if (dl) { rt_new.succ = d_nsl; hipMemsetAsync(d_nsl, 0, ncols * sizeof(uint16_t), es); }
                    const unsigned lflags = cfg.flags | (lvl_inwave && park_levels ? MC_FI_PARK : 0u);
                    timed(0, c1 - c0, [&] {
                        launch_expand<S, false>(!(cfg.flags & MC_F_NOFAMILY), lflags, ncols, es, sg, prm,
                                                (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr,
                                                cfg.flags, rt_new, parity);
                        if (dl) hipLaunchKernelGGL(k_deadlock_slices, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, es,
                                                   (const uint16_t *)d_nsl, c0, c1, ncols, (const LevelCtl *)nullptr, d_ctr);
                    }, es);
                    finish_materialise(base, ncols, parity, es);
                }
                c0 = c1;
            }
            if ((rc = read_counters())) return rc;
            if ((rc = check_dev_error())) return rc;
            {   // what this level sent through the new-list, per chunk, scaled by the level's growth: the next level's k_materialise grid
                const uint64_t via = h_ctr->via_list - via_seen;
                via_seen = h_ctr->via_list;
                const double growth = hi > lo ? (double)(h_ctr->arena_next - hi) / (double)(hi - lo) : 1.0;
                mat_list_hint = (uint64_t)((double)(via / (chunk_no ? chunk_no : 1u)) * (growth > 1.0 ? growth : 1.0)) + 1;
            }
            lo = hi;
            hi = h_ctr->arena_next;
            if (hi > lo) {
                level_start.push_back(lo);
                if (level < MC_MAX_LEVELS) out->level_distinct[level] = hi - lo;
                level++;
            }
            if (level >= MC_MAX_LEVELS) { set_error("too many BFS levels"); return MC_EBADCFG; }
            progress(level, lo, hi);
        }
        // a run that stops with an unexpanded frontier (budget) has not evaluated that level's check-on-expand invariants yet
        if (hi > lo && h_ctr->viol_key == ~0ull && !stop_frontier && (rc = check_frontier(lo, hi))) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        out->seconds = std::chrono::duration<double>(t1 - t0).count();
        out->distinct = h_ctr->arena_next;
        last_distinct = h_ctr->arena_next;
        kstat[2].units = h_ctr->arena_next;
        out->generated = 0;
        kstat_cells = 0;
        for (int t = 0; t < NSHARD; t++) { out->generated += h_ctr->generated[t].v; kstat_cells += h_ctr->cells[t].v; }
        last_generated = out->generated;
        // (a resumed run counts from the checkpoint on: via_list starts at 0 there)
        kstat_inwave = inwave_ok() && h_ctr->arena_next >= h_ctr->via_list + (resuming ? ck_distinct : 0)
                           ? h_ctr->arena_next - h_ctr->via_list - (resuming ? ck_distinct : 0) : 0;  // (Init's states went through the list too)
        have_run = true;
        out->queue_left = hi - lo;
        out->depth = level;
        out->levels = level;
        run_lo = lo;
        run_hi = hi;
        if (h_ctr->viol_key != ~0ull) {
            have_viol = true;
            last_viol = h_ctr->viol_key;
            const unsigned kind = (unsigned)(last_viol & 7u);
            out->verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT
                         : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
            if (kind == VK_INVARIANT) out->violated_invariant = (int)(last_viol >> 3 & 31u);
            std::vector<uint64_t> chain;
            if (build_chain(chain) == MC_OK) {
                const unsigned slot = (unsigned)(last_viol >> 8 & 0xffffu);
                const bool extra = kind == VK_INVARIANT && slot < SLOT_PARENT;  // violating successor itself
                out->trace_len = (uint32_t)chain.size() + (extra ? 1u : 0u);
            }
        } else {
            out->verdict = budget ? MC_V_BUDGET : MC_V_OK;
        }
        return MC_OK;
    }
    uint64_t kstat_cells = 0, kstat_inwave = 0;
    uint64_t stop_frontier = 0, run_lo = 0, run_hi = 0;  // run() stops before a level of >= stop_frontier states (0 = never)

    // ------------------------------------------------------------------------------- checkpoint / recover
    // TLC checkpoints a run into its states/ directory and continues it with -recover (testout1:10: "-- Checkpointing of
    // run states/01-08-03-18-14-01 completed."; .gitignore:2).  Here the resident arena already IS the state store: a
    // checkpoint is the arena's used blocks as they lie in HBM, the level boundaries, the counters and (with MC_F_TRACE)
    // the parent pointers.  The seen-set is not written: recovery re-inserts word 0 of every state (k_reseed_table).
    struct CkHeader {
        char magic[8];
        uint32_t spec_id, nparams;
        int64_t params[16];
        uint32_t words, has_trace;
        uint64_t distinct, generated, cells, lo, hi, nlevels;
    };
    struct FileCloser {
        FILE *f;
        ~FileCloser() { if (f) fclose(f); }
    };
    bool have_run = false, ck_pending = false, ck_in_place = false;
    // mc_engine_step: `levels` more BFS levels.  The first call (or a call after a run that ended: complete, or with an error)
    // starts from Init; a call after a budget stop continues IN PLACE — arena, seen-set and parent pointers stay where they
    // are in HBM, only the counters and the level table are handed over (the same hand-over as restore(), without the file
    // and without rebuilding the seen-set).
    int step(uint32_t levels, mc_result *out) override {
        if (!levels) return MC_EBADCFG;
        const uint64_t saved = cfg.max_levels;
        const bool cont = have_run && !have_viol && run_hi > run_lo && !ck_pending;
        if (cont) {
            ck_distinct = last_distinct; ck_generated = last_generated; ck_cells = kstat_cells; ck_lo = run_lo;
            ck_level_start = level_start;
            ck_pending = ck_in_place = true;
            cfg.max_levels = (uint64_t)level_start.size() + levels;
        } else if (ck_pending) {  // restore() handed a checkpoint over: `levels` more levels beyond the checkpointed ones
            cfg.max_levels = (uint64_t)ck_level_start.size() + levels;
        } else {
            cfg.max_levels = levels;
        }
        const int rc = run(out);
        cfg.max_levels = saved;
        if (rc) ck_pending = ck_in_place = false;  // after a failure the next step starts over
        return rc;
    }
    uint64_t last_generated = 0, ck_distinct = 0, ck_generated = 0, ck_cells = 0, ck_lo = 0;
    std::vector<uint64_t> ck_level_start;
    bool ck_params_comparable() const { return desc.spec_id != MC_SPEC_PCAL; }  // a compiled program's parameter is a host pointer
    int dev_to_file(const void *dev, size_t bytes, FILE *f) {
        std::vector<char> buf(bytes < (64u << 20) ? bytes : (64u << 20));
        for (size_t off = 0; off < bytes; off += buf.size()) {
            const size_t n = bytes - off < buf.size() ? bytes - off : buf.size();
            HIP_TRY(hipMemcpy(buf.data(), (const char *)dev + off, n, hipMemcpyDeviceToHost));
            if (fwrite(buf.data(), 1, n, f) != n) { set_error("checkpoint: short write"); return MC_EBADCFG; }
        }
        return MC_OK;
    }
    int file_to_dev(void *dev, size_t bytes, FILE *f) {
        std::vector<char> buf(bytes < (64u << 20) ? bytes : (64u << 20));
        for (size_t off = 0; off < bytes; off += buf.size()) {
            const size_t n = bytes - off < buf.size() ? bytes - off : buf.size();
            if (fread(buf.data(), 1, n, f) != n) { set_error("restore: the checkpoint file is truncated"); return MC_EPARSE; }
            HIP_TRY(hipMemcpy((char *)dev + off, buf.data(), n, hipMemcpyHostToDevice));
        }
        return MC_OK;
    }
    int checkpoint(const char *path) override {
        if (cfg.shard_count > 1) { set_error("checkpoint: not available for a sharded engine"); return MC_EBADCFG; }
        if (!have_run || have_viol) { set_error("checkpoint: needs a completed mc_engine_run that stopped on a budget (or finished) without a violation"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        FILE *f = fopen(path, "wb");
        if (!f) { set_error(std::string("checkpoint: cannot write ") + path); return MC_EBADCFG; }
        FileCloser closer{f};  // early returns (HIP errors) must not leak the handle
        CkHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, "TLAMCCK1", 8);
        h.spec_id = desc.spec_id;
        h.nparams = ck_params_comparable() ? desc.nparams : 1;
        for (uint32_t i = 0; i < h.nparams && i < 16; i++) h.params[i] = desc.params[i];
        if (!ck_params_comparable()) h.params[0] = (int64_t)program_hash;  // a compiled program is identified by its image
        h.words = (uint32_t)W;
        h.has_trace = d_parent ? 1u : 0u;
        h.distinct = last_distinct; h.generated = last_generated; h.cells = kstat_cells;
        h.lo = run_lo; h.hi = run_hi; h.nlevels = level_start.size();
        int rc = MC_OK;
        if (fwrite(&h, sizeof h, 1, f) != 1 || fwrite(level_start.data(), sizeof(uint64_t), level_start.size(), f) != level_start.size()) {
            set_error("checkpoint: short write");
            rc = MC_EBADCFG;
        }
        const size_t blocks = (size_t)((last_distinct + 63) >> 6);
        if (!rc) rc = dev_to_file(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_parent, (size_t)last_distinct * sizeof(uint32_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_pslot, (size_t)last_distinct * sizeof(uint16_t), f);
        closer.f = nullptr;
        if (fclose(f) != 0 && !rc) { set_error("checkpoint: close failed"); rc = MC_EBADCFG; }
        return rc;
    }
    int restore(const char *path) override {
        if (cfg.shard_count > 1) { set_error("restore: not available for a sharded engine"); return MC_EBADCFG; }
        FILE *f = fopen(path, "rb");
        if (!f) { set_error(std::string("restore: cannot read ") + path); return MC_EPARSE; }
        FileCloser closer{f};
        CkHeader h;
        int rc = MC_OK;
        auto fail = [&](int code, const char *msg) { set_error(msg); rc = code; };
        if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "TLAMCCK1", 8) != 0) fail(MC_EPARSE, "restore: not a checkpoint file");
        else if (h.spec_id != desc.spec_id || h.words != (uint32_t)W) fail(MC_EBADCFG, "restore: the checkpoint belongs to another spec");
        else if (ck_params_comparable() && (h.nparams != desc.nparams || memcmp(h.params, desc.params, sizeof(int64_t) * (h.nparams < 16 ? h.nparams : 16)) != 0))
            fail(MC_EBADCFG, "restore: the checkpoint was written with other constants / invariants");
        else if (!ck_params_comparable() && (h.nparams != 1 || (uint64_t)h.params[0] != program_hash))
            fail(MC_EBADCFG, "restore: the checkpoint was written by another compiled program (algorithm, constants, invariants or constraints differ)");
        // a run may legitimately stop on a budget at a seen-set load of 0.5 - 0.8 (the bench model at 2^27 slots ends at 0.76): its
        // checkpoint must restore into the same configuration; beyond 0.9 the reseeding itself could not finish (DEV_ETABLE)
        else if (h.distinct * 10 > table_cap * 9) fail(MC_ETABLEFULL, "restore: table_capacity cannot hold the checkpoint's states (load > 0.9)");
        else if (h.distinct > arena_cap) fail(MC_EARENA, "restore: arena_capacity is smaller than the checkpoint");
        else if (h.hi != h.distinct || h.lo > h.hi || h.nlevels == 0 || h.nlevels >= MC_MAX_LEVELS) fail(MC_EPARSE, "restore: inconsistent header");
        else if (d_parent && !h.has_trace) fail(MC_EBADCFG, "restore: the checkpoint holds no parent pointers (written without MC_F_TRACE); run without MC_F_TRACE");
        if (!rc) {
            ck_level_start.assign((size_t)h.nlevels, 0);
            if (fread(ck_level_start.data(), sizeof(uint64_t), ck_level_start.size(), f) != ck_level_start.size()) fail(MC_EPARSE, "restore: the checkpoint file is truncated");
            else {  // level table: starts at 0, strictly increasing, below the fill level; an unexpanded frontier IS the last level
                bool ok = ck_level_start[0] == 0 && ck_level_start.back() < h.hi && (h.lo == h.hi || ck_level_start.back() == h.lo);
                for (size_t k = 1; k < ck_level_start.size() && ok; k++) ok = ck_level_start[k] > ck_level_start[k - 1];
                if (!ok) fail(MC_EPARSE, "restore: inconsistent level table");
            }
        }
        if (!rc) {
            HIP_TRY(hipSetDevice(cfg.device));
            const size_t blocks = (size_t)((h.distinct + 63) >> 6);
            rc = file_to_dev(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
            if (!rc && d_parent && h.has_trace) rc = file_to_dev(d_parent, (size_t)h.distinct * sizeof(uint32_t), f);
            if (!rc && d_parent && h.has_trace) rc = file_to_dev(d_pslot, (size_t)h.distinct * sizeof(uint16_t), f);
        }
        if (rc) return rc;
        ck_distinct = h.distinct; ck_generated = h.generated; ck_cells = h.cells; ck_lo = h.lo;
        ck_pending = true;  // the next run() continues from here
        return MC_OK;
    }

    int fetch_state(uint64_t idx, uint64_t *words) {
        const uint64_t *src = d_arena + ((idx >> 6) * (uint64_t)W) * 64 + (idx & 63);
        HIP_TRY(hipMemcpy2D(words, sizeof(uint64_t), src, 64 * sizeof(uint64_t), sizeof(uint64_t), W, hipMemcpyDeviceToHost));
        return MC_OK;
    }
    // arena indices from an initial state to the state the violation was found in / from
    int build_chain(std::vector<uint64_t> &chain) {
        chain.clear();
        if (!have_viol) return MC_ESTATE;
        const unsigned slot = (unsigned)(last_viol >> 8 & 0xffffu);
        if (slot == SLOT_INIT) return MC_OK;  // violated by an initial state: handled by the caller
        if (!d_parent) { set_error("engine created without MC_F_TRACE"); return MC_ESTATE; }
        uint64_t idx = viol_idx(last_viol);
        for (int guard = 0; guard < MC_MAX_LEVELS; ++guard) {
            chain.push_back(idx);
            uint32_t p;
            HIP_TRY(hipMemcpy(&p, d_parent + idx, sizeof p, hipMemcpyDeviceToHost));
            if (p == 0xffffffffu) break;
            idx = p;
        }
        std::vector<uint64_t> rev(chain.rbegin(), chain.rend());
        chain.swap(rev);
        return MC_OK;
    }
    int trace(uint8_t *states_out, int32_t *actions_out, size_t *n_inout) override {
        if (!have_viol) { *n_inout = 0; return MC_OK; }
        HIP_TRY(hipSetDevice(cfg.device));
        const unsigned kind = (unsigned)(last_viol & 7u), slot = (unsigned)(last_viol >> 8 & 0xffffu);
        std::vector<uint64_t> words;
        std::vector<int32_t> acts;
        if (slot == SLOT_INIT) {  // an initial state violates an invariant
            words.resize(W);
            S::init(prm, viol_idx(last_viol), WordRef{words.data(), 1});
            acts.push_back(-1);
        } else {
            std::vector<uint64_t> chain;
            int rc = build_chain(chain);
            if (rc) return rc;
            words.resize(chain.size() * W);
            for (size_t k = 0; k < chain.size(); ++k) {
                if ((rc = fetch_state(chain[k], &words[k * W]))) return rc;
                if (k == 0) acts.push_back(-1);
                else {
                    uint16_t ps;
                    HIP_TRY(hipMemcpy(&ps, d_pslot + chain[k], sizeof ps, hipMemcpyDeviceToHost));
                    acts.push_back(S::action_of(prm, &words[(k - 1) * W], (int)ps));
                }
            }
            if (kind == VK_INVARIANT && slot < SLOT_PARENT) {  // append the violating successor
                std::vector<uint64_t> last(words.end() - W, words.end());
                words.resize(words.size() + W);
                S::apply(prm, CWordRef{last.data(), 1}, (int)slot, WordRef{&words[words.size() - W], 1});
                acts.push_back(S::action_of(prm, last.data(), (int)slot));
            }
        }
        const size_t n = acts.size();
        if (n > *n_inout) { *n_inout = n; set_error("trace buffer too small"); return MC_EBADCFG; }
        if constexpr (HasExport<S>::value) {  // rows leave the engine in the spec's PUBLIC layout (generated code: the interpreter's row)
            for (size_t k = 0; k < n; ++k) S::export_row(prm, &words[k * W], (uint64_t *)states_out + k * (size_t)S::EXPORT_WORDS);
        } else
        memcpy(states_out, words.data(), n * W * sizeof(uint64_t));
        memcpy(actions_out, acts.data(), n * sizeof(int32_t));
        *n_inout = n;
        return MC_OK;
    }
    int read_states(uint64_t first, uint64_t count, uint8_t *out) override {
        if (!count) return MC_OK;
        if (first + count > last_distinct) { set_error("read_states: range beyond the states found by the last run"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        uint64_t *tmp = nullptr;
        const uint64_t piece = 1ull << 20;
        HIP_TRY(hipMalloc(&tmp, (count < piece ? count : piece) * W * sizeof(uint64_t)));
        for (uint64_t off = 0; off < count; off += piece) {
            const uint64_t n = count - off < piece ? count - off : piece;
            hipLaunchKernelGGL(k_gather_states, dim3((unsigned)((n * W + 255) / 256)), dim3(256), 0, stream, d_arena, W, first + off, n, tmp);
            hipError_t e1, e2;
            if constexpr (HasExport<S>::value) {
                std::vector<uint64_t> rows((size_t)n * W);
                e1 = hipMemcpyAsync(rows.data(), tmp, n * W * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
                e2 = hipStreamSynchronize(stream);
                if (e1 == hipSuccess && e2 == hipSuccess)
                    for (uint64_t k = 0; k < n; ++k) S::export_row(prm, &rows[(size_t)k * W], (uint64_t *)out + (size_t)(off + k) * (size_t)S::EXPORT_WORDS);
            } else {
                e1 = hipMemcpyAsync(out + off * W * sizeof(uint64_t), tmp, n * W * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
                e2 = hipStreamSynchronize(stream);
            }
            if (e1 != hipSuccess || e2 != hipSuccess) { hipFree(tmp); set_error("read_states: copy failed"); return MC_EHIP; }
        }
        hipFree(tmp);
        return MC_OK;
    }
    uint64_t last_distinct = 0;
    // Profiling aid: expand every resident state again (seen-set already full, so every probe hits)
    // with optional ablation flags; returns the kernel time.  State counts are not changed.
    int debug_flags(uint32_t set, uint32_t clear) override { cfg.flags = (cfg.flags & ~clear) | set; return MC_OK; }
    int debug_phases(uint64_t *out48, int reset) override {
#ifdef MC_PHASE_PROF
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_phase), 48 * sizeof(uint64_t)));
        if (reset) { static const unsigned long long zero[48] = {}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero)); }
        return MC_OK;
#else
        (void)out48; (void)reset;
        set_error("the library was built without MC_PHASE_PROF");
        return MC_ESTATE;
#endif
    }
    int debug_reexpand(unsigned extra_flags, double *ms) override {
        HIP_TRY(hipSetDevice(cfg.device));
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, stream));
        for (uint64_t c0 = 0; c0 < last_distinct; c0 += chunk) {
            const uint64_t c1 = c0 + chunk < last_distinct ? c0 + chunk : last_distinct;
            const uint64_t ncols = ((c1 - c0) + 63) & ~63ull;
            launch_expand<S, false>(!((cfg.flags | extra_flags) & MC_F_NOFAMILY), cfg.flags | extra_flags, ncols, stream, 0u, prm,
                                    (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr,
                                    cfg.flags | extra_flags, RouteArgs{}, 0u);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream, d_ctr, 0u);
        }
        HIP_TRY(hipEventRecord(b, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, a, b));
        *ms = t;
        hipEventDestroy(a);
        hipEventDestroy(b);
        // restore arena_next (k_commit added nothing: no probe can be new) and leave counters as they were
        return MC_OK;
    }
    // ------------------------------------------------------------------ sharded step API
    uint64_t sh_lo = 0, sh_hi = 0, sh_next = 0;  // local frontier [sh_lo, sh_hi), arena fill level
    // Two expand slots: the route-mode expand of chunk r+1 (on `stream`) overlaps the exchange, probe and keep /
    // materialise of chunk r (kernels on `stream2`, collectives on the caller's stream).
    struct ShSlot {
        uint64_t *rt_fp = nullptr;
        uint32_t *rt_src = nullptr, *pend_src = nullptr;
        PaddedCounter *rt_cur = nullptr;  // [nranks*NSHARD] route cursors
        uint64_t rt_subcap = 0, pend_cap = 0, pend_total = 0, chunk_base = 0, count = 0, ncols = 0;
        BlockPlan plan;          // of the slot's last shard_materialise (for shard_materialise_parents)
        uint64_t moved = 0;
        OwnerOffsets pend_off;
        bool launched = false, keep_pending = false;
    } sl[2];
    uint32_t *d_new_src = nullptr, *d_incl = nullptr;
    uint64_t new_cap = 0;
    unsigned long long *d_ends = nullptr, *h_ends = nullptr;
    PaddedCounter *h_cur = nullptr;  // pinned copy of one slot's route cursors
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    unsigned nranks() const { return cfg.shard_count > 1 ? cfg.shard_count : 1; }

    int shard_alloc(ShSlot &q, uint64_t send_cap) {
        const unsigned P = nranks();
        if (send_cap > q.pend_cap) {
            if (q.rt_fp) { hipFree(q.rt_fp); hipFree(q.rt_src); hipFree(q.pend_src); q.rt_fp = nullptr; }
            q.rt_subcap = (send_cap / (P * NSHARD)) * 2 + 4096;
            HIP_TRY(hipMalloc(&q.rt_fp, (size_t)P * NSHARD * q.rt_subcap * sizeof(uint64_t)));
            HIP_TRY(hipMalloc(&q.rt_src, (size_t)P * NSHARD * q.rt_subcap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&q.pend_src, send_cap * sizeof(uint32_t)));
            q.pend_cap = send_cap;
        }
        if (!q.rt_cur) HIP_TRY(hipMalloc(&q.rt_cur, (size_t)(P * NSHARD) * sizeof(PaddedCounter)));
        if (send_cap > new_cap) {
            if (d_new_src) { hipFree(d_new_src); hipFree(d_incl); d_new_src = nullptr; }
            HIP_TRY(hipMalloc(&d_new_src, send_cap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&d_incl, send_cap * sizeof(uint32_t)));
            new_cap = send_cap;
        }
        if (!d_ends) {
            HIP_TRY(hipMalloc(&d_ends, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipHostMalloc(&h_ends, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipHostMalloc(&h_cur, (size_t)8 * NSHARD * sizeof(PaddedCounter)));
        }
        return MC_OK;
    }
    int scan_answers(const uint8_t *answers_back, uint64_t n, hipStream_t on) {
        hipcub::TransformInputIterator<uint32_t, AnswerCast, const uint8_t *> in(answers_back, AnswerCast());
        size_t need = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need, in, d_incl, (int)n, on));
        if (need > scan_tmp_bytes) {
            if (d_scan_tmp) hipFree(d_scan_tmp);
            HIP_TRY(hipMalloc(&d_scan_tmp, need));
            scan_tmp_bytes = need;
        }
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(d_scan_tmp, need, in, d_incl, (int)n, on));
        return MC_OK;
    }
    // a sharded run that failed mid-level (a full route bucket, MC_ETABLEFULL ...) leaves its slots as they were: the remedy the
    // error message names — run the same engine again with a larger allowance — must not trip over an expand "still in flight"
    void shard_reset_slots() {
        hipSetDevice(cfg.device);
        hipDeviceSynchronize();  // (whatever the failed run left on the streams has drained; its events are complete)
        for (auto &q : sl) { q.launched = false; q.keep_pending = false; q.pend_total = 0; q.moved = 0; q.count = 0; }
    }
    int shard_begin() override {
        if constexpr (PackedRows<S>::value) { set_error("this engine of generated code stores rows packed to its program's cell ranges and serves ONE GPU: create a sharded engine with shard_count > 1 (or $TLAMC_JIT_PACK=0)"); return MC_EBADCFG; }
        if (nranks() > 8) { set_error("at most 8 shards"); return MC_EBADCFG; }
        sh_dup = 0;
        sh_resume = sh_ck_ok = false;
        shard_reset_slots();
        HIP_TRY(hipSetDevice(cfg.device));
        memset(kstat, 0, sizeof kstat);
        HIP_TRY(hipMemsetAsync(d_table, 0, table_cap * sizeof(uint64_t), stream));
        DevCounters init_c;
        memset(&init_c, 0, sizeof init_c);
        init_c.viol_key = ~0ull;
        HIP_TRY(hipMemcpyAsync(d_ctr, &init_c, sizeof init_c, hipMemcpyHostToDevice, stream));
        const uint64_t ninit = S::num_init(prm);
        for (uint64_t first = 0; first < ninit; first += chunk) {
            const uint64_t count = ninit - first < chunk ? ninit - first : chunk;
            const uint64_t ncols = (count + 63) & ~63ull;
            hipLaunchKernelGGL(k_init_cand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, first, count,
                               d_inittmp, d_cand, ncols, d_nsl, d_ctr, cfg.shard_rank, cfg.shard_count);
            finish_chunk<true>(first, ncols, 1);
        }
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        sh_lo = 0;
        sh_hi = sh_next = h_ctr->arena_next;
        last_distinct = sh_next;
        return MC_OK;
    }
    // The first levels of a run are tiny: sharding them costs several collectives per level and balances nothing.
    // Every rank therefore runs the SAME fused BFS (run()) until a level has at least min_frontier states, keeps every
    // nranks-th state of that level as its local frontier, and the sharded rounds start there.  The seen-set of each
    // rank then holds ALL prefix fingerprints (a superset of the ones it owns: harmless); rank 0 alone reports the
    // prefix's `generated`, and distinct_local excludes what other ranks already count.
    uint64_t sh_dup = 0;
    std::unique_ptr<mc_result> prefix_res{new mc_result()};
    int shard_begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) override {
        if constexpr (PackedRows<S>::value) { set_error("this engine of generated code stores rows packed to its program's cell ranges and serves ONE GPU: create a sharded engine with shard_count > 1 (or $TLAMC_JIT_PACK=0)"); return MC_EBADCFG; }
        if (nranks() > 8) { set_error("at most 8 shards"); return MC_EBADCFG; }
        mc_result &res = *prefix_res;  // large (level table): not on the stack, and not shared between engines / threads
        sh_resume = sh_ck_ok = false;
        shard_reset_slots();
        const uint64_t saved_md = cfg.max_distinct, saved_ml = cfg.max_levels;
        cfg.max_distinct = max_distinct;  // the whole job's budgets: the prefix stops where the single-GPU run would
        cfg.max_levels = max_levels;
        stop_frontier = min_frontier ? min_frontier : 1;
        const int rc = run(&res);
        stop_frontier = 0;
        cfg.max_distinct = saved_md;
        cfg.max_levels = saved_ml;
        if (rc) return rc;
        const uint32_t cap = *nlevels;
        *nlevels = res.levels;
        for (uint32_t k = 0; k < res.levels && k < cap; k++) levels_out[k] = res.level_distinct[k];
        if (res.levels > cap) { set_error("shard_begin_replicated: level buffer too small"); return MC_EBADCFG; }
        const unsigned P = nranks(), r = cfg.shard_rank;
        const uint64_t lo = run_lo, hi = run_hi;
        const bool go_on = h_ctr->viol_key == ~0ull && hi > lo;  // otherwise finished or failed inside the prefix
        if (go_on)
            hipLaunchKernelGGL(k_take_owned<S>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, stream, prm, d_arena, lo, hi, r, P, hi,
                               arena_cap, d_parent, d_pslot, d_ctr);
        hipLaunchKernelGGL(k_after_prefix, dim3(1), dim3(1), 0, stream, d_ctr, (unsigned long long)hi, r != 0 ? 1 : 0);
        int rc2 = read_counters();
        if (rc2) return rc2;
        if ((rc2 = check_dev_error())) return rc2;
        sh_lo = hi;
        sh_hi = sh_next = h_ctr->arena_next;
        sh_dup = r == 0 ? sh_hi - hi : sh_hi;
        last_distinct = sh_next;
        return MC_OK;
    }
    int shard_level_size(uint64_t *n) override { *n = sh_hi - sh_lo; return MC_OK; }
    // ------------------------------------------------------------------ one checkpoint file per rank (testout1:10)
    // A rank's share of a sharded run cannot be rebuilt from its arena alone: the states a rank HOLDS are the ones it generated
    // (stay levels) or was sent (move levels), the fingerprints its seen-set slice holds are the ones it OWNS.  So a rank's file is
    // its arena, its parent pointers (index, slot, rank), its seen-set slice as it lies in HBM, its counters, and the level table of
    // the whole job (the same on every rank; the level loop hands it over when a run ends: shard_note_levels).  Restoring needs
    // the same world size, rank, table_capacity and spec; the next mc_shard_run* then continues with the unexpanded frontier.
    struct ShCkHeader {
        char magic[8];
        uint32_t spec_id, nparams;
        int64_t params[16];
        uint32_t words, has_trace, rank, world;
        uint64_t table_cap, lo, hi, next, dup, nlevels;
    };
    // the counters a per-rank checkpoint carries, field by field (the file held a raw DevCounters, whose layout —
    // padding, alignment, members — changes with the kernels; "TLAMCSK1" files are refused as "not a per-rank checkpoint file")
    struct ShCkCounters {
        uint64_t arena_next, viol_key, via_list;
        uint64_t generated[NSHARD], cells[NSHARD];
        uint32_t error, pad;
    };
    std::vector<uint64_t> sh_levels;
    bool sh_ck_ok = false, sh_resume = false;
    int shard_note_levels(const uint64_t *levels, uint32_t n, int32_t verdict) override {
        sh_levels.assign(levels, levels + n);
        sh_ck_ok = verdict == MC_V_OK || verdict == MC_V_BUDGET;
        return MC_OK;
    }
    int shard_resume(uint64_t *levels_out, uint32_t *nlevels) override {
        const uint32_t cap = *nlevels;
        *nlevels = 0;
        if (!sh_resume) return MC_OK;
        if (sh_levels.size() > cap) { set_error("shard_resume: level buffer too small"); return MC_EBADCFG; }
        for (size_t k = 0; k < sh_levels.size(); k++) levels_out[k] = sh_levels[k];
        *nlevels = (uint32_t)sh_levels.size();
        sh_resume = false;
        return MC_OK;
    }
    int shard_checkpoint(const char *path) override {
        if (!sh_ck_ok || sh_levels.empty()) { set_error("mc_shard_checkpoint: needs a sharded run that ended (finished, or stopped on a budget) without an error"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        FILE *f = fopen(path, "wb");
        if (!f) { set_error(std::string("mc_shard_checkpoint: cannot write ") + path); return MC_EBADCFG; }
        FileCloser closer{f};
        ShCkHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, "TLAMCSK2", 8);
        h.spec_id = desc.spec_id;
        h.nparams = ck_params_comparable() ? desc.nparams : 1;
        for (uint32_t i = 0; i < h.nparams && i < 16; i++) h.params[i] = desc.params[i];
        if (!ck_params_comparable()) h.params[0] = (int64_t)program_hash;
        h.words = (uint32_t)W; h.has_trace = d_parent ? 1u : 0u; h.rank = cfg.shard_rank; h.world = nranks();
        h.table_cap = table_cap | (seen_sparse ? SEEN_SPARSE : 0); h.lo = sh_lo; h.hi = sh_hi; h.next = sh_next; h.dup = sh_dup; h.nlevels = sh_levels.size();
        int rc = MC_OK;
        DevCounters dc;
        HIP_TRY(hipMemcpy(&dc, d_ctr, sizeof dc, hipMemcpyDeviceToHost));
        ShCkCounters c;
        memset(&c, 0, sizeof c);
        c.arena_next = dc.arena_next; c.viol_key = dc.viol_key; c.via_list = dc.via_list; c.error = dc.error;
        for (int t = 0; t < NSHARD; t++) { c.generated[t] = dc.generated[t].v; c.cells[t] = dc.cells[t].v; }
        if (fwrite(&h, sizeof h, 1, f) != 1 || fwrite(sh_levels.data(), sizeof(uint64_t), sh_levels.size(), f) != sh_levels.size() || fwrite(&c, sizeof c, 1, f) != 1) {
            set_error("mc_shard_checkpoint: short write");
            rc = MC_EBADCFG;
        }
        const size_t blocks = (size_t)((sh_next + 63) >> 6);
        if (!rc) rc = dev_to_file(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_parent, (size_t)sh_next * sizeof(uint32_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_pslot, (size_t)sh_next * sizeof(uint16_t), f);
        if (!rc && d_prank) rc = dev_to_file(d_prank, (size_t)sh_next, f);
        if (!rc) rc = dev_to_file(d_table, (size_t)table_cap * sizeof(uint64_t), f);
        closer.f = nullptr;
        if (fclose(f) != 0 && !rc) { set_error("mc_shard_checkpoint: close failed"); rc = MC_EBADCFG; }
        return rc;
    }
    int shard_restore(const char *path) override {
        if constexpr (PackedRows<S>::value) { set_error("this engine of generated code stores rows packed to its program's cell ranges and serves ONE GPU: create a sharded engine with shard_count > 1 (or $TLAMC_JIT_PACK=0)"); return MC_EBADCFG; }
        FILE *f = fopen(path, "rb");
        if (!f) { set_error(std::string("mc_shard_restore: cannot read ") + path); return MC_EPARSE; }
        FileCloser closer{f};
        ShCkHeader h;
        int rc = MC_OK;
        auto fail = [&](int code, const char *msg) { set_error(msg); rc = code; };
        if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "TLAMCSK2", 8) != 0) fail(MC_EPARSE, "mc_shard_restore: not a per-rank checkpoint file (of this library version)");
        else if (h.spec_id != desc.spec_id || h.words != (uint32_t)W) fail(MC_EBADCFG, "mc_shard_restore: the checkpoint belongs to another spec");
        else if (ck_params_comparable() && (h.nparams != desc.nparams || memcmp(h.params, desc.params, sizeof(int64_t) * (h.nparams < 16 ? h.nparams : 16)) != 0))
            fail(MC_EBADCFG, "mc_shard_restore: the checkpoint was written with other constants / invariants");
        else if (!ck_params_comparable() && (h.nparams != 1 || (uint64_t)h.params[0] != program_hash))
            fail(MC_EBADCFG, "mc_shard_restore: the checkpoint was written by another compiled program");
        else if (h.rank != cfg.shard_rank || h.world != nranks()) fail(MC_EBADCFG, "mc_shard_restore: the file is another rank's, or of a run with another number of ranks");
        else if (h.table_cap != (table_cap | (seen_sparse ? SEEN_SPARSE : 0)))
            fail(MC_EBADCFG, "mc_shard_restore: table_capacity differs from the checkpointed run's, or the bucket form it implies (4 slots when the table is >= 3 x the arena): the seen-set slice is stored as it lay in HBM");
        else if (h.next > arena_cap) fail(MC_EARENA, "mc_shard_restore: arena_capacity is smaller than the checkpoint");
        else if (h.lo > h.hi || h.hi != h.next || h.dup > h.next || h.nlevels == 0 || h.nlevels >= MC_MAX_LEVELS) fail(MC_EPARSE, "mc_shard_restore: inconsistent header");
        else if ((d_parent != nullptr) != (h.has_trace != 0)) fail(MC_EBADCFG, "mc_shard_restore: MC_F_TRACE differs from the checkpointed run's");
        ShCkCounters fc;
        memset(&fc, 0, sizeof fc);
        if (!rc) {
            sh_levels.assign((size_t)h.nlevels, 0);
            if (fread(sh_levels.data(), sizeof(uint64_t), sh_levels.size(), f) != sh_levels.size() || fread(&fc, sizeof fc, 1, f) != 1) fail(MC_EPARSE, "mc_shard_restore: the checkpoint file is truncated");
            else if (fc.arena_next != h.next || fc.viol_key != ~0ull || fc.error) fail(MC_EPARSE, "mc_shard_restore: inconsistent counters");
        }
        if (!rc) {
            HIP_TRY(hipSetDevice(cfg.device));
            HIP_TRY(hipDeviceSynchronize());
            const size_t blocks = (size_t)((h.next + 63) >> 6);
            rc = file_to_dev(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
            if (!rc && d_parent) rc = file_to_dev(d_parent, (size_t)h.next * sizeof(uint32_t), f);
            if (!rc && d_parent) rc = file_to_dev(d_pslot, (size_t)h.next * sizeof(uint16_t), f);
            if (!rc && d_prank) rc = file_to_dev(d_prank, (size_t)h.next, f);
            if (!rc) rc = file_to_dev(d_table, (size_t)table_cap * sizeof(uint64_t), f);
            if (!rc && fgetc(f) != EOF) fail(MC_EPARSE, "mc_shard_restore: trailing bytes (not the file this engine's configuration wrote)");
        }
        if (rc) { sh_levels.clear(); return rc; }
        DevCounters c;
        memset(&c, 0, sizeof c);
        c.arena_next = fc.arena_next; c.viol_key = fc.viol_key; c.via_list = fc.via_list;
        for (int t = 0; t < NSHARD; t++) { c.generated[t].v = fc.generated[t]; c.cells[t].v = fc.cells[t]; }
        HIP_TRY(hipMemcpy(d_ctr, &c, sizeof c, hipMemcpyHostToDevice));
        memset(kstat, 0, sizeof kstat);
        sh_lo = h.lo; sh_hi = h.hi; sh_next = h.next; sh_dup = h.dup;
        last_distinct = sh_next;
        have_viol = false;
        sl[0].launched = sl[1].launched = false;
        sh_resume = true;
        sh_ck_ok = false;
        return MC_OK;
    }
    int shard_info(void **main_stream, uint64_t *chunk_states, int32_t *traced) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (main_stream) *main_stream = (void *)stream;
        if (chunk_states) *chunk_states = chunk;
        if (traced) *traced = d_parent != nullptr;
        return MC_OK;
    }
    size_t state_bytes() const override { return (size_t)W * 8; }
    // Side stream of the sharded path.  By default the engine's own second stream, and every step call returns
    // with its work finished.  A caller that runs its collectives on a HIP stream hands that stream over with
    // shard_set_stream: compaction / probe / keep / materialise / ingest are then enqueued on it WITHOUT host
    // synchronisation (stream order ties them to the caller's all-to-alls), and only the calls that must return
    // counts to the host wait.
    hipStream_t ext_stream = nullptr;
    bool ext_side = false;
    hipEvent_t ev_slot[2] = {nullptr, nullptr}, ev_keep[2] = {nullptr, nullptr}, ev_ans = nullptr;
    hipStream_t side() const { return ext_side ? ext_stream : stream2; }
    // Three kinds of kernels append states at arena_next (the chunk's locally owned new states, keep, ingest) and they run on
    // different streams: one event chains them, each appender waits for the previous one and records when it is enqueued.
    hipEvent_t ev_append = nullptr;
    void append_begin(hipStream_t s) { if (ev_append) hipStreamWaitEvent(s, ev_append, 0); }
    void append_end(hipStream_t s) {
        if (!ev_append) hipEventCreateWithFlags(&ev_append, hipEventDisableTiming);
        hipEventRecord(ev_append, s);
    }
    int side_done() {
        if (!ext_side) HIP_TRY(hipStreamSynchronize(stream2));
        return MC_OK;
    }
    int shard_set_stream(void *hip_stream, int enable) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        ext_stream = (hipStream_t)hip_stream;
        ext_side = enable != 0;
        return MC_OK;
    }
    int shard_expand_launch(unsigned slot, uint64_t first, uint64_t count, uint64_t send_cap) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) { set_error("shard_expand: slot must be 0 or 1"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        const unsigned P = nranks();
        int rc = shard_alloc(q, send_cap);
        if (rc) return rc;
        q.pend_total = 0;
        q.count = count;
        q.launched = true;
        for (unsigned t = 0; t <= 8; t++) q.pend_off.off[t] = 0;
        if (first + count > sh_hi - sh_lo) { set_error("shard_expand: chunk outside the local frontier"); return MC_EBADCFG; }
        if (count > chunk) { set_error("shard_expand: chunk larger than chunk_states"); return MC_EBADCFG; }
        if (!count) return MC_OK;
        if (!ev_slot[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_slot[slot], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(stream, ev_slot[slot], 0));  // the slot's previous buckets have been compacted
        if (ev_mat[slot]) HIP_TRY(hipStreamWaitEvent(stream, ev_mat[slot], 0));  // ... and its locally owned new states written
        HIP_TRY(hipMemsetAsync(q.rt_cur, 0, (size_t)(P * NSHARD) * sizeof(PaddedCounter), stream));
        const uint64_t c0 = sh_lo + first, c1 = c0 + count, base = c0 & ~63ull;
        const uint64_t ncols = ((c1 - base) + 63) & ~63ull;
        q.chunk_base = base;
        RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
        rt.my_rank = cfg.shard_rank;
        rt.new_fp = d_newfp;
        q.ncols = ncols;
        timed(0, count, [&] {  // new-list parity = slot: the locally owned new states of this chunk (local-owner shortcut)
            launch_expand<S, true>(!(cfg.flags & MC_F_NOFAMILY), cfg.flags, ncols, stream, 0u, prm,
                                   (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr, cfg.flags, rt, slot);
        });
        return MC_OK;
    }
    int shard_expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1 || !sl[slot].launched) { set_error("shard_expand_finish: no expand in flight for this slot"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        q.launched = false;
        const unsigned P = nranks();
        for (unsigned t = 0; t < P; t++) send_counts[t] = 0;
        if (!q.count) return MC_OK;
        // one wait on the expand stream only: the side stream may still be busy with the previous round
        PaddedCounter *cur = h_cur;
        HIP_TRY(hipMemcpyAsync(cur, q.rt_cur, (size_t)(P * NSHARD) * sizeof(PaddedCounter), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int rc;
        if ((rc = check_dev_error())) return rc;
        uint64_t total = 0;
        for (unsigned t = 0; t < P; t++) {
            q.pend_off.off[t] = total;
            for (unsigned x = 0; x < NSHARD; x++) {
                if (cur[t * NSHARD + x].v > q.rt_subcap) { set_error("shard_expand: route bucket overflow (raise send_cap)"); return MC_EROUTE; }
                send_counts[t] += cur[t * NSHARD + x].v;
            }
            total += send_counts[t];
        }
        for (unsigned t = P; t <= 8; t++) q.pend_off.off[t] = total;
        // (the pending list is as long as the largest send capacity asked for so far: more candidates than it holds is "more successors
        //  per state than the fan-out allowance" — MC_EROUTE, the level loop starts over with twice the allowance — not a full arena)
        if (total >= (1ull << 31)) { set_error("shard_expand: more than 2^31 candidates in one round (lower chunk_states)"); return MC_EARENA; }
        if (total > q.pend_cap) { set_error("shard_expand: more candidates in one round than the slot's pending list holds (raise the fan-out allowance)"); return MC_EROUTE; }
        if (total > send_cap) { set_error("shard_expand: send buffer too small (raise the fan-out allowance)"); return MC_EROUTE; }
        q.pend_total = total;
        if (q.keep_pending) {  // the slot's previous keep still reads pend_src
            HIP_TRY(hipStreamWaitEvent(side(), ev_keep[slot], 0));
            q.keep_pending = false;
        }
        {   // the chunk's locally owned new states (local-owner shortcut): materialised on the side stream, in order with the
            // other kernels that append at arena_next (keep, ingest); the expand kernel has finished (stream was synchronised)
            const unsigned bx = (unsigned)((q.ncols + 255) / 256);
            const unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
            append_begin(stream2);
            timed(2, 0, [&] {
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm ? gm : 1, NSHARD), dim3(256), 0, stream2, prm, d_arena, q.chunk_base, d_newlist, seg_cap,
                                   arena_cap, d_parent, d_pslot, d_ctr, slot, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
            }, stream2);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, slot);
            append_end(stream2);
            if (!ev_mat[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_mat[slot], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_mat[slot], stream2));
        }
        if (total) {
            RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
            hipLaunchKernelGGL(k_compact_buckets, dim3(64, P * NSHARD), dim3(256), 0, side(), rt, send_fp, q.pend_src);
        }
        HIP_TRY(hipEventRecord(ev_slot[slot], side()));
        return side_done();
    }
    int shard_probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n) return MC_OK;
        timed(1, n, [&] {
            hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), recv_fp, n, d_table, seen_arg(), answers, d_ctr);
        }, side());
        return side_done();
    }
    // ---- fixed-capacity rounds: nothing of a round waits for the host (sizes travel in band; errors surface at shard_end_level)
    hipEvent_t ev_exp[2] = {nullptr, nullptr};
    // the slot's new-list segment and its counters (parity = slot) are read by the local materialise + commit on the second
    // stream: the NEXT expand into the same slot waits for them (a host that never blocks enqueues expands back to back)
    hipEvent_t ev_mat[2] = {nullptr, nullptr};
    int shard_expand_pack(unsigned slot, uint64_t *send_fp, uint64_t cap) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1 || !sl[slot].launched) { set_error("shard_expand_pack: no expand in flight for this slot"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        q.launched = false;
        const unsigned P = nranks();
        if (cap < 2 || (uint64_t)P * cap > q.pend_cap || (uint64_t)P * cap >= (1ull << 31)) { set_error("shard_expand_pack: capacity does not fit the slot's buffers"); return MC_EBADCFG; }
        q.pend_total = (uint64_t)P * cap;  // the sender scans the whole packed range: answers outside the counts are 0
        for (unsigned t = 0; t <= 8; t++) q.pend_off.off[t] = (t < P ? t : P) * cap;
        if (!ev_exp[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_exp[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_exp[slot], stream));  // behind the slot's expand kernel (or behind nothing: an empty chunk)
        // The buckets are compacted on the EXPAND stream, right behind the kernel that filled them: on the side stream the
        // compaction of round r+1 would queue behind the probes of round r (or those behind it), and the fingerprint exchange of
        // round r+1 could never overlap them.  The caller orders its collective behind this call with an event on that stream
        // (mc_shard_info hands it out).
        if (q.keep_pending) {  // the slot's previous keep still reads pend_src
            HIP_TRY(hipStreamWaitEvent(stream, ev_keep[slot], 0));
            q.keep_pending = false;
        }
        if (q.count) {  // the chunk's locally owned new states (local-owner shortcut), as in shard_expand_finish
            const unsigned bx = (unsigned)((q.ncols + 255) / 256);
            const unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
            HIP_TRY(hipStreamWaitEvent(stream2, ev_exp[slot], 0));
            append_begin(stream2);
            timed(2, 0, [&] {
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm ? gm : 1, NSHARD), dim3(256), 0, stream2, prm, d_arena, q.chunk_base, d_newlist, seg_cap,
                                   arena_cap, d_parent, d_pslot, d_ctr, slot, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
            }, stream2);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, slot);
            append_end(stream2);
            if (!ev_mat[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_mat[slot], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_mat[slot], stream2));
            RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
            hipLaunchKernelGGL(k_compact_packed, dim3(64, P * NSHARD), dim3(256), 0, stream, rt, cap, send_fp, q.pend_src, d_ctr);
        } else {  // no chunk for this rank in this round: empty buckets
            for (unsigned t = 0; t < P; t++) HIP_TRY(hipMemsetAsync(send_fp + (uint64_t)t * cap, 0, sizeof(uint64_t), stream));
        }
        if (!ev_slot[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_slot[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_slot[slot], stream));
        if (!ext_side) HIP_TRY(hipStreamSynchronize(stream));
        return side_done();
    }
    int shard_probe_pack(const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) override {
        HIP_TRY(hipSetDevice(cfg.device));
        const uint64_t total = (uint64_t)nranks() * cap;
        if (!total) return MC_OK;
        timed(1, total, [&] {
            hipLaunchKernelGGL(k_probe_packed, dim3((unsigned)(nranks() * ((cap + 255) / 256))), dim3(256), 0, side(), recv_fp, cap, nranks(), d_table,
                               seen_arg(), answers, d_ctr);
        }, side());
        return side_done();
    }
    int shard_keep_pack(unsigned slot, const uint8_t *answers_back, uint64_t cap) override {
        if (slot > 1 || sl[slot].pend_total != (uint64_t)nranks() * cap) { set_error("shard_keep_pack: not the capacity the slot was packed with"); return MC_EBADCFG; }
        return shard_keep(slot, answers_back, nullptr);
    }
    // the caller's stream waits (on the device) until the slot's last keep has consumed its answers buffer: a caller that
    // reuses ONE answers buffer per slot calls this before the collective that overwrites it
    int shard_wait_keep(unsigned slot) override {
        if (slot > 1) return MC_EBADCFG;
        HIP_TRY(hipSetDevice(cfg.device));
        if (sl[slot].keep_pending && ev_keep[slot]) HIP_TRY(hipStreamWaitEvent(side(), ev_keep[slot], 0));
        return MC_OK;
    }
    int shard_materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        const unsigned P = nranks();
        for (unsigned t = 0; t < P; t++) send_counts[t] = 0;
        q.moved = 0;
        if (!q.pend_total) return MC_OK;
        const unsigned bx = (unsigned)((q.pend_total + 255) / 256);
        int rc = scan_answers(answers_back, q.pend_total, side());
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_range_ends, dim3(1), dim3(64), 0, side(), d_incl, q.pend_off, P, d_ends);
        unsigned long long *ends = h_ends;  // the per-owner counts go back to the host: the caller sizes its all-to-all with them
        HIP_TRY(hipMemcpyAsync(ends, d_ends, P * sizeof(unsigned long long), hipMemcpyDeviceToHost, side()));
        HIP_TRY(hipStreamSynchronize(side()));
        OwnerOffsets start;
        std::vector<PaddedCounter> cnt(P);
        for (unsigned t = 0; t <= 8; t++) start.off[t] = 0;
        for (unsigned t = 0; t < P; t++) {
            start.off[t] = t ? ends[t - 1] : 0;
            cnt[t].v = ends[t] - start.off[t];
        }
        hipLaunchKernelGGL(k_compact_new, dim3(bx), dim3(256), 0, side(), answers_back, d_incl, q.pend_src, q.pend_total, q.pend_off, start, P,
                           d_new_src);
        BlockPlan plan;
        uint64_t blocks = 0;
        for (unsigned t = 0; t < 8; t++) {
            plan.blk_off[t] = blocks;
            plan.cnt[t] = t < P ? cnt[t].v : 0;
            if (t < P) { send_counts[t] = cnt[t].v; blocks += (cnt[t].v + 63) / 64; }
        }
        plan.blk_off[8] = blocks;
        for (unsigned t = P; t < 8; t++) plan.blk_off[t] = blocks;
        q.plan = plan;
        q.moved = 0;
        for (unsigned t = 0; t < P; t++) q.moved += plan.cnt[t];
        if (blocks * 64 > send_cap) { set_error("shard_materialise: state send buffer too small"); return MC_EARENA; }  // (the loop sizes it for the upper bound and repeats)
        if (blocks) {
            timed(2, blocks * 64, [&] {
                hipLaunchKernelGGL(k_send_materialise<S>, dim3((unsigned)((blocks * 64 + 255) / 256)), dim3(256), 0, side(), prm, d_arena,
                                   q.chunk_base, d_new_src, q.pend_off, plan, P, (uint64_t *)send_states);
            }, side());
        }
        return side_done();
    }
    // "stay" mode: materialise the positively answered candidates of slot's expand into the LOCAL arena.
    // The count of new states stays on the device (arena_next); the host learns it at shard_end_level.
    int shard_keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (n_new) *n_new = 0;
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        if (!q.pend_total) return MC_OK;
        const unsigned bx = (unsigned)((q.pend_total + 255) / 256);
        // keep runs on the engine's own second stream, behind the caller's stream at this point (the answers
        // are ready there), so the caller's next exchange does not queue behind the materialisation
        hipStream_t ks = stream2;
        if (ext_side) {
            if (!ev_ans) HIP_TRY(hipEventCreateWithFlags(&ev_ans, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_ans, ext_stream));
            HIP_TRY(hipStreamWaitEvent(ks, ev_ans, 0));
        }
        int rc = scan_answers(answers_back, q.pend_total, ks);
        if (rc) return rc;
        append_begin(ks);
        const uint32_t *d_total = d_incl + (q.pend_total - 1);
        OwnerOffsets one, zero;  // a single range covering every pending candidate
        for (unsigned t = 0; t <= 8; t++) { one.off[t] = t ? q.pend_total : 0; zero.off[t] = 0; }
        hipLaunchKernelGGL(k_compact_new, dim3(bx), dim3(256), 0, ks, answers_back, d_incl, q.pend_src, q.pend_total, one, zero, 1u, d_new_src);
        // grid sized for the upper bound (every candidate new); surplus threads leave at once
        timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise_list<S>, dim3(bx), dim3(256), 0, ks, prm, d_arena, q.chunk_base, d_new_src, d_total, arena_cap,
                               d_parent, d_pslot, d_ctr);
        }, ks);
        hipLaunchKernelGGL(k_bump_arena_next, dim3(1), dim3(1), 0, ks, d_ctr, d_total, 0ull, (unsigned long long)arena_cap);
        append_end(ks);
        if (!ev_keep[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_keep[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_keep[slot], ks));
        q.keep_pending = true;
        if (!ext_side) {
            uint32_t total32 = 0;
            HIP_TRY(hipMemcpyAsync(&total32, d_total, sizeof total32, hipMemcpyDeviceToHost, ks));
            HIP_TRY(hipStreamSynchronize(ks));
            if (n_new) *n_new = total32;
        }
        return MC_OK;
    }
    // recv_states: one bucket per source rank, back to back, each a whole number of 64-state blocks;
    // n = valid states of ONE bucket starting at recv_states (call once per source)
    int shard_ingest(const uint8_t *recv_states, uint64_t n) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n) return MC_OK;
        append_begin(side());
        hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), d_arena, W, (const uint64_t *)recv_states, n,
                           arena_cap, d_parent, d_ctr);
        hipLaunchKernelGGL(k_bump_arena_next, dim3(1), dim3(1), 0, side(), d_ctr, (const uint32_t *)nullptr, (unsigned long long)n,
                           (unsigned long long)arena_cap);
        append_end(side());
        return side_done();
    }
    // (parent index on this rank << 16 | slot) of the states the slot's last shard_materialise put into its send buffer, same
    // order (owner by owner, without the block padding); call it before the next shard_materialise / shard_keep of any slot
    int shard_materialise_parents(unsigned slot, uint64_t *send_parents) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        if (!q.moved) return MC_OK;
        hipLaunchKernelGGL(k_send_parents, dim3((unsigned)((q.moved + 255) / 256)), dim3(256), 0, side(), (const uint32_t *)d_new_src, q.chunk_base,
                           q.pend_off, q.plan, nranks(), send_parents);
        return side_done();
    }
    // right after the shard_ingest of the same bucket: the n states just appended came from `src_rank`
    int shard_ingest_parents(const uint64_t *recv_parents, uint64_t n, unsigned src_rank) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n || !d_prank) return MC_OK;  // without MC_F_TRACE nothing is recorded
        hipLaunchKernelGGL(k_ingest_parents, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), recv_parents, n, src_rank, d_parent, d_pslot,
                           d_prank, (const DevCounters *)d_ctr);
        return side_done();
    }
    int shard_violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        int rc = read_counters();
        if (rc) return rc;
        *found = h_ctr->viol_key != ~0ull;
        *idx = 0; *slot = 0; *verdict = MC_V_OK; *invariant = -1;
        if (*found) {
            const unsigned long long k = h_ctr->viol_key;
            const unsigned kind = (unsigned)(k & 7u);
            *idx = viol_idx(k);
            *slot = (uint32_t)(k >> 8 & 0xffffu);
            *verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
            if (kind == VK_INVARIANT) *invariant = (int32_t)(k >> 3 & 31u);
        }
        return MC_OK;
    }
    // one step of a counterexample walk: the state at arena index idx of THIS rank and where its parent lives
    // (parent_rank == shard_rank: here; parent_idx == 0xffffffff: an initial state; parent_slot 0xfffc: the entry is a copy
    // of state parent_idx made by the replicated prefix, not a step)
    int shard_fetch(uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (idx & (1ull << 63)) {  // an initial state, rebuilt from its ordinal in Init's enumeration (a violating initial state is not looked up in the arena)
            const uint64_t ord = idx & ~(1ull << 63);
            if (ord >= S::num_init(prm)) return MC_EBADCFG;
            S::init(prm, ord, WordRef{(uint64_t *)state_out, 1});
            *parent_rank = cfg.shard_rank;
            *parent_idx = 0xffffffffull;
            *parent_slot = SLOT_INIT;
            return MC_OK;
        }
        if (!d_parent) { set_error("engine created without MC_F_TRACE"); return MC_ESTATE; }
        if (idx >= arena_cap) return MC_EBADCFG;
        HIP_TRY(hipDeviceSynchronize());
        int rc = fetch_state(idx, (uint64_t *)state_out);
        if (rc) return rc;
        uint32_t p = 0;
        uint16_t ps = 0;
        uint8_t pr = 0xff;
        HIP_TRY(hipMemcpy(&p, d_parent + idx, sizeof p, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&ps, d_pslot + idx, sizeof ps, hipMemcpyDeviceToHost));
        if (d_prank) HIP_TRY(hipMemcpy(&pr, d_prank + idx, sizeof pr, hipMemcpyDeviceToHost));
        *parent_rank = pr == 0xff ? cfg.shard_rank : pr;
        *parent_idx = p;
        *parent_slot = ps;
        return MC_OK;
    }
    int shard_end_level(uint64_t *new_local) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (sl[0].launched || sl[1].launched) { set_error("shard_end_level: an expand is still in flight"); return MC_EBADCFG; }
        HIP_TRY(hipStreamSynchronize(side()));
        HIP_TRY(hipStreamSynchronize(stream2));
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        sh_next = h_ctr->arena_next;
        last_distinct = sh_next;
        sh_lo = sh_hi;
        sh_hi = sh_next;
        *new_local = sh_hi - sh_lo;
        return MC_OK;
    }
    // invariants of the unexpanded frontier [lo, hi) for specs that check on expansion (see k_check_frontier); no-op otherwise
    int check_frontier(uint64_t lo, uint64_t hi) {
        if constexpr (ChecksOnExpand<S>::value) {
            if (hi > lo) {
                hipLaunchKernelGGL(k_check_frontier<S>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, stream, prm,
                                   (const uint64_t *)d_arena, lo, hi, d_ctr);
                return read_counters();
            }
        }
        return MC_OK;
    }
    int shard_check_frontier() override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        HIP_TRY(hipStreamSynchronize(stream2));
        return check_frontier(sh_lo, sh_hi);
    }
    int shard_counters(uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        *generated = 0;
        for (int t = 0; t < NSHARD; t++) *generated += h_ctr->generated[t].v;
        *distinct_local = h_ctr->arena_next - sh_dup;
        *verdict = MC_V_OK;
        if (h_ctr->viol_key != ~0ull) {
            const unsigned kind = (unsigned)(h_ctr->viol_key & 7u);
            *verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
        }
        return MC_OK;
    }
    int kernel_stats(mc_kernel_stats *o) override {
        o->expand = kstat[0];
        o->insert = kstat[1];
        o->materialise = kstat[2];
        o->state_bytes = (uint64_t)W * 8;
        o->cand_cells = kstat_cells;
        o->inwave_states = kstat_inwave;
        return MC_OK;
    }
};

}  // namespace mc

// --------------------------------------------------------------------------------------- engine factories
// group 1: atomic_add + pcal_intro, 2: raft (2 servers), 3: raft (3), 4: raft (5), 5: serializableSnapshotIsolation,
// 6: compiled PlusCal, 7: Voting / Paxos
namespace mc {
static int spec_group(const mc_spec_desc *d) {
    if (!d) return 0;
    switch (d->spec_id) {
    case MC_SPEC_ATOMIC_ADD: case MC_SPEC_PCAL_INTRO: return 1;
    case MC_SPEC_RAFT: return d->nparams < 1 ? 0 : d->params[0] == 2 ? 2 : d->params[0] == 3 ? 3 : d->params[0] == 5 ? 4 : 0;
    case MC_SPEC_SSI: return 5;
    case MC_SPEC_PCAL: return 6;
    case MC_SPEC_PAXOS: return 7;
    default: return 0;
    }
}
template <class S>
static int make_engine(const typename S::Params &prm, const mc_spec_desc *spec, const mc_config *cfg, EngineBase **out) {
    auto *e = new Engine<S>();
    e->prm = prm;
    e->desc = *spec;
    e->cfg = *cfg;
    const int r = e->alloc();
    if (r) { delete e; return r; }
    *out = e;
    return MC_OK;
}
}  // namespace mc
extern "C" {
int mc_make_engine_1(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_2(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_3(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_4(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_5(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_6(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_7(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
void *mc_jit_factory_opts(const void *program, int pack);   // pcal_codegen.cpp: the factory of the engine library built from the program's generated code, or null
#if MC_TU == 1 || MC_TU == -1
int mc_make_engine_1(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    if (d->spec_id == MC_SPEC_ATOMIC_ADD) {
        mc::SpecAtomicAdd::Params p;
        if (mc::SpecAtomicAdd::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return mc::make_engine<mc::SpecAtomicAdd>(p, d, c, out);
    }
    mc::SpecPcalIntro::Params p;
    if (mc::SpecPcalIntro::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecPcalIntro>(p, d, c, out);
}
#endif
#define MC_RAFT_FACTORY(K, SPEC)                                                                   \
    int mc_make_engine_##K(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {      \
        mc::RaftParams p;                                                                          \
        if (mc::SPEC::make_params(d->params, d->nparams, p)) return MC_EBADCFG;                    \
        return mc::make_engine<mc::SPEC>(p, d, c, out);                                            \
    }
#if MC_TU == 2 || MC_TU == -1
MC_RAFT_FACTORY(2, SpecRaft2)
#endif
#if MC_TU == 3 || MC_TU == -1
MC_RAFT_FACTORY(3, SpecRaft3)
#endif
#if MC_TU == 4 || MC_TU == -1
MC_RAFT_FACTORY(4, SpecRaft5)
#endif
#if MC_TU == 5 || MC_TU == -1
int mc_make_engine_5(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::SsiParams p;
    if (mc::SpecSsi::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecSsi>(p, d, c, out);
}
#endif
#if MC_TU == 7 || MC_TU == -1
// Voting / Paxos (spec_paxos.h)
int mc_make_engine_7(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::PaxosParams p;
    if (mc::SpecPaxos::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecPaxos>(p, d, c, out);
}
#endif
#if MC_TU == 6 || MC_TU == -1
// compiled PlusCal (spec_vm.h): the program image is copied to the device, the kernels read it through prm.code
int mc_make_engine_6(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::VmParams p;
    if (mc::vm_make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    if (hipSetDevice(c->device) != hipSuccess) { mc::set_error("hipSetDevice failed"); return MC_EHIP; }
    int32_t *d_code = nullptr;
    if (hipMalloc(&d_code, (size_t)p.code_len * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(d_code, p.code, (size_t)p.code_len * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        mc::set_error("cannot upload the PlusCal program image");
        return MC_EHIP;
    }
    // identity of the compiled program for checkpoints: the whole image (header, tables, code) and the scalars derived from it
    uint64_t ph = 0xcbf29ce484222325ull;
    auto mixin = [&](uint64_t v) { ph = (ph ^ v) * 0x100000001b3ull; ph ^= ph >> 29; };
    for (int i = 0; i < p.code_len; i++) mixin((uint32_t)p.code[i]);
    for (int v : {p.nv, p.words, p.ninst, p.maxch, p.pc_base, p.done, p.init_entry, p.ninv, p.ncon, p.label_tab, p.self_tab, p.code_len}) mixin((uint32_t)v);
    for (int i = 0; i < 8; i++) mixin((uint32_t)p.inv_entry[i]);
    mixin(p.num_init);
    p.code = d_code;  // host-side helpers (format, action_of) use p.host only
    const int rc = p.nv <= 16 ? mc::make_engine<mc::SpecVm16>(p, d, c, out)
                 : p.nv <= 32 ? mc::make_engine<mc::SpecVm32>(p, d, c, out)
                 : p.nv <= 64 ? mc::make_engine<mc::SpecVm64>(p, d, c, out) : mc::make_engine<mc::SpecVm>(p, d, c, out);
    if (rc) hipFree(d_code);
    else { (*out)->owned_device_blob = d_code; (*out)->program_hash = ph ? ph : 1; }
    return rc;
}
#endif
#if MC_TU == 9
// A compiled PlusCal program as GENERATED code (spec_gen.h; pcal_codegen.cpp writes the header and builds this translation unit into a
// library of its own when the engine is created: mc_jit_factory).  Same parameters, same packed states, same program identity as group 6.
}  // extern "C"
#include MC_GEN_HEADER
extern "C" {
int mc_make_engine_gen(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::VmParams p;
    if (mc::SpecGen::make_params(d->params, d->nparams, p)) { mc::set_error("jit: the generated engine was built from another program"); return MC_EBADCFG; }
    uint64_t ph = 0xcbf29ce484222325ull;
    auto mixin = [&](uint64_t v) { ph = (ph ^ v) * 0x100000001b3ull; ph ^= ph >> 29; };
    for (int i = 0; i < p.code_len; i++) mixin((uint32_t)p.code[i]);
    for (int v : {p.nv, p.words, p.ninst, p.maxch, p.pc_base, p.done, p.init_entry, p.ninv, p.ncon, p.label_tab, p.self_tab, p.code_len}) mixin((uint32_t)v);
    for (int i = 0; i < 8; i++) mixin((uint32_t)p.inv_entry[i]);
    mixin(p.num_init);
    p.code = nullptr;   // the device never reads the image: the program IS the kernels
    const int rc = mc::make_engine<mc::SpecGen>(p, d, c, out);
    if (mc::SpecGen::PACKED_ROWS) { mixin(0x7061636bu); mixin((uint32_t)mc::SpecGen::MAX_WORDS); }   // a checkpoint of packed rows is not the interpreter's
    if (!rc) (*out)->program_hash = ph ? ph : 1;
    return rc;
}
#endif
}  // extern "C"

#if MC_TU == 0 || MC_TU == -1
static thread_local std::string g_last_error;
// --------------------------------------------------------------------------------------- C ABI
using namespace mc;

struct mc_engine {
    EngineBase *impl;
};

extern "C" {

int mc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mc_engine_create(const mc_spec_desc *spec, const mc_config *cfg, mc_engine **out) {
    if (!spec || !cfg || !out) return MC_EBADCFG;
    *out = nullptr;
    if (mc_device_count() <= 0) { set_error("no HIP device: libtlamc has no CPU fallback"); return MC_EHIP; }
    g_last_error.clear();
    EngineBase *impl = nullptr;
    const int group = spec_group(spec);
    int rc = MC_EBADCFG;
    switch (group) {
    case 1: rc = mc_make_engine_1(spec, cfg, &impl); break;
    case 2: rc = mc_make_engine_2(spec, cfg, &impl); break;
    case 3: rc = mc_make_engine_3(spec, cfg, &impl); break;
    case 4: rc = mc_make_engine_4(spec, cfg, &impl); break;
    case 5: rc = mc_make_engine_5(spec, cfg, &impl); break;
    case 6: {
        // MC_F_JIT / $TLAMC_JIT: the program as generated, compiled code (pcal_codegen.cpp); the interpreter when that is not to be had
        const char *ej = getenv("TLAMC_JIT");
        rc = MC_EBADCFG;
        bool jit = false;
        if (((cfg->flags & MC_F_JIT) || (ej && *ej && *ej != '0')) && spec->nparams >= 1 && spec->params[0]) {
            typedef int (*factory_t)(const mc_spec_desc *, const mc_config *, EngineBase **);
            factory_t fn = (factory_t)mc_jit_factory_opts((const void *)(intptr_t)spec->params[0], cfg->shard_count > 1 ? 0 : 1);
            // (the library is there and ITS engine cannot be made — out of device memory, a bad configuration: the caller's error, the
            //  interpreter's engine would meet it too; the interpreter stands in only for generated code that is not to be had)
            if (fn) { rc = fn(spec, cfg, &impl); jit = true; }
            else fprintf(stderr, "tlamc: MC_F_JIT: %s; interpreting the program on the device instead\n", g_last_error.c_str());
        }
        if (!jit) rc = mc_make_engine_6(spec, cfg, &impl);
        break;
    }
    case 7: rc = mc_make_engine_7(spec, cfg, &impl); break;
    default: break;
    }
    if (rc == MC_OK) *out = new mc_engine{impl};
    else if (rc == MC_EBADCFG && g_last_error.empty()) set_error("unknown spec id or constants out of range");
    return rc;
}
int mc_engine_run(mc_engine *e, mc_result *out) { return e && out ? e->impl->run(out) : MC_EBADCFG; }
int mc_engine_step(mc_engine *e, uint32_t levels, mc_result *out) { return e && out ? e->impl->step(levels, out) : MC_EBADCFG; }
int mc_engine_set_progress(mc_engine *e, mc_progress_fn fn, void *user, double min_interval_seconds) {
    if (!e) return MC_EBADCFG;
    e->impl->progress_fn = fn;
    e->impl->progress_user = user;
    e->impl->progress_interval = min_interval_seconds > 0 ? min_interval_seconds : 0.0;
    return MC_OK;
}
int mc_engine_request_stop(mc_engine *e) {
    if (!e) return MC_EBADCFG;
    e->impl->stop_requested = true;
    return MC_OK;
}
int mc_engine_trace(mc_engine *e, uint8_t *states_out, int32_t *actions_out, size_t *n_inout) {
    return e && n_inout ? e->impl->trace(states_out, actions_out, n_inout) : MC_EBADCFG;
}
int mc_engine_kernel_stats(mc_engine *e, mc_kernel_stats *out) { return e && out ? e->impl->kernel_stats(out) : MC_EBADCFG; }
int mc_engine_read_states(mc_engine *e, uint64_t first, uint64_t count, uint8_t *out) {
    return e && (out || !count) ? e->impl->read_states(first, count, out) : MC_EBADCFG;
}
int mc_engine_checkpoint(mc_engine *e, const char *path) { return e && path ? e->impl->checkpoint(path) : MC_EBADCFG; }
int mc_engine_restore(mc_engine *e, const char *path) { return e && path ? e->impl->restore(path) : MC_EBADCFG; }
void mc_engine_destroy(mc_engine *e) {
    if (!e) return;
    delete e->impl;
    delete e;
}

size_t mc_state_bytes(const mc_spec_desc *spec) {
    size_t n = 0;
    dispatch_spec(spec, [&](auto s, const auto &prm) { n = sizeof(uint64_t) * decltype(s)::words(prm); return 0; });
    return n;
}
uint32_t mc_fp_owner(uint64_t fp, uint32_t shard_count) { return fp_owner(fp, shard_count); }
int mc_state_format(const mc_spec_desc *spec, const uint8_t *state, char *buf, size_t cap) {
    if (!state || !buf || !cap) return MC_EBADCFG;
    int n = MC_EBADCFG;
    int rc = dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        n = S::format(prm, w, buf, cap);
        if ((size_t)n < cap) buf[n] = 0; else buf[cap - 1] = 0;
        return 0;
    });
    return rc ? rc : n;
}
// host-side evaluation of one (state, slot) pair: the action it is (mc_action_name's argument) and its successor
int mc_state_action(const mc_spec_desc *spec, const uint8_t *state, int32_t slot) {
    if (!state) return MC_EBADCFG;
    int a = -1;
    int rc = dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        a = S::action_of(prm, w, (int)slot);
        return 0;
    });
    return rc ? rc : a;
}
int mc_state_apply(const mc_spec_desc *spec, const uint8_t *state, int32_t slot, uint8_t *successor_out) {
    if (!state || !successor_out) return MC_EBADCFG;
    return dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS], o[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        S::apply(prm, CWordRef{w, 1}, (int)slot, WordRef{o, 1});
        memcpy(successor_out, o, sizeof(uint64_t) * S::words(prm));
        return 0;
    });
}
const char *mc_action_name(const mc_spec_desc *spec, int32_t action) {
    const char *nm = "?";
    dispatch_spec(spec, [&](auto s, const auto &prm) {
        if constexpr (std::is_same_v<std::decay_t<decltype(prm)>, VmParams>) nm = vm_action_name(prm.host, action);  // label names live in the program
        else nm = decltype(s)::action_name(action);
        return 0;
    });
    return nm;
}
const char *mc_strerror(int code) {
    switch (code) {
    case MC_OK: return "ok";
    case MC_EBADCFG: return "bad spec or configuration";
    case MC_EHIP: return "HIP error / no device";
    case MC_EOVERFLOW: return "packed-state slot array overflow";
    case MC_ETABLEFULL: return "seen-set full";
    case MC_EARENA: return "state arena full";
    case MC_ERCCL: return "exchange failure";
    case MC_ESTATE: return "call sequence error";
    case MC_EPARSE: return "parse error";
    case MC_ENOSPEC: return "module is not a lowered spec";
    case MC_EROUTE: return "exchange bucket full";
    default: return "unknown error";
    }
}
const char *mc_last_error(void) { return g_last_error.c_str(); }
void mc_set_error_internal(const char *msg) { g_last_error = msg ? msg : ""; }


int mc_engine_debug_reexpand(mc_engine *e, unsigned extra_flags, double *ms) { return e && ms ? e->impl->debug_reexpand(extra_flags, ms) : MC_EBADCFG; }
// profiling builds only (-DMC_PHASE_PROF): cycles per phase of k_expand_family summed over all wavefronts since the last reset
int mc_engine_debug_phases(mc_engine *e, uint64_t *out48, int reset) { return e && out48 ? e->impl->debug_phases(out48, reset) : MC_EBADCFG; }
int mc_engine_debug_flags(mc_engine *e, uint32_t set, uint32_t clear) { return e ? e->impl->debug_flags(set, clear) : MC_EBADCFG; }
// ---- sharded (multi-GPU) step API
int mc_shard_begin(mc_engine *e) { return e ? e->impl->shard_begin() : MC_EBADCFG; }
int mc_shard_begin_replicated(mc_engine *e, uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out,
                              uint32_t *nlevels) {
    return e && levels_out && nlevels ? e->impl->shard_begin_replicated(min_frontier, max_distinct, max_levels, levels_out, nlevels) : MC_EBADCFG;
}
int mc_shard_level_size(mc_engine *e, uint64_t *n) { return e && n ? e->impl->shard_level_size(n) : MC_EBADCFG; }
int mc_shard_info(mc_engine *e, void **main_stream_out, uint64_t *chunk_states_out, int32_t *traced_out) {
    return e ? e->impl->shard_info(main_stream_out, chunk_states_out, traced_out) : MC_EBADCFG;
}
size_t mc_engine_state_bytes_internal(mc_engine *e) { return e ? e->impl->state_bytes() : 0; }
int mc_shard_expand(mc_engine *e, uint64_t first, uint64_t count, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) {
    if (!e || !send_counts) return MC_EBADCFG;
    const int rc = e->impl->shard_expand_launch(0, first, count, send_cap);
    return rc ? rc : e->impl->shard_expand_finish(0, send_fp, send_cap, send_counts);
}
int mc_shard_set_stream(mc_engine *e, void *hip_stream, int enable) { return e ? e->impl->shard_set_stream(hip_stream, enable) : MC_EBADCFG; }
int mc_shard_expand_launch(mc_engine *e, uint32_t slot, uint64_t first, uint64_t count, uint64_t send_cap) {
    return e ? e->impl->shard_expand_launch(slot, first, count, send_cap) : MC_EBADCFG;
}
int mc_shard_expand_finish(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_expand_finish(slot, send_fp, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_probe(mc_engine *e, const uint64_t *recv_fp, uint64_t n, uint8_t *answers) {
    return e ? e->impl->shard_probe(recv_fp, n, answers) : MC_EBADCFG;
}
int mc_shard_materialise(mc_engine *e, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_materialise(0, answers_back, send_states, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_materialise_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap,
                              uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_materialise(slot, answers_back, send_states, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_ingest(mc_engine *e, const uint8_t *recv_states, uint64_t n) { return e ? e->impl->shard_ingest(recv_states, n) : MC_EBADCFG; }
int mc_shard_expand_pack(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t cap) { return e && send_fp ? e->impl->shard_expand_pack(slot, send_fp, cap) : MC_EBADCFG; }
int mc_shard_probe_pack(mc_engine *e, const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) {
    return e && recv_fp && answers ? e->impl->shard_probe_pack(recv_fp, cap, answers) : MC_EBADCFG;
}
int mc_shard_keep_pack(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t cap) {
    return e && answers_back ? e->impl->shard_keep_pack(slot, answers_back, cap) : MC_EBADCFG;
}
int mc_shard_wait_keep(mc_engine *e, uint32_t slot) { return e ? e->impl->shard_wait_keep(slot) : MC_EBADCFG; }
int mc_shard_keep(mc_engine *e, const uint8_t *answers_back, uint64_t *n_new) { return e && n_new ? e->impl->shard_keep(0, answers_back, n_new) : MC_EBADCFG; }
int mc_shard_keep_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t *n_new) {
    return e ? e->impl->shard_keep(slot, answers_back, n_new) : MC_EBADCFG;
}
int mc_shard_end_level(mc_engine *e, uint64_t *new_local) { return e && new_local ? e->impl->shard_end_level(new_local) : MC_EBADCFG; }
int mc_shard_counters(mc_engine *e, uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) {
    return e && generated && distinct_local && verdict ? e->impl->shard_counters(generated, distinct_local, verdict) : MC_EBADCFG;
}
int mc_shard_check_frontier(mc_engine *e) { return e ? e->impl->shard_check_frontier() : MC_EBADCFG; }
int mc_shard_materialise_parents(mc_engine *e, uint32_t slot, uint64_t *send_parents) { return e ? e->impl->shard_materialise_parents(slot, send_parents) : MC_EBADCFG; }
int mc_shard_ingest_parents(mc_engine *e, const uint64_t *recv_parents, uint64_t n, uint32_t src_rank) {
    return e ? e->impl->shard_ingest_parents(recv_parents, n, src_rank) : MC_EBADCFG;
}
int mc_shard_violation(mc_engine *e, int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) {
    return e && found && idx && slot && verdict && invariant ? e->impl->shard_violation(found, idx, slot, verdict, invariant) : MC_EBADCFG;
}
int mc_shard_note_levels(mc_engine *e, const uint64_t *levels, uint32_t n, int32_t verdict) { return e && (levels || !n) ? e->impl->shard_note_levels(levels, n, verdict) : MC_EBADCFG; }
int mc_shard_resume(mc_engine *e, uint64_t *levels_out, uint32_t *nlevels) { return e && levels_out && nlevels ? e->impl->shard_resume(levels_out, nlevels) : MC_EBADCFG; }
int mc_shard_checkpoint(mc_engine *e, const char *path) { return e && path ? e->impl->shard_checkpoint(path) : MC_EBADCFG; }
int mc_shard_restore(mc_engine *e, const char *path) { return e && path ? e->impl->shard_restore(path) : MC_EBADCFG; }
int mc_shard_fetch(mc_engine *e, uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) {
    return e && state_out && parent_rank && parent_idx && parent_slot ? e->impl->shard_fetch(idx, state_out, parent_rank, parent_idx, parent_slot) : MC_EBADCFG;
}

}  // extern "C"
#endif  // MC_TU == 0 || MC_TU == -1
