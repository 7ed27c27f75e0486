// shard_loop.h — THE level loop of the fingerprint-sharded search, written once against two small interfaces:
//
//   * `mc_transport` (include/tlamc.h): the collectives — RCCL in the product (shard_rccl.cpp), torch.distributed through
//     callbacks (tla_rust_amd/sharded.py: backend "nccl", or gloo staged through the host in the CPU tests);
//   * `Ops`: the step calls of ONE rank's engine — mc_shard_* of the C ABI in the product (shard_rccl.cpp `AbiOps`), the host
//     build of the lowerings in the CPU tests (tests/_shim/shim.cpp `ShimOps`).
//
// Round 2 had two loops (this one's ancestor in C++ over RCCL, which only knew the "stay" form, and a Python one over
// torch.distributed that the tests drove); they are one now: every multi-rank test, `mc X.tla -gpus P`, `bench.py --gpus N`
// and the torch front door run the code below.
//
// Per level ONE collective for the job's state: the all-gather of {frontier size, verdict, status} of every rank (host-paced rounds
// add two host waits each: the round's route cursors and the all-gather of its bucket counts).  A rank-local failure
// is sticky and collective: the failing rank keeps taking part in the level's collectives with empty buckets, its status travels
// with the all-gather, and all ranks leave together (nobody is left waiting in a collective).
//
//   STAY level (large, balanced frontier): the new states stay on the rank that generated them; 9 bytes per routed candidate are
//     what the exchange needs.  Default form: host-paced rounds with EXACT sizes (the first half of a move round + mc_shard_keep_slot:
//     one host wait for the round's expand and two small all-gathers per round (the bucket counts; "every rank has its buffers"), both behind the launch of the
//     next round's expand).  MC_SHARD_PACKED: fixed-capacity rounds, nothing waits for the host.  Streams: MAIN (the engine's expand
//     stream: expand r, bucket compaction r), COMM (the transport's: the all-to-alls), WORK (probes), the engine's second
//     stream (materialisation of the kept states).  Issue order on COMM is fp(0) fp(1) ans(0) fp(2) ans(1) ...: the fingerprint
//     exchange of round r+1 overlaps the probes of round r, the expand of round r+1 overlaps both.  9 bytes per routed
//     candidate cross xGMI; new states stay on the rank that generated them.
//   MOVE level (small frontier, or the largest rank holds more than rebalance_ratio x the mean): host-paced rounds with
//     exact sizes; the new states travel to their owners as whole 64-state blocks (+ their parent pointers with MC_F_TRACE),
//     which is what spreads a small frontier — and a drifted one — evenly again.
#ifndef MC_SHARD_LOOP_H
#define MC_SHARD_LOOP_H
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/tlamc.h"

extern "C" void mc_set_error_internal(const char *msg);

namespace mc_shard {

enum Stream { S_COMM = 0, S_WORK = 1, S_MAIN = 2 };
// event ids of a stay level (two slots each) and of the serialised move rounds
enum Event { EV_PACKED = 0, EV_FP = 2, EV_PROBED = 4, EV_ANS = 6, EV_TMP = 8, EV_COUNT = 10 };

constexpr uint32_t SLOT_NONE = 0xffffu, SLOT_INIT = 0xfffeu, SLOT_PARENT = 0xfffdu, SLOT_COPY = 0xfffcu;
constexpr uint64_t NO_PARENT = 0xffffffffull;
constexpr uint64_t FETCH_INIT = 1ull << 63;  // Ops::fetch / mc_shard_fetch: the low bits are an ordinal of Init's enumeration, not an arena index

// exchange buffer from the transport's allocator; grows on demand (only ever between levels or in host-paced rounds)
struct NetBuf {
    const mc_transport *t = nullptr;
    void *p = nullptr;
    size_t bytes = 0;
    int need(size_t n) {
        if (n <= bytes && p) return MC_OK;
        if (p) t->release(t->user, p);
        p = nullptr;
        // (geometric: the exact sizes of host-paced rounds grow level by level, and a release waits for the device)
        if (bytes) n = std::max(n, bytes + bytes / 2);
        bytes = 0;
        n = (n + 4095) & ~(size_t)4095;
        p = t->alloc(t->user, n);
        if (!p) { mc_set_error_internal("sharded search: cannot allocate an exchange buffer"); return MC_EHIP; }
        bytes = n;
        return MC_OK;
    }
    ~NetBuf() { if (p && t) t->release(t->user, p); }
};

struct LevelInfo { uint64_t n, verdict, status, pad; };

template <class Ops>
struct Loop {
    Ops &e;
    const mc_transport &net;
    const uint32_t P, me;
    std::vector<LevelInfo> all;
    std::vector<uint64_t> sizes;
    int lrc = MC_OK;  // this rank's sticky status: once non-zero, no engine call is made any more, the collectives still are
    mc_shard_stats st;
    // Bucket capacities from MEASURED fill (the `measured` form of a stay level, MC_SHARD_PACKED without MC_SHARD_FIXED_CAPS).  A
    // fixed-capacity round moves its buckets whole, so their capacity IS the exchange volume.  packed_fanout allows for 16 in-model
    // successors per expanded state; the models route 3 - 6 per state to each owner's share, and a level sends about what the level
    // before it sent.  Every rank therefore reports, with the level's all-gather, the fullest bucket its rounds filled per expanded
    // state (`fill`, in 2^-20 entries per state: Ops::route_fill — the HIP ops read every bucket's count word back behind its
    // compaction; a host-paced round knows its exact sizes), every rank takes the maximum, and the next stay level's buckets hold
    // cap_safety_pct of that (+ 1024 entries for the small-number noise), never more than the buffers were allocated for.  Counts never
    // depend on it: a bucket that does not fit fails the level as before (MC_EROUTE), and the search restarted for it sizes its buckets
    // from twice packed_fanout (MC_SHARD_FIXED_CAPS).  The bet can be lost — inside a level the candidates per state rise in BFS order
    //  — which is why the exact form is the default.
    static constexpr uint64_t FILL_ONE = 1ull << 20, FILL_MIN_STATES = 1024;
    uint64_t fill_prev = 0;        // max over ranks, previous level; 0 = not known
    uint64_t fill_mine = 0;        // this rank, current level
    bool level_measured = false;   // the current (or failing) level's buckets were sized from fill_prev
    void note_fill(uint64_t bucket_entries, uint64_t states) {
        if (states >= FILL_MIN_STATES) fill_mine = std::max(fill_mine, (bucket_entries * FILL_ONE + states - 1) / states);
    }

    Loop(Ops &ops, const mc_transport &t) : e(ops), net(t), P(t.world), me(t.rank), all(t.world), sizes(t.world) { memset(&st, 0, sizeof st); probe_rot = t.rank; }

    // engine step: skipped once this rank has failed.  The host time inside engine calls (their stream waits included) and inside
    // the transport's collectives is what mc_shard_stats.engine_ns / collective_ns report: with the GPU kernels' own times
    // (mc_engine_kernel_stats) they say where a sharded step's wall time goes — a rank waiting for its own streams, for its
    // slowest peer inside a collective, or for neither.
    static uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    template <class F>
    void step(F &&f) {
        if (lrc) return;
        const uint64_t t0 = now_ns();
        lrc = f();
        st.engine_ns += now_ns() - t0;
    }
    template <class Fn, class... A>
    int timed_net(Fn fn, A... a) {
        const uint64_t t0 = now_ns();
        const int rc = fn(net.user, a...);
        st.collective_ns += now_ns() - t0;
        ++st.collectives;
        return rc;
    }

    // TEST-ONLY (include/tlamc.h): $TLAMC_TEST_FAIL_AT = "rank:level:code[,rank:level:code...]" makes that rank fail the level with
    // that status, ONCE per process (a restarted search is not failed again) — how the tests put two different failures into one
    // level without building a model that produces them.
    // ONCE per process is the hook's contract (the restarted search builds a new Loop and must not be failed again), so the one-shot flag
    // is process-wide; the variable itself is read once per process, not once per level
    int test_fault(size_t level) {
        static const char *const env = getenv("TLAMC_TEST_FAIL_AT");
        static std::atomic<bool> spent{false};
        if (!env || spent.load()) return MC_OK;
        for (const char *q = env; *q;) {
            unsigned r = 0, lv = 0;
            int code = 0, used = 0;
            if (sscanf(q, "%u:%u:%d%n", &r, &lv, &code, &used) < 3) break;
            if (r == me && lv == level) {
                spent.store(true);
                mc_set_error_internal("TLAMC_TEST_FAIL_AT: injected failure");
                return code;
            }
            q += used;
            if (*q == ',') ++q;
        }
        return MC_OK;
    }
    // What the ranks AGREE the failure is, from the all-gathered statuses (the same value on every rank): a rank must never act on
    // its own status alone — run_restarting starts over on MC_EROUTE, and a rank that restarted while another returned would wait
    // in the next collective for ever.  MC_EROUTE (restart with twice the allowance) only if EVERY failing rank reports it; else the
    // first failing rank's status that is not MC_EROUTE.  A rank's own error text stays where its own failure put it.
    int agreed_failure() {
        int first = 0, first_other = 0;
        uint32_t who = 0, who_other = 0;
        for (uint32_t p = 0; p < P; ++p) {
            const int stt = (int)(int64_t)all[p].status;
            if (!stt) continue;
            if (!first) { first = stt; who = p; }
            if (!first_other && stt != MC_EROUTE) { first_other = stt; who_other = p; }
        }
        const int code = first_other ? first_other : first;
        const uint32_t p = first_other ? who_other : who;
        if (code && p != me && !lrc) mc_set_error_internal(("sharded search: rank " + std::to_string(p) + " failed: " + mc_strerror(code)).c_str());
        return code;
    }
    // ONE collective per level: every rank's frontier size, verdict and status.  Returns the transport's error only.
    int level_info(uint64_t local_n, int32_t verdict, uint64_t &frontier, int32_t &worst, int &failed, uint64_t sig = 0) {
        LevelInfo mine{local_n, (uint64_t)verdict, (uint64_t)(int64_t)lrc, sig};
        int trc = timed_net(net.all_gather, &mine, all.data(), sizeof mine);
        if (trc) return trc;
        frontier = 0;
        worst = 0;
        failed = 0;
        for (uint32_t p = 0; p < P; ++p) {
            sizes[p] = all[p].n;
            frontier += sizes[p];
            worst = std::max(worst, (int32_t)all[p].verdict);
        }
        failed = agreed_failure();
        return MC_OK;
    }

    int run(const mc_shard_opts &o, mc_result *out) {
        if (!out || !P || me >= P || P > 8) { mc_set_error_internal("sharded search: bad transport (1..8 ranks)"); return MC_EBADCFG; }
        memset(out, 0, sizeof *out);
        out->violated_invariant = -1;
        uint64_t chunk = o.chunk_states ? o.chunk_states : (1ull << 19);
        if (e.chunk_limit() && chunk > e.chunk_limit()) chunk = e.chunk_limit();  // never more than one launch of the engine takes
        const uint64_t fan = o.packed_fanout ? o.packed_fanout : 16, mfan = o.move_fanout ? o.move_fanout : 32;
        // (default = the replicated prefix's own threshold: the prefix hands every rank an even share of a level that size, so the
        //  first sharded level can already keep its new states where they are generated)
        const uint64_t stay_threshold = o.stay_threshold ? o.stay_threshold : (1ull << 15);
        const double ratio = o.rebalance_ratio > 0 ? o.rebalance_ratio : 1.25;
        const uint64_t safety = (o.flags & MC_SHARD_FIXED_CAPS) ? 0 : o.cap_safety_pct ? o.cap_safety_pct : 140;
        const size_t W = e.state_bytes();
        const bool traced = e.traced();
        int trc;

        // ---- a run restored from per-rank checkpoints continues with its unexpanded frontier (mc_shard_restore)
        std::vector<uint64_t> levels(MC_MAX_LEVELS);
        {
            uint32_t rn = MC_MAX_LEVELS;
            step([&] { return e.resume(levels.data(), &rn); });
            levels.resize(lrc ? 0 : rn);
        }
        const bool resumed = !levels.empty();
        uint64_t resume_sig = 0;  // every rank must continue the SAME run: the level tables travel with the first all-gather
        for (uint64_t v : levels) resume_sig = resume_sig * 0x9e3779b97f4a7c15ull + v + 1;
        // ---- the small levels: the same fused BFS on every rank, then each rank keeps the states it owns
        if (resumed) {
        } else if (o.flags & MC_SHARD_NO_PREFIX) {
            step([&] { return e.begin(); });
        } else {
            levels.resize(MC_MAX_LEVELS);
            uint32_t nlev = MC_MAX_LEVELS;
            const uint64_t until = (o.replicate_until ? o.replicate_until : (1ull << 15)) * P;
            step([&] { return e.begin_replicated(until, o.max_distinct, o.max_levels, levels.data(), &nlev); });
            levels.resize(lrc ? 0 : nlev);
            st.replicated_levels = levels.size();
        }
        uint64_t local_n = 0, gen = 0, dl = 0, frontier = 0;
        int32_t verdict = 0, worst = 0;
        int failed = 0;
        step([&] { return e.level_size(&local_n); });
        step([&] { return e.counters(&gen, &dl, &verdict); });
        uint64_t routed_base = 0, routed_now = 0, bucket_max = 0;
        step([&] { return e.route_fill(&bucket_max, &routed_base); });
        routed_now = routed_base;
        if ((trc = level_info(local_n, verdict, frontier, worst, failed, resume_sig))) return trc;
        if (failed) return failed;   // (the agreed code, the same on every rank: agreed_failure)
        for (uint32_t p = 0; p < P; ++p)
            if (all[p].pad != resume_sig) { mc_set_error_internal("sharded search: the ranks did not restore checkpoints of the same run (mc_shard_restore on every rank, or on none)"); return MC_EBADCFG; }
        uint64_t cum = 0;
        if (resumed) {
            // A checkpoint of a run that STOPPED ON A BUDGET leaves its last level unexpanded: the ranks' frontiers add up to it.  A
            // checkpoint of a FINISHED run (mc_shard_checkpoint accepts MC_V_OK, `mc -gpus P -checkpoint` writes it) has no frontier
            // left: the loop below does not run and the finished result is reported again.  Either way the states the ranks hold
            // must add up to the level table.
            uint64_t total = 0, held = 0;
            for (uint64_t v : levels) total += v;
            std::vector<uint64_t> every(P);
            uint64_t mine_dl = lrc ? 0 : dl;
            if ((trc = timed_net(net.all_gather, &mine_dl, every.data(), sizeof mine_dl))) return trc;
            for (uint32_t p = 0; p < P; ++p) held += every[p];
            if ((frontier != 0 && frontier != levels.back()) || held != total) {
                mc_set_error_internal("sharded search: the restored frontiers / states do not add up to the checkpointed level table");
                return MC_EBADCFG;
            }
        } else if (o.flags & MC_SHARD_NO_PREFIX) {
            if (frontier) levels.push_back(frontier);
        } else if (worst != 0) {
            frontier = levels.empty() ? 0 : levels.back();
        }
        for (uint64_t v : levels) cum += v;
        bool budget = false;

        NetBuf send[2], recv[2], ans[2], back[2], states, rstates, parents, rparents;
        for (NetBuf *b : {&send[0], &send[1], &recv[0], &recv[1], &ans[0], &ans[1], &back[0], &back[1], &states, &rstates, &parents, &rparents}) b->t = &net;

        while (frontier > 0 && worst == 0) {
            if ((o.max_levels && levels.size() >= o.max_levels) || (o.max_distinct && cum >= o.max_distinct)) { budget = true; break; }
            uint64_t max_n = 0;
            for (uint32_t p = 0; p < P; ++p) max_n = std::max(max_n, sizes[p]);
            // imbalance of the worst large level (max over ranks against the mean), before this level's stay / move decision
            if (frontier >= stay_threshold * P && (double)max_n * st.mean_frontier * P >= (double)st.max_frontier * frontier) {
                st.max_frontier = max_n;
                st.mean_frontier = std::max<uint64_t>(frontier / P, 1);
            }
            const uint64_t mine = sizes[me];
            const bool stay = frontier >= stay_threshold * P && (double)max_n * P <= ratio * (double)frontier;
            // a move level ships whole states: smaller rounds keep its buffers modest
            // the stay level as host-paced rounds with exact sizes (a single rank exchanges nothing: it keeps the rounds that never wait)
            const bool exact = stay && P > 1 && !(o.flags & (MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS));
            const uint64_t ch = stay ? chunk : std::min<uint64_t>(chunk, 1ull << 17);
            const uint64_t rounds = (max_n + ch - 1) / ch;
            (stay ? st.stay_levels : st.move_levels)++;
            st.rounds += rounds;
            const uint64_t f = stay ? fan : mfan;
            const uint64_t send_cap = (uint64_t)P * (std::min(ch, max_n) * f / P + 4096);  // candidates of one round, all owners
            fill_mine = 0;
            level_measured = stay && !exact && safety && fill_prev;
            if (level_measured) st.measured_levels++;
            auto launch = [&](uint64_t r) {
                const uint64_t first = std::min(r * ch, mine), n = std::min(ch, mine - first);
                step([&] { return e.expand_launch((uint32_t)(r & 1), first, n, send_cap); });
            };
            if (stay && !exact) {
                if ((trc = stay_level(rounds, ch, fan, max_n, level_measured ? safety : 0, launch, send, recv, ans, back))) return trc;
            } else {
                if ((trc = move_level(rounds, ch, send_cap, W, traced, exact, launch, send, recv, ans, back, states, rstates, parents, rparents))) return trc;
            }
            uint64_t new_local = 0;
            step([&] { return e.end_level(&new_local); });  // waits for the engine's streams; device errors surface here
            step([&] { return test_fault(levels.size()); });
            step([&] { return e.counters(&gen, &dl, &verdict); });
            if (stay && !exact) {  // (host-paced rounds noted their exact sizes round by round)
                bucket_max = 0;
                step([&] { return e.route_fill(&bucket_max, &routed_now); });
                if (!lrc) note_fill(bucket_max, std::min(ch, mine));
            }
            if ((trc = level_info(new_local, verdict, frontier, worst, failed, lrc ? 0 : fill_mine))) return trc;
            if (failed) return failed;   // (the agreed code, the same on every rank: agreed_failure)
            fill_prev = 0;
            for (uint32_t p = 0; p < P; ++p) fill_prev = std::max(fill_prev, all[p].pad);
            if (me == 0 && getenv("TLAMC_SHARD_DEBUG"))
                fprintf(stderr, "[shard] level %zu %s%s: %llu states, %llu rounds, fullest bucket %.3f entries per expanded state -> %llu new\n", levels.size(),
                        stay ? "stay" : "move", level_measured ? " (measured caps)" : "", (unsigned long long)(cum ? levels.back() : 0),
                        (unsigned long long)rounds, (double)fill_prev / (double)FILL_ONE, (unsigned long long)frontier);
            if (frontier > 0) {
                if (levels.size() >= MC_MAX_LEVELS) { mc_set_error_internal("more BFS levels than MC_MAX_LEVELS"); return MC_EBADCFG; }
                levels.push_back(frontier);
                cum += frontier;
            }
        }
        if (frontier > 0 && worst == 0) step([&] { return e.check_frontier(); });  // a budget stop leaves a level unexpanded
        step([&] { return e.counters(&gen, &dl, &verdict); });
        {   // global counters: sum of generated, worst verdict (the unexpanded frontier's check included), status once more
            LevelInfo mine2{gen, (uint64_t)verdict, (uint64_t)(int64_t)lrc, 0};
            if ((trc = timed_net(net.all_gather, &mine2, all.data(), sizeof mine2))) return trc;
            gen = 0;
            worst = 0;
            for (uint32_t p = 0; p < P; ++p) {
                gen += all[p].n;
                worst = std::max(worst, (int32_t)all[p].verdict);
            }
            if (const int f = agreed_failure()) return f;
        }
        st.distinct_local = dl;
        st.routed_candidates += routed_now - routed_base;
        if (o.stats) *o.stats = st;
        out->distinct = cum;
        out->generated = gen;
        out->queue_left = frontier;
        out->depth = (uint32_t)levels.size();
        out->levels = (uint32_t)levels.size();
        for (size_t k = 0; k < levels.size(); ++k) out->level_distinct[k] = levels[k];
        out->verdict = worst != 0 ? worst : budget ? MC_V_BUDGET : MC_V_OK;
        step([&] { return e.note_levels(levels.data(), (uint32_t)levels.size(), out->verdict); });  // what mc_shard_checkpoint writes
        return lrc;
    }

    // ------------------------------------------------------------------ STAY: fixed-capacity rounds, counts in band, no host wait
    size_t stay_bytes = 0;
    template <class L>
    int stay_level(uint64_t rounds, uint64_t ch, uint64_t fan, uint64_t max_n, uint64_t safety, L &&launch, NetBuf *send, NetBuf *recv, NetBuf *ans,
                   NetBuf *back) {
        int trc;
        // every rank derives the same capacities from the level's frontier sizes (which all ranks know): the exchanges are
        // equal-split.  A rank's candidates spread evenly over the P owners (its own share is probed locally, its own bucket stays empty).
        const uint64_t cap_max = std::min(ch, max_n) * fan / P + 4096;
        auto cap_of = [&](uint64_t r) {
            uint64_t n_round = 0;
            for (uint32_t p = 0; p < P; ++p) n_round = std::max(n_round, std::min(ch, sizes[p] > r * ch ? sizes[p] - r * ch : 0));
            // measured form: what the fullest bucket of the previous level held per state, with the safety margin (it may also be MORE
            // than packed_fanout's share, up to what the buffers were allocated for); otherwise packed_fanout candidates per state,
            // spread over the P owners
            if (safety) return std::min((uint64_t)((unsigned __int128)n_round * fill_prev * safety / (100 * FILL_ONE)) + 1024, cap_max);
            return std::min((P > 1 ? n_round * fan / P : 0) + 1024, cap_max);  // (one rank: its only bucket is its own, always empty)
        };
        const size_t total_max = (size_t)P * (ch * fan / P + 4096);  // the largest a level of this run can ask for
        if (stay_bytes < total_max) {
            // Allocated once, before the first round that needs them, and agreed on: an allocation failure on one rank must not
            // leave a peer waiting inside the rounds (nothing else in a stay round can fail on one rank alone).
            for (int s = 0; s < 2; ++s) {
                step([&] { return send[s].need(total_max * 8); });
                step([&] { return recv[s].need(total_max * 8); });
                step([&] { return ans[s].need(total_max); });
                step([&] { return back[s].need(total_max); });
            }
            uint64_t mine = (uint64_t)(int64_t)lrc;
            std::vector<uint64_t> every(P);
            if ((trc = timed_net(net.all_gather, &mine, every.data(), sizeof mine))) return trc;
            for (uint32_t p = 0; p < P; ++p)
                if ((int64_t)every[p] != 0) return MC_OK;  // the level ends here; run() reports it through level_info
            stay_bytes = total_max;
        }
        auto answers = [&](uint64_t q) -> int {  // second half of round q: answers travel back, the sender keeps its new states
            const uint32_t t = (uint32_t)(q & 1);
            const uint64_t cap = cap_of(q);
            e.wait(S_COMM, EV_PROBED + t);
            step([&] { return e.wait_keep(t); });  // (on WORK) the slot's previous keep has read back[t] ...
            e.record(EV_TMP, S_WORK);
            e.wait(S_COMM, EV_TMP);                // ... before this exchange overwrites it
            int rc = timed_net(net.all_to_all, ans[t].p, back[t].p, cap);
            if (rc) return rc;
            st.sent_bytes += cap * (P - 1);
            st.fp_answer_bytes += cap * (P - 1);
            e.record(EV_ANS + t, S_COMM);
            e.wait(S_WORK, EV_ANS + t);
            step([&] { return e.keep_pack(t, (const uint8_t *)back[t].p, cap); });  // on the engine's second stream, behind WORK here
            return MC_OK;
        };
        if (rounds) launch(0);
        for (uint64_t r = 0; r < rounds; ++r) {
            const uint32_t s = (uint32_t)(r & 1);
            const uint64_t cap = cap_of(r);
            if (r >= 2) e.wait(S_MAIN, EV_FP + s);  // the exchange of round r-2 has read send[s]
            step([&] { return e.expand_pack(s, (uint64_t *)send[s].p, cap); });  // on MAIN, behind expand r: no host wait
            if (lrc) e.clear_counts((uint64_t *)send[s].p, P, cap);  // a failed rank sends empty buckets
            e.record(EV_PACKED + s, S_MAIN);
            e.wait(S_COMM, EV_PACKED + s);
            if (r >= 2) e.wait(S_COMM, EV_PROBED + s);  // the probes of round r-2 have read recv[s]
            if (net.all_to_all_others) {  // the rank's own bucket is empty by construction: only its count word matters, and the loop writes that
                e.clear_bytes((uint64_t *)recv[s].p + (uint64_t)me * cap, sizeof(uint64_t), S_COMM);
                if ((trc = timed_net(net.all_to_all_others, send[s].p, recv[s].p, cap * 8))) return trc;
            } else if ((trc = timed_net(net.all_to_all, send[s].p, recv[s].p, cap * 8))) {
                return trc;
            }
            st.sent_bytes += cap * 8 * (P - 1);
            st.fp_answer_bytes += cap * 8 * (P - 1);
            e.record(EV_FP + s, S_COMM);
            if (r >= 1 && (trc = answers(r - 1))) return trc;  // issued AFTER fp(r): probes of r-1 ran while fp(r) travelled
            // (only now: round r+1 reuses the engine slot of round r-1, whose keep has just been issued)
            if (r + 1 < rounds) launch(r + 1);  // overlaps the exchanges and probes of rounds r and r+1
            e.wait(S_WORK, EV_FP + s);
            if (r >= 2) e.wait(S_WORK, EV_ANS + s);  // the answers exchange of round r-2 has read ans[s]
            step([&] { return e.probe_pack((const uint64_t *)recv[s].p, cap, (uint8_t *)ans[s].p); });
            if (lrc) e.clear_bytes(ans[s].p, (size_t)P * cap, S_WORK);
            e.record(EV_PROBED + s, S_WORK);
        }
        if (rounds && (trc = answers(rounds - 1))) return trc;
        return MC_OK;
    }

    // ------------------------------------------------------------------ MOVE: host-paced rounds, new states travel to their owners
    int exchange_counts(const uint64_t *mine, std::vector<uint64_t> &from_peers) {
        std::vector<uint64_t> m(P * (size_t)P);
        int rc = timed_net(net.all_gather, mine, m.data(), P * sizeof(uint64_t));
        if (rc) return rc;
        from_peers.resize(P);
        for (uint32_t p = 0; p < P; ++p) from_peers[p] = m[(size_t)p * P + me];
        return MC_OK;
    }
    static constexpr uint32_t PROBE_SEGS = 4;
    uint64_t probe_rot = 0;  // which source goes first, round by round (starts at this rank's number: the owners differ too)
    void comm_after_work() { e.record(EV_TMP, S_WORK); e.wait(S_COMM, EV_TMP); }
    void work_after_comm() { e.record(EV_TMP + 1, S_COMM); e.wait(S_WORK, EV_TMP + 1); }
    int a2a_v(const void *s, const std::vector<uint64_t> &sc, void *r, const std::vector<uint64_t> &rc_, uint64_t elem) {
        std::vector<uint64_t> so(P), sb(P), ro(P), rb(P);
        uint64_t a = 0, b = 0;
        for (uint32_t p = 0; p < P; ++p) {
            so[p] = a; sb[p] = sc[p] * elem; a += sb[p];
            ro[p] = b; rb[p] = rc_[p] * elem; b += rb[p];
            if (p != me) st.sent_bytes += sb[p];
        }
        comm_after_work();
        int rc = timed_net(net.all_to_all_v, s, so.data(), sb.data(), r, ro.data(), rb.data());
        work_after_comm();
        return rc;
    }
    template <class L>
    int move_level(uint64_t rounds, uint64_t move_ch, uint64_t send_cap, size_t W, bool traced, bool keep_local, L &&launch, NetBuf *send, NetBuf *recv, NetBuf *ans, NetBuf *back,
                   NetBuf &states, NetBuf &rstates, NetBuf &parents, NetBuf &rparents) {
        int trc;
        std::vector<uint64_t> counts(P), rcounts, scounts(P), rsc, blocks(P), rblocks(P);
        if (rounds) launch(0);
        for (uint64_t r = 0; r < rounds; ++r) {
            const uint32_t s = (uint32_t)(r & 1);
            std::fill(counts.begin(), counts.end(), 0);
            step([&] { return send[s].need(send_cap * 8); });
            step([&] { return e.expand_finish(s, (uint64_t *)send[s].p, send_cap, counts.data()); });  // waits for expand r only
            if (lrc) std::fill(counts.begin(), counts.end(), 0);
            if (r + 1 < rounds) launch(r + 1);  // overlaps everything below
            if ((trc = exchange_counts(counts.data(), rcounts))) return trc;
            uint64_t n = 0, total = 0, fullest = 0;
            for (uint32_t p = 0; p < P; ++p) {
                n += rcounts[p];
                total += counts[p];
                if (p != me) { fullest = std::max(fullest, counts[p]); st.routed_candidates += counts[p]; st.fp_answer_bytes += 9 * counts[p]; }
            }
            note_fill(fullest, std::min(move_ch, sizes[me] > r * move_ch ? sizes[me] - r * move_ch : 0));
            // (the answers come back into the slot's own buffer when the states stay: the keep of round r reads it on the engine's
            //  second stream while round r+1 is already being exchanged)
            NetBuf &bk = keep_local ? back[s] : back[0];
            step([&] { return recv[0].need(std::max<uint64_t>(n, 1) * 8); });
            step([&] { return ans[0].need(std::max<uint64_t>(n, 1)); });
            if (keep_local) step([&] { return e.wait_keep(s); });  // (on WORK) the slot's previous keep has read back[s] — also before it may be re-allocated
            step([&] { return bk.need(std::max<uint64_t>(total, 1)); });
            // (one agreement for the round's two all-to-alls: all four buffers were asked for above)
            Move mv[2] = {{&send[s], &counts, &recv[0], &rcounts, 8, nullptr, nullptr}, {&ans[0], &rcounts, &bk, &counts, 1, nullptr, nullptr}};
            const int go = agree_moves(mv, 2);
            if (go < 0) return go;
            if (!go && (trc = a2a_v(mv[0].sp, counts, mv[0].rp, rcounts, 8))) return trc;
            if (keep_local && P > 1) {
                // WHO KEEPS A STATE that several ranks generated in the same round is decided by whose candidate reaches the table first.
                // Probed as they lie (source 0's bucket first), the lower ranks win those ties systematically and their frontiers grow
                // level after level (8 ranks on the 10^8-state bench graph: 17.3 M states on rank 0 against 11.1 M on rank 5, every other
                // level a rebalancing one — profiles/r04zc; the fixed-capacity form walks the sources interleaved inside its kernel for
                // the same reason).  Here: each source's bucket in PROBE_SEGS segments, the segments in rotating source order, one launch
                // per segment on the same stream — a tie between two sources falls either way equally often.
                const uint32_t K = n >= (1u << 16) ? PROBE_SEGS : 1;
                std::vector<uint64_t> off(P + 1, 0);
                for (uint32_t p = 0; p < P; ++p) off[p + 1] = off[p] + rcounts[p];
                for (uint32_t k = 0; k < K; ++k)
                    for (uint32_t j = 0; j < P; ++j) {
                        const uint32_t p = (uint32_t)((j + k + probe_rot) % P);
                        const uint64_t a = rcounts[p] * k / K, b = rcounts[p] * (k + 1) / K;
                        if (b > a) step([&] { return e.probe((const uint64_t *)recv[0].p + off[p] + a, b - a, (uint8_t *)ans[0].p + off[p] + a); });
                    }
                ++probe_rot;
            } else {
                step([&] { return e.probe((const uint64_t *)recv[0].p, n, (uint8_t *)ans[0].p); });
            }
            if (!go && (trc = a2a_v(mv[1].sp, rcounts, mv[1].rp, counts, 1))) return trc;
            if (keep_local) {
                // STAY with exact sizes (the default form, include/tlamc.h): the positively answered candidates become states of THIS rank; what crossed
                // xGMI is 9 bytes per routed candidate and the P counts, nothing else — no capacity to guess, no bucket to overflow
                step([&] { return e.keep(s, (const uint8_t *)bk.p); });
                continue;
            }
            // the sender materialises its positively answered candidates, bucketed by owner, as whole 64-state blocks
            std::fill(scounts.begin(), scounts.end(), 0);
            const uint64_t guess = std::min<uint64_t>(total, total / 4 + 4096) + 64ull * P;
            step([&] { return states.need(guess * W); });
            if (!lrc) {
                int rc = e.materialise_slot(s, (const uint8_t *)back[0].p, (uint8_t *)states.p, states.bytes / W, scounts.data());
                if (rc == MC_EARENA) {  // more new states than guessed: size the buffer for the upper bound (every candidate new) and repeat
                    lrc = states.need((total + 64ull * P) * W);
                    if (!lrc) rc = e.materialise_slot(s, (const uint8_t *)back[0].p, (uint8_t *)states.p, states.bytes / W, scounts.data());
                }
                if (!lrc) lrc = rc;
            }
            if (lrc) std::fill(scounts.begin(), scounts.end(), 0);
            if ((trc = exchange_counts(scounts.data(), rsc))) return trc;
            uint64_t rb = 0, moved = 0, rmoved = 0;
            for (uint32_t p = 0; p < P; ++p) {
                blocks[p] = (scounts[p] + 63) / 64;
                rblocks[p] = (rsc[p] + 63) / 64;
                rb += rblocks[p];
                moved += scounts[p];
                rmoved += rsc[p];
            }
            step([&] { return rstates.need(std::max<uint64_t>(rb, 1) * 64 * W); });
            if ((trc = a2a_v_safe(states, blocks, rstates, rblocks, 64 * W))) return trc;
            if (traced) {  // (index on the sending rank << 16 | slot) of every moved state, same owner order, no block padding
                step([&] { return parents.need(std::max<uint64_t>(moved, 1) * 8); });
                step([&] { return rparents.need(std::max<uint64_t>(rmoved, 1) * 8); });
                step([&] { return e.materialise_parents(s, (uint64_t *)parents.p); });
                if ((trc = a2a_v_safe(parents, scounts, rparents, rsc, 8))) return trc;
            }
            uint64_t off = 0, poff = 0;
            for (uint32_t src = 0; src < P; ++src) {  // one bucket per source rank
                if (rsc[src]) {
                    step([&] { return e.ingest((const uint8_t *)rstates.p + off * 64 * W, rsc[src]); });
                    if (traced) step([&] { return e.ingest_parents((const uint64_t *)rparents.p + poff, rsc[src], src); });
                }
                off += rblocks[src];
                poff += rsc[src];
            }
        }
        return MC_OK;
    }
    // What a rank ANNOUNCED it sends it must still move — or every rank must know that it cannot.  A rank whose own buffers are
    // missing (it failed after announcing: an allocation, an engine call) moves the sizes out of / into scratch memory; if even
    // the scratch cannot be had, nobody enters the collective: the ranks agree on that with one small all-gather first (a rank
    // that returned alone would leave its peers waiting in the all-to-all forever — an out-of-memory rank is the likeliest case).
    // The level then runs on with stale buffers and ends at its level_info, where the failed rank's status stops every rank.
    // ONE agreement covers all the all-to-alls whose sizes are known when it is made: the two of an exchange round (fingerprints
    // out, answers back) share it, so a round costs the all-gather of the counts + this one, not three.
    struct Move { NetBuf *s; const std::vector<uint64_t> *sc; NetBuf *r; const std::vector<uint64_t> *rc; uint64_t elem; void *sp, *rp; };
    NetBuf scratch[4];
    // 0: every rank has its buffers (m[i].sp / rp say which); 1: some rank has not — nobody moves anything; < 0: the transport failed
    int agree_moves(Move *m, size_t nm) {
        uint64_t cannot = 0;
        for (size_t i = 0; i < nm && i < 2; ++i) {
            uint64_t need_s = 0, need_r = 0;
            for (uint32_t p = 0; p < P; ++p) { need_s += (*m[i].sc)[p] * m[i].elem; need_r += (*m[i].rc)[p] * m[i].elem; }
            m[i].sp = m[i].s->p;
            m[i].rp = m[i].r->p;
            if (!m[i].sp || m[i].s->bytes < need_s) {
                NetBuf &x = scratch[2 * i];
                x.t = &net;
                if (x.need(std::max<uint64_t>(need_s, 8))) cannot = 1;
                m[i].sp = x.p;
            }
            if (!m[i].rp || m[i].r->bytes < need_r) {
                NetBuf &x = scratch[2 * i + 1];
                x.t = &net;
                if (x.need(std::max<uint64_t>(need_r, 8))) cannot = 1;
                m[i].rp = x.p;
            }
        }
        if (cannot && !lrc) lrc = MC_EHIP;
        std::vector<uint64_t> every(P);
        const int trc = timed_net(net.all_gather, &cannot, every.data(), sizeof cannot);
        if (trc) return trc < 0 ? trc : -trc;
        for (uint32_t p = 0; p < P; ++p)
            if (every[p]) return 1;  // nobody moves anything; the failed rank's status travels with the level's all-gather
        return 0;
    }
    int a2a_v_safe(NetBuf &s, const std::vector<uint64_t> &sc, NetBuf &r, const std::vector<uint64_t> &rc_, uint64_t elem) {
        Move m{&s, &sc, &r, &rc_, elem, nullptr, nullptr};
        const int go = agree_moves(&m, 1);
        if (go) return go < 0 ? go : MC_OK;
        return a2a_v(m.sp, sc, m.rp, rc_, elem);
    }

    // ------------------------------------------------------------------ counterexample across ranks
    int trace(uint8_t *states_out, int32_t *slots_out, size_t *n_inout, int32_t *final_slot) {
        const size_t W = e.state_bytes(), cap = *n_inout;
        *n_inout = 0;
        if (final_slot) *final_slot = -1;
        struct Viol { int64_t found, idx, slot, verdict, invariant, status; } mine{0, 0, 0, 0, -1, 0};
        {
            int32_t found = 0, v = 0, inv = -1;
            uint64_t idx = 0;
            uint32_t slot = 0;
            int rc = e.violation(&found, &idx, &slot, &v, &inv);
            mine = Viol{found, (int64_t)idx, (int64_t)slot, v, inv, rc};
        }
        std::vector<Viol> every(P);
        int trc = timed_net(net.all_gather, &mine, every.data(), sizeof mine);
        if (trc) return trc;
        int owner = -1;
        for (uint32_t p = 0; p < P; ++p) {
            if (every[p].status) return (int)every[p].status;
            if (owner < 0 && every[p].found) owner = (int)p;
        }
        if (owner < 0) return MC_OK;
        const Viol v = every[owner];
        // one record per step, gathered from every rank; only the rank that holds the state fills it
        const size_t rec = 32 + W;
        std::vector<uint8_t> mine_rec(rec), all_rec(rec * P);
        std::vector<std::vector<uint8_t>> st_rev;
        std::vector<int32_t> slot_rev;
        uint32_t cur_rank = (uint32_t)owner;
        // An invariant violated by an INITIAL state: the violation's index is the ordinal of that state in Init's enumeration, not an
        // arena index (an owner-filtered or deduplicated Init stores it elsewhere, or nowhere on this rank) — the owner rebuilds the
        // state from the ordinal (fetch with FETCH_INIT set) and the behaviour is that one state.
        uint64_t cur_idx = (uint32_t)v.slot == SLOT_INIT ? (FETCH_INIT | (uint64_t)v.idx) : (uint64_t)v.idx;
        for (int guard = 0; guard < (1 << 16); ++guard) {
            memset(mine_rec.data(), 0, rec);
            if (cur_rank == me) {
                uint32_t prank = 0, pslot = 0;
                uint64_t pidx = 0;
                int64_t rc = e.fetch(cur_idx, mine_rec.data() + 32, &prank, &pidx, &pslot);
                int64_t hdr[4] = {rc, (int64_t)prank, (int64_t)pidx, (int64_t)pslot};
                memcpy(mine_rec.data(), hdr, sizeof hdr);
            }
            if ((trc = timed_net(net.all_gather, mine_rec.data(), all_rec.data(), rec))) return trc;
            int64_t hdr[4];
            memcpy(hdr, all_rec.data() + (size_t)cur_rank * rec, sizeof hdr);
            if (hdr[0]) return (int)hdr[0];
            const uint32_t pslot = (uint32_t)hdr[3];
            if (pslot == SLOT_COPY) {  // the replicated prefix copied the state into this rank's slice: not a step
                cur_idx = (uint64_t)hdr[2];
                continue;
            }
            st_rev.emplace_back(all_rec.begin() + (long)((size_t)cur_rank * rec + 32), all_rec.begin() + (long)((size_t)cur_rank * rec + 32 + W));
            slot_rev.push_back((int32_t)pslot);
            if ((uint64_t)hdr[2] == NO_PARENT) break;
            cur_rank = (uint32_t)hdr[1];
            cur_idx = (uint64_t)hdr[2];
        }
        const size_t n = st_rev.size();
        if (n > cap) { mc_set_error_internal("mc_shard_trace: trace buffer too small"); return MC_EBADCFG; }
        for (size_t k = 0; k < n; ++k) {
            memcpy(states_out + k * W, st_rev[n - 1 - k].data(), W);
            slots_out[k] = k == 0 ? -1 : slot_rev[n - 1 - k];   // slot_rev[j] produced st_rev[j] from st_rev[j + 1]
        }
        *n_inout = n;
        // an invariant violated by a SUCCESSOR: that state is stored nowhere, the caller rebuilds it from its parent.  (A failed
        // Assert / an evaluation error has no successor: TLC's behaviour ends at the state the action was taken from.)
        if (final_slot && v.verdict == MC_V_INVARIANT && (uint32_t)v.slot != SLOT_NONE && (uint32_t)v.slot != SLOT_PARENT && (uint32_t)v.slot != SLOT_INIT)
            *final_slot = (int32_t)v.slot;
        return MC_OK;
    }
};

// The whole search.  A round whose exchange bucket was too small for its candidates (more in-model successors per state than
// packed_fanout / move_fanout allow for) fails the level on EVERY rank with MC_EROUTE — the statuses travel with the level's
// all-gather, nothing was truncated and nobody is left inside a collective — and the search is started over from Init with twice
// the allowance: the engine's begin / begin_replicated clear the seen-set and the arena as they do for every run, the exchange
// buffers belong to the Loop.  (A run restored from checkpoints starts over from Init too: correct, only slower.)  Every rank takes
// the same decision from the same status, so the restarts need no further agreement.
template <class Ops>
int run_restarting(Ops &ops, const mc_transport &t, const mc_shard_opts &o, mc_result *out) {
    mc_shard_opts cur = o;
    uint64_t restarts = 0;
    for (;;) {
        Loop<Ops> loop(ops, t);
        const int rc = loop.run(cur, out);
        if (rc != MC_EROUTE || restarts == 5) {
            if (o.stats) o.stats->restarts = restarts;
            return rc;
        }
        // Was it a level whose buckets were sized from the previous level's measured fill (it grew faster than cap_safety_pct allows)?
        // Every rank left the loop at the same level with the same sizing decision (it depends on all-gathered values only), so
        // every rank takes the same branch here.
        const bool any = loop.level_measured;
        if (any) cur.flags |= MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS;  // (a measured bucket does not grow with the allowance)
        cur.packed_fanout = 2 * (cur.packed_fanout ? cur.packed_fanout : 16);
        cur.move_fanout = 2 * (cur.move_fanout ? cur.move_fanout : 32);
        ++restarts;
    }
}

}  // namespace mc_shard
#endif
