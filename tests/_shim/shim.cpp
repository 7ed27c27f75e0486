// tests/_shim/shim.cpp — TEST-ONLY host build of the device lowerings (tla_rust_amd/csrc/spec_*.h).
//
// There is no GPU in the development container, so the spec lowerings (which are MC_HD
// host+device code) are compiled here with g++ and driven by a trivial sequential BFS with a
// std::unordered_set of 64-bit fingerprints.  This lets `pytest -m "not gpu"` compare the
// LOWERING (guards, successor construction, incremental fingerprints, state text) with the
// independent oracle under oracle/ before a single GPU-minute is spent.  It is NOT part of
// the product: libtlamc.so contains no CPU back-end and fails without a HIP device.
//
// The same step functions (expand / probe / materialise / ingest) are exported so the
// multi-rank exchange logic of tla_rust_amd/sharded.py can be exercised with world_size-2
// gloo tests on CPU.
#include "../../tla_rust_amd/csrc/spec_registry.h"
#include "../../tla_rust_amd/csrc/pcal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <unordered_set>
#include <vector>
#include <tuple>
#include <algorithm>

using namespace mc;

// open-addressing fingerprint set (the std::unordered_set it replaces spent minutes in malloc)
struct FpSet {
    std::vector<uint64_t> tab;
    size_t n = 0;
    FpSet() : tab(1 << 16, 0) {}
    void clear() { tab.assign(1 << 16, 0); n = 0; }
    void grow() {
        std::vector<uint64_t> old;
        old.swap(tab);
        tab.assign(old.size() * 2, 0);
        for (uint64_t v : old) if (v) put(v);
    }
    bool put(uint64_t fp) {
        size_t h = fp & (tab.size() - 1);
        while (tab[h]) { if (tab[h] == fp) return false; h = (h + 1) & (tab.size() - 1); }
        tab[h] = fp;
        return true;
    }
    bool contains(uint64_t fp) const {
        size_t h = fp & (tab.size() - 1);
        while (tab[h]) { if (tab[h] == fp) return true; h = (h + 1) & (tab.size() - 1); }
        return false;
    }
    // std::unordered_set-like insert: .second = newly inserted
    struct R { int first; bool second; };
    R insert(uint64_t fp) {
        if ((n + 1) * 2 > tab.size()) grow();
        const bool is_new = put(fp);
        n += is_new;
        return R{0, is_new};
    }
};

struct ShimResult {
    uint64_t distinct, generated, queue_left;
    uint32_t depth;
    int32_t verdict, violated_invariant;
    uint32_t trace_len, levels;
    uint64_t fp_mismatch;  // states whose stored fingerprint != full recomputation (must be 0)
    uint64_t level_distinct[MC_MAX_LEVELS];
};

template <class S>
static uint64_t stored_fp(const typename S::Params &p, const uint64_t *w) { return S::fp_of(p, CWordRef{w, 1}); }

// by-pairs interface (specs with PAIR_FAMILIES, engine_pairs.h): for every state the lowering visits, (1) the guard mask names every
// enabled slot (and ONLY enabled slots, the spec's declared inexact slots aside: S::guard_is_exact), (2) every slot lies in exactly one
// family and one round, (3) eval_pair<F> from the Summary gives eval's status and fingerprint, (4) write_pair gives apply's row
template <class S, class = void>
struct PairCheck {
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &, typename S::Local &, Ref, int) { return 0; }
};
template <class S>
struct PairCheck<S, decltype((void)S::PAIR_FAMILIES)> {
    template <int F, class Ref>
    static unsigned run(int fam, const typename S::Params &p, const typename S::Summary &q, Ref s, int slot, uint64_t &fp, typename S::PairOut &o) {
        if constexpr (F < S::PAIR_FAMILIES) {
            if (fam == F) return S::template eval_pair<F>(p, q, s, slot, fp, o);
            return run<F + 1>(fam, p, q, s, slot, fp, o);
        } else {
            return 0;
        }
    }
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &p, typename S::Local &l, Ref s, int ns) {
        uint64_t bad = 0, glo = 0, ghi = 0;
        S::guards(p, l, glo, ghi);
        typename S::Summary q;
        S::summarize(l, q);
        uint64_t staged[S::MAX_WORDS];  // the kernel's LDS copy of the row: word W_PAIR_BASE holds the pair base
        for (int w = 0; w < S::MAX_WORDS; w++) staged[w] = w < S::words(p) ? s.get(w) : 0;
        staged[S::W_PAIR_BASE] = S::pair_base(p, l, s);
        const CWordRef srow{staged, 1};
        unsigned per_round[S::PAIR_ROUNDS] = {};
        for (int slot = 0; slot < ns; slot++) {
            const bool g = slot < 64 ? (glo >> slot & 1u) : (ghi >> (slot - 64) & 1u);
            int fam = -1, nfam = 0, nround = 0;
            for (int f = 0; f < S::PAIR_FAMILIES; f++) {
                const auto m = S::family_mask(f);
                if (slot < 64 ? (m.lo >> slot & 1u) : (m.hi >> (slot - 64) & 1u)) { fam = f; nfam++; }
            }
            for (int r = 0; r < S::PAIR_ROUNDS; r++) {
                const auto m = S::round_mask(r);
                if (slot < 64 ? (m.lo >> slot & 1u) : (m.hi >> (slot - 64) & 1u)) { nround++; per_round[r]++; }
            }
            if (nfam != 1 || nround != 1 || fam != S::slot_family(slot)) bad++;
            uint64_t f0 = 0, f1 = 0;
            const unsigned st0 = S::eval(p, l, s, slot, f0);
            if ((st0 & ST_ENABLED) && !g) bad++;                                  // an enabled slot the guards miss: a lost successor
            if (!(st0 & ST_ENABLED) && g && S::guard_is_exact(slot)) bad++;       // (an idle lane, not an error of the search — but the contract says exact)
            typename S::PairOut o;
            const unsigned st1 = run<0>(fam, p, q, srow, slot, f1, o);
            if (st0 != st1) { bad++; continue; }
            if (!(st0 & ST_ENABLED) || (st0 & ST_OVERFLOW)) continue;
            if (f0 != f1) bad++;
            uint64_t a[S::MAX_WORDS], b[S::MAX_WORDS];
            S::apply(p, s, slot, WordRef{a, 1});
            for (int w = 0; w < S::MAX_WORDS; w++) b[w] = ~0ull;
            S::write_pair(p, srow, o, WordRef{b, 1});
            for (int w = 0; w < S::words(p); w++) if (a[w] != b[w]) { bad++; break; }
        }
        for (int r = 0; r < S::PAIR_ROUNDS; r++) if (per_round[r] > (unsigned)S::PAIR_ROUND_SLOTS) bad++;
        return bad;
    }
};

// specs whose kernels evaluate the invariants of a stored state from what its last step can have changed (S::parent_status_step): the
// verdict must be parent_status's on EVERY state the search expands or stops on — violating models included (the states of the level a
// violation is found on were all generated from states that passed)
template <class S, class = void>
struct StepStatusCheck {
    template <class Ref>
    static uint64_t mismatch(const typename S::Params &, const typename S::Local &, Ref, unsigned) { return 0; }
};
template <class S>
struct StepStatusCheck<S, decltype((void)S::STEP_STATUS)> {
    template <class Ref>
    static uint64_t mismatch(const typename S::Params &p, const typename S::Local &l, Ref s, unsigned full) { return S::parent_status_step(p, l, s) != full ? 1 : 0; }
};

// expand-by-family interface (specs with NFAM): for every (state, slot) the family-pruned evaluation through the
// guard must reproduce exactly what the generic evaluation does
template <class S, class = void>
struct FamCheck {
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &, typename S::Local &, Ref, int) { return 0; }
};
template <class S>
struct FamCheck<S, decltype((void)S::NFAM)> {
    // analysis aid (SHIM_FAMSTATS=1): lane utilisation of the by-family expand for several tile sizes
    struct Stats {
        static constexpr int NT = 4;
        uint64_t tile[NT][32] = {}, n_in_tile[NT] = {}, batches[NT] = {}, pairs = 0, fam_total[32] = {};
        bool on = getenv("SHIM_FAMSTATS") != nullptr;
        void state_done(const unsigned *c) {
            for (int t = 0; t < NT; t++) {
                for (int f = 0; f < S::NFAM; f++) tile[t][f] += c[f];
                if (++n_in_tile[t] == (64u << t)) flush(t);
            }
            for (int f = 0; f < S::NFAM; f++) { pairs += c[f]; fam_total[f] += c[f]; }
        }
        void flush(int t) {
            for (int f = 0; f < S::NFAM; f++) { batches[t] += (tile[t][f] + 63) / 64; tile[t][f] = 0; }
            n_in_tile[t] = 0;
        }
        ~Stats() {
            if (!on || !pairs) return;
            for (int t = 0; t < NT; t++) {
                flush(t);
                fprintf(stderr, "famstats: tile %4u states: %llu batches, lane utilisation %.3f\n", 64u << t,
                        (unsigned long long)batches[t], (double)pairs / (64.0 * (double)batches[t]));
            }
            for (int f = 0; f < S::NFAM; f++) fprintf(stderr, "famstats: family %2d: %.4f of pairs\n", f, (double)fam_total[f] / (double)pairs);
        }
    };
    static Stats &stats() { static Stats st; return st; }
    template <int F, class Ref>
    static unsigned run(int fam, const typename S::Params &p, const typename S::Summary &q, Ref s, int slot, uint64_t &fp) {
        if constexpr (F < S::NFAM) {
            if (fam == F) return S::template eval_pair<F>(p, q, s, slot, fp);
            return run<F + 1>(fam, p, q, s, slot, fp);
        } else {
            return 0;
        }
    }
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &p, typename S::Local &l, Ref s, int ns) {
        typename S::Guards g;
        S::guards(p, l, g);
        typename S::Summary q;
        S::summarize(l, q);
        uint64_t bad = 0;
        unsigned cnt[32] = {};
        typename S::Guards g2c;
        {   // load_expand (the kernel's phase A: the whole row at once, in-flight mask) == load + guards
            typename S::Local l2;
            typename S::Guards g2;
            S::load_expand(p, s, l2, g2);
            g2c = g2;
            bool same = l2.fp == l.fp && l2.glob == l.glob && l2.clog == l.clog && l2.nm == l.nm && l2.inflight == l.inflight &&
                        l2.addmask == l.addmask && l2.nadd == l.nadd && l2.add_fp == l.add_fp && l2.vany == l.vany && l2.dig == l.dig &&
                        l2.sig.w0 == l.sig.w0 && l2.sig.w1 == l.sig.w1 && l2.sig.w2 == l.sig.w2 && l2.sig.w3 == l.sig.w3 &&
                        g2.fixed == g.fixed && g2.fixed_hi == g.fixed_hi;
            for (int i = 0; i < p.n; i++) same = same && l2.sv.get(i) == l.sv.get(i) && l2.log.get(i) == l.log.get(i) && l2.vlh.get(i) == l.vlh.get(i);
            for (int k = 0; k < l.nm && k < S::GUARD_SLOTS; k++)   // the in-flight mask names exactly the messages with count > 0
                same = same && ((S::inflight_slots(g2) >> k & 1u) != 0) == (S::m_count(S::rd_msg(s, k)) > 0);
            if (!same) bad++;
        }
        for (int slot = 0; slot < ns; slot++) {
            uint64_t f0 = 0, f1 = 0;
            const unsigned st0 = S::eval(p, l, s, slot, f0);
            {   // generated-only shortcuts of the expand kernel: a slot they name must be enabled, never storable, and raise nothing
                const bool go = slot < S::FIX ? (S::fixed_bit(g, slot) && S::fixed_generated_only(p, l.inflight, slot)) : S::message_generated_only(p, l, s, slot);
                if (go && (!(st0 & ST_ENABLED) || !(st0 & (ST_OUT_OF_MODEL | ST_SELFLOOP)) || (st0 & (ST_ASSERT | ST_SPECERR | ST_INVARIANT | ST_OVERFLOW)))) bad++;
            }
            int fam = -1;
            // queued: the sparse fixed slots (the dense pairs, slots < DENSE_SLOTS, and the message slots are evaluated by eval itself)
            const bool queued = slot >= S::DENSE_SLOTS && slot < S::FIX;
            if (queued && S::fixed_bit(g, slot)) fam = S::fixed_family(slot);
            const unsigned st1 = fam >= 0 ? run<0>(fam, p, q, s, slot, f1) : 0u;
            // (a stuttering step carries no fingerprint through the by-family path: the flag alone drops it)
            if (queued && (st0 != st1 || ((st0 & ST_ENABLED) && !(st0 & ST_SELFLOOP) && f0 != f1))) bad++;
            if (!queued && slot >= S::FIX && (st0 & ST_ENABLED) && !(S::inflight_slots(g2c) >> ((slot - S::FIX) / 3) & 1u) && (slot - S::FIX) / 3 < S::GUARD_SLOTS) bad++;  // enabled but not in flight
            if (st0 & ST_SELFLOOP) {  // claimed stuttering step: apply() must reproduce the parent word for word
                uint64_t a[S::MAX_WORDS];
                S::apply(p, s, slot, WordRef{a, 1});
                for (int w = 0; w < S::words(p); w++) if (a[w] != s.get(w)) { bad++; break; }
            }
            if (queued && (st1 & ST_ENABLED) && !(st1 & (ST_OUT_OF_MODEL | ST_OVERFLOW | ST_SPECERR | ST_ASSERT | ST_SELFLOOP))) {
                uint64_t a[S::MAX_WORDS];
                S::apply(p, s, slot, WordRef{a, 1});
                if (S::fp_of(p, CWordRef{a, 1}) != f1) bad++;  // the successor carries the fingerprint the by-family evaluation announced
                uint64_t c[S::MAX_WORDS];  // k_materialise with the fingerprint handed over by the expand kernel
                S::apply_known_fp(p, s, slot, f1, WordRef{c, 1});
                if (memcmp(a, c, sizeof(uint64_t) * (size_t)S::words(p)) != 0) bad++;
            }
            if ((st0 & ST_ENABLED) && !(st0 & (ST_OUT_OF_MODEL | ST_OVERFLOW | ST_SPECERR | ST_ASSERT | ST_SELFLOOP))) {
                // the in-wave writer of the expand kernel (round 4): row copy + action + patch, starting from the parent's Summary
                // — EVERY kind of slot, word for word what apply() writes
                uint64_t a[S::MAX_WORDS], c[S::MAX_WORDS];
                S::apply(p, s, slot, WordRef{a, 1});
                for (int w = 0; w < S::MAX_WORDS; w++) c[w] = 0xdeadbeefdeadbeefull;
                S::apply_summary_patch(p, q, s, slot, f0, WordRef{c, 1});
                if (memcmp(a, c, sizeof(uint64_t) * (size_t)S::words(p)) != 0) bad++;
            }
            if (fam >= 0) cnt[fam]++;
        }
        if (stats().on) stats().state_done(cnt);
        return bad;
    }
};

// dense slot pairs (S::eval_dense: two slots of one server evaluated together with shared hash terms) must give
// exactly what the generic evaluation of the two slots gives
template <class S, class = void>
struct DenseCheck {
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &, typename S::Local &, Ref) { return 0; }
};
template <class S>
struct DenseCheck<S, decltype((void)S::DENSE_PAIRS)> {
    template <class Ref>
    static uint64_t mismatches(const typename S::Params &p, typename S::Local &l, Ref s) {
        uint64_t bad = 0;
        for (int i = 0; i < S::DENSE_PAIRS; i++) {
            unsigned st[2] = {0, 0};
            uint64_t fp[2] = {0, 0};
            S::eval_dense(p, l, s, i, st[0], fp[0], st[1], fp[1]);
            for (int h = 0; h < 2; h++) {
                uint64_t f0 = 0;
                const unsigned st0 = S::eval(p, l, s, S::dense_slot(i, h), f0);
                if (st0 != st[h] || ((st0 & ST_ENABLED) && f0 != fp[h])) bad++;
            }
        }
        return bad;
    }
};

template <class S, class = void>
struct FpCheck {
    static bool ok(const typename S::Params &, const uint64_t *) { return true; }
};
template <class S>
struct FpCheck<S, decltype((void)S::W_FP)> {
    static bool ok(const typename S::Params &p, const uint64_t *w) { return S::fp_recompute(p, CWordRef{w, 1}) == w[S::W_FP]; }
};

template <class S>
static int run(S, const typename S::Params &prm, uint64_t max_levels, uint64_t max_distinct, int check_deadlock,
               const char *dump_path, ShimResult *r) {
    const int W = S::words(prm);
    memset(r, 0, sizeof *r);
    r->violated_invariant = -1;
    std::vector<uint64_t> cur, next;   // only two BFS levels are resident
    FpSet seen;
    FILE *dump = dump_path ? fopen(dump_path, "w") : nullptr;
    std::vector<char> txt(1 << 16);
    auto add_state = [&](const uint64_t *w, uint32_t level) {
        next.insert(next.end(), w, w + W);
        r->distinct++;
        r->level_distinct[level - 1]++;
        if (!FpCheck<S>::ok(prm, w)) r->fp_mismatch++;
        if (dump) {
            int k = S::format(prm, w, txt.data(), txt.size());
            for (int i = 0; i < k; i++) if (txt[i] == '\n') txt[i] = ' ';
            fprintf(dump, "L%u %.*s\n", level, k, txt.data());
        }
    };
    auto violation = [&](unsigned st, uint32_t trace_len) {
        if (r->verdict) return;
        r->verdict = (st & ST_ASSERT) ? MC_V_ASSERT : (st & ST_SPECERR) ? MC_V_SPECERR : MC_V_INVARIANT;
        if (r->verdict == MC_V_INVARIANT) r->violated_invariant = (int)(st >> 8 & 255);
        r->trace_len = trace_len;
    };
    uint64_t tmp[S::MAX_WORDS];
    const uint64_t ninit = S::num_init(prm);
    for (uint64_t k = 0; k < ninit; k++) {
        S::init(prm, k, WordRef{tmp, 1});
        r->generated++;
        const unsigned st = S::init_status(prm, CWordRef{tmp, 1});
        if (st & ST_INVARIANT) violation(st, 1);
        if (st & ST_OUT_OF_MODEL) continue;
        if (seen.insert(stored_fp<S>(prm, tmp)).second) add_state(tmp, 1);
    }
    cur.swap(next);
    uint32_t level = 1;
    int budget = 0;
    const bool kindstats = getenv("SHIM_KINDSTATS") != nullptr;
    uint64_t kind_cnt[16][4] = {};
    struct KindReport {
        const bool &on; uint64_t (&c)[16][4];
        ~KindReport() {
            if (!on) return;
            for (int a = 0; a < 16; a++)
                if (c[a][0]) fprintf(stderr, "kindstats: action %2d %-20s generated %12llu  stutter %12llu  out-of-model %12llu  already seen %12llu  new %12llu\n", a,
                                     S::action_name(a), (unsigned long long)c[a][0], (unsigned long long)c[a][1], (unsigned long long)c[a][2],
                                     (unsigned long long)c[a][3], (unsigned long long)(c[a][0] - c[a][1] - c[a][2] - c[a][3]));
        }
    } kind_report{kindstats, kind_cnt};
    while (!cur.empty()) {
        if (r->verdict) break;
        if (max_levels && level >= max_levels) { budget = 1; break; }
        if (max_distinct && r->distinct >= max_distinct) { budget = 1; break; }
        const uint64_t nstates = cur.size() / (size_t)W;
        for (uint64_t i = 0; i < nstates; i++) {
            CWordRef s{&cur[i * W], 1};
            typename S::Local loc;
            S::load(prm, s, loc);
            const int ns = S::nslots(prm, loc);
            const unsigned ps = S::parent_status(prm, loc, s);
            if (ps & ST_INVARIANT) violation(ps, level);
            r->fp_mismatch += StepStatusCheck<S>::mismatch(prm, loc, s, ps);
            r->fp_mismatch += FamCheck<S>::mismatches(prm, loc, s, ns);
            r->fp_mismatch += PairCheck<S>::mismatches(prm, loc, s, ns);
            r->fp_mismatch += DenseCheck<S>::mismatches(prm, loc, s);
            uint64_t nsucc = 0;
            for (int slot = 0; slot < ns; slot++) {
                uint64_t fp = 0;
                const unsigned st = S::eval(prm, loc, s, slot, fp);
                if (!(st & ST_ENABLED)) continue;
                nsucc++;
                r->generated++;
                if (kindstats) {  // analysis aid (SHIM_KINDSTATS=1): what becomes of the successors of each action kind
                    const int a = S::action_of(prm, &cur[i * W], slot);
                    uint64_t *k = kind_cnt[a >= 0 && a < 15 ? a : 15];
                    k[0]++;
                    if (st & ST_SELFLOOP) k[1]++;
                    else if (st & ST_OUT_OF_MODEL) k[2]++;
                    else if (seen.contains(fp)) k[3]++;
                }
                if (st & ST_OVERFLOW) { if (dump) fclose(dump); return MC_EOVERFLOW; }
                if (st & (ST_ASSERT | ST_SPECERR)) { violation(st, level); continue; }
                if (st & ST_INVARIANT) violation(st, level + 1);
                if (st & ST_OUT_OF_MODEL) continue;
                if (seen.insert(fp).second) {
                    S::apply(prm, s, slot, WordRef{tmp, 1});
                    if (stored_fp<S>(prm, tmp) != fp) r->fp_mismatch++;
                    add_state(tmp, level + 1);
                }
            }
            if (nsucc == 0 && check_deadlock && !r->verdict) { r->verdict = MC_V_DEADLOCK; r->trace_len = level; }
        }
        cur.clear();
        cur.swap(next);
        if (!cur.empty()) level++;
        if (level >= MC_MAX_LEVELS) break;
    }
    if (!r->verdict && !cur.empty()) {  // engine.hip k_check_frontier: check-on-expand invariants of the level the run stops on
        const uint64_t nstates = cur.size() / (size_t)W;
        for (uint64_t i = 0; i < nstates && !r->verdict; i++) {
            CWordRef s{&cur[i * W], 1};
            typename S::Local loc;
            S::load(prm, s, loc);
            const unsigned ps = S::parent_status(prm, loc, s);
            r->fp_mismatch += StepStatusCheck<S>::mismatch(prm, loc, s, ps);
            if (ps & ST_INVARIANT) violation(ps, level);
        }
    }
    r->depth = level;
    r->levels = level;
    r->queue_left = cur.size() / (size_t)W;
    if (!r->verdict && budget) r->verdict = MC_V_BUDGET;
    if (dump) fclose(dump);
    return 0;
}

// TEST DIAGNOSTIC: a native fault inside the host build prints its own backtrace before Python's faulthandler runs
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
struct sigaction g_prev_segv;
char g_altstack[1 << 16];
void shim_segv(int sig, siginfo_t *info, void *ctx) {
    static const char msg[] = "\n[shim] SIGSEGV backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    void *bt[64];
    const int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    char line[128];
    const int k = snprintf(line, sizeof line, "[shim] fault address %p\n", info ? info->si_addr : nullptr);
    (void)!write(2, line, (size_t)k);
    if (g_prev_segv.sa_flags & SA_SIGINFO) { if (g_prev_segv.sa_sigaction) g_prev_segv.sa_sigaction(sig, info, ctx); }
    else if (g_prev_segv.sa_handler && g_prev_segv.sa_handler != SIG_DFL && g_prev_segv.sa_handler != SIG_IGN) g_prev_segv.sa_handler(sig);
    signal(SIGSEGV, SIG_DFL);
    raise(SIGSEGV);
}
struct InstallSegv {
    InstallSegv() {
        if (!getenv("TLAMC_SHIM_BACKTRACE")) return;
        stack_t ss{};
        ss.ss_sp = g_altstack; ss.ss_size = sizeof g_altstack;
        sigaltstack(&ss, nullptr);
        struct sigaction sa{};
        sa.sa_sigaction = shim_segv;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, &g_prev_segv);
    }
} g_install_segv;
}  // namespace

extern "C" int shim_run(const mc_spec_desc *d, uint64_t max_levels, uint64_t max_distinct, int check_deadlock,
                        const char *dump_path, ShimResult *r) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) { return run(spec, prm, max_levels, max_distinct, check_deadlock, dump_path, r); });
}

// ------------------------------------------------------------------------------------------
// PlusCal front-end on the host (pcal.cpp / pcal_compile.cpp are linked into the shim): translate, compile,
// and describe a program so that shim_run / the shard emulation execute it like any other lowering.
static std::string g_pcal_error;
// the per-element hash of the additive fingerprints (mc_common.h), for tests/test_hash_quality.py
extern "C" void shim_hmum(const uint64_t *x, uint64_t n, uint64_t salt, uint64_t *out) {
    for (uint64_t i = 0; i < n; i++) out[i] = hmum(x[i], salt);
}
extern "C" const char *shim_pcal_error() { return g_pcal_error.c_str(); }
extern "C" int shim_pcal_translate(const char *tla_text, char *out, size_t cap) {
    pcal::Module m;
    const std::string text(tla_text);
    g_pcal_error = pcal::parse_module(text, m);
    if (!g_pcal_error.empty()) return -1;
    const std::string tr = pcal::transpile_text(text, m);
    if (tr.find("\\* TRANSLATION ERROR: ") != std::string::npos) { g_pcal_error = tr; return -1; }
    if (out && cap) { const size_t n = tr.size() < cap ? tr.size() : cap - 1; memcpy(out, tr.data(), n); out[n] = 0; }
    return (int)tr.size();
}
// invariants: comma separated names; constants: "N=3,M=2" (integers only)
extern "C" void *shim_program_compile2(const char *tla_text, const char *invariants, const char *constants, const char *constraints);
extern "C" void *shim_program_compile(const char *tla_text, const char *invariants, const char *constants) {
    return shim_program_compile2(tla_text, invariants, constants, "");
}
extern "C" void *shim_program_compile2(const char *tla_text, const char *invariants, const char *constants, const char *constraints) {
    pcal::Config cf;
    auto split = [](const char *s, char sep) {
        std::vector<std::string> v;
        std::string cur;
        for (const char *p = s ? s : ""; ; p++) {
            if (*p == sep || !*p) { if (!cur.empty()) v.push_back(cur); cur.clear(); if (!*p) break; }
            else if (*p != ' ') cur += *p;
        }
        return v;
    };
    cf.invariants = split(invariants, ',');
    cf.constraints = split(constraints, ',');
    for (const auto &kv : split(constants, ',')) {
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) { g_pcal_error = "bad constant " + kv; return nullptr; }
        pcal::ConstVal v;
        v.k = pcal::ConstVal::INT;
        v.i = atoll(kv.c_str() + eq + 1);
        cf.constants.push_back({kv.substr(0, eq), v});
    }
    pcal::Module m;
    const std::string text(tla_text);
    g_pcal_error = pcal::parse_module(text, m);
    if (!g_pcal_error.empty()) return nullptr;
    auto *P = new pcal::Program();
    g_pcal_error = pcal::compile(m, text, cf, *P);
    if (!g_pcal_error.empty()) { delete P; return nullptr; }
    return P;
}
extern "C" void shim_program_free(void *p) { delete (pcal::Program *)p; }
extern "C" long pcal_codegen_text(const pcal::Program *p, char *buf, size_t cap);   // tla_rust_amd/csrc/pcal_codegen.cpp (linked as is, like the front-end)
extern "C" long shim_program_codegen(void *p, char *buf, size_t cap) { return pcal_codegen_text((const pcal::Program *)p, buf, cap); }
extern "C" const char *shim_program_translated(void *p) { return ((pcal::Program *)p)->translated.c_str(); }

// the DEVICE lowering's invariants on a hand-made history (known-answer tests of the reference's specs):
// events = n x {op, txn, key, ver, reason}; returns the ST_* status of Spec::parent_status with `inv_mask`
extern "C" unsigned shim_ssi_history_status(int nt, int nk, int inv_mask, int textbook, const int *events, int n) {
    SsiParams p{nt, nk, inv_mask, 0, textbook};
    uint64_t w[SpecSsi::MAX_WORDS] = {0};
    w[SpecSsi::W_META] = (uint64_t)n;  // Len(history); no locks, no conflicts
    for (int t = 0; t < SpecSsi::NT; t++) w[SpecSsi::W_META] = SpecSsi::m_set_txn(w[SpecSsi::W_META], t, SpecSsi::mk_txn(0, SpecSsi::NOLOCK, 0, 0, 0));
    for (int i = 0; i < n; i++) {
        const uint64_t e = SpecSsi::mk_event(events[5 * i], events[5 * i + 1], events[5 * i + 2], events[5 * i + 3], events[5 * i + 4]);
        w[SpecSsi::W_H0 + i / 4] |= e << (16 * (i % 4));
    }
    SpecSsi::Local l;
    SpecSsi::load(p, CWordRef{w, 1}, l);
    return SpecSsi::parent_status(p, l, CWordRef{w, 1});
}

// Voting census through the DEVICE lowering (spec_paxos.h): every type-correct packed state — out = {how many, how many satisfy Inv
// (parent_status), successors generated from those (enabled slots), successors violating Inv}; the oracle's oracle_voting_census
// and oracle/tlaplus.py on MCVoting's MCSpecI configuration give the same four numbers
extern "C" int shim_voting_census(const mc_spec_desc *d, uint64_t out[4]) {
    PaxosParams p;
    if (d->spec_id != MC_SPEC_PAXOS || SpecPaxos::make_params(d->params, d->nparams, p) || p.kind != 1 || p.sym) return -1;
    const int vbits = p.nb * p.nv;
    const uint64_t per = ((uint64_t)1 << vbits) * (uint64_t)(p.nb + 1);
    uint64_t total = 1;
    for (int a = 0; a < p.na; a++) total *= per;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (uint64_t k = 0; k < total; k++) {
        uint64_t w[SpecPaxos::MAX_WORDS] = {0}, t[SpecPaxos::MAX_WORDS];
        uint64_t r = k;
        for (int a = 0; a < p.na; a++) {
            const uint64_t dgt = r % per;
            r /= per;
            w[1 + a] = (dgt % (uint64_t)(p.nb + 1)) | (dgt / (uint64_t)(p.nb + 1)) << p.o_2b;  // maxBal + 1 | votes
        }
        out[0]++;
        SpecPaxos::Local l;
        SpecPaxos::load(p, CWordRef{w, 1}, l);
        if (SpecPaxos::parent_status(p, l, CWordRef{w, 1}) & ST_INVARIANT) continue;
        out[1]++;
        for (int slot = 0; slot < SpecPaxos::max_slots(p); slot++) {
            const unsigned st = SpecPaxos::successor(p, w, slot, t);
            if (!(st & ST_ENABLED)) continue;
            out[2]++;
            if (SpecPaxos::check_invariants(p, t) & ST_INVARIANT) out[3]++;
        }
    }
    return 0;
}

extern "C" size_t shim_state_bytes(const mc_spec_desc *d) {
    size_t n = 0;
    dispatch_spec(d, [&](auto spec, const auto &prm) { n = sizeof(uint64_t) * decltype(spec)::words(prm); return 0; });
    return n;
}

// one (state, slot) pair of the device lowering on hand-made words (negative controls of invariants no reachable state violates)
extern "C" int shim_init_state(const mc_spec_desc *d, uint64_t k, uint64_t *words_out) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) { decltype(spec)::init(prm, k, WordRef{words_out, 1}); return 0; });
}
extern "C" int shim_eval_slot(const mc_spec_desc *d, const uint64_t *words, int slot, unsigned *status, uint64_t *fp) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        CWordRef s{words, 1};
        typename S::Local loc;
        S::load(prm, s, loc);
        *status = slot < S::nslots(prm, loc) ? S::eval(prm, loc, s, slot, *fp) : 0u;
        return 0;
    });
}

// ------------------------------------------------------------------------------------------
// Host emulation of the sharded step API (mc_shard_* of include/tlamc.h), same semantics, plain
// host pointers.  Used only by the world_size-2 gloo tests of tla_rust_amd/sharded.py.
extern "C" void mc_set_error_internal(const char *msg);
struct ShimShardBase {
    virtual ~ShimShardBase() {}
    virtual int begin() = 0;
    virtual int begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) = 0;
    virtual uint64_t level_size() = 0;
    virtual int expand_launch(unsigned slot, uint64_t first, uint64_t count) = 0;
    virtual int expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) = 0;
    virtual int materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int ingest(const uint8_t *recv_states, uint64_t n) = 0;
    virtual int keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) = 0;
    virtual uint64_t end_level() = 0;
    virtual void counters(uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) = 0;
    virtual void check_frontier() = 0;
    // counterexamples across ranks (always recorded here; the engine needs MC_F_TRACE)
    virtual int materialise_parents(unsigned slot, uint64_t *send_parents) = 0;
    virtual int ingest_parents(const uint64_t *recv_parents, uint64_t n, unsigned src_rank) = 0;
    virtual int violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) = 0;
    virtual int fetch(uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) = 0;
    virtual size_t state_bytes() = 0;
    // per-rank checkpoints (include/tlamc.h mc_shard_checkpoint / mc_shard_restore), in the host build's own format
    virtual int checkpoint(const char *path) = 0;
    virtual int restore(const char *path) = 0;
    std::vector<uint64_t> ck_levels;
    bool ck_ok = false, ck_resume = false;
};

template <class S>
struct ShimShard : ShimShardBase {
    typename S::Params prm;
    uint32_t rank, nranks;
    int W;
    std::vector<uint64_t> arena;
    FpSet seen;
    uint64_t lo = 0, hi = 0, generated = 0;
    int32_t verdict = MC_V_OK;
    struct Pending { uint64_t parent; int slot; };
    struct Slot {                              // two expand slots like the engine (tlamc.h: mc_shard_expand_launch)
        std::vector<Pending> pending;          // aligned with the compacted send_fp order
        std::vector<uint64_t> pend_off;        // per-owner offsets into pending
        std::vector<std::vector<uint64_t>> fps;
        std::vector<std::vector<Pending>> src;
        bool launched = false;
    } sl[2];

    uint64_t nstates() const { return arena.size() / (size_t)W; }
    // (rank, index, slot) of every state's parent, like the engine's d_prank / d_parent / d_pslot with MC_F_TRACE
    struct Par { uint32_t rank; uint64_t idx; uint32_t slot; };
    std::vector<Par> par;
    std::vector<uint64_t> moved_par[2];  // (parent index << 16 | slot) of the states the slot's last materialise sent, owner order
    bool v_found = false;
    uint64_t v_idx = 0;
    uint32_t v_slot = 0;
    void viol(int32_t kind, uint64_t idx, uint32_t slot) {
        if (verdict == MC_V_OK) verdict = kind;
        if (!v_found) { v_found = true; v_idx = idx; v_slot = slot; }
    }
    void push_state(const uint64_t *w, uint32_t prank, uint64_t pidx, uint32_t pslot) {
        arena.insert(arena.end(), w, w + W);
        par.push_back(Par{prank, pidx, pslot});
    }
    size_t state_bytes() override { return (size_t)W * 8; }
    int materialise_parents(unsigned slot, uint64_t *out) override {
        for (size_t k = 0; k < moved_par[slot & 1].size(); k++) out[k] = moved_par[slot & 1][k];
        return 0;
    }
    int ingest_parents(const uint64_t *pp, uint64_t n, unsigned src) override {
        for (uint64_t j = 0; j < n; j++) par[par.size() - n + j] = Par{src, pp[j] >> 16, (uint32_t)(pp[j] & 0xffffu)};
        return 0;
    }
    int violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *v, int32_t *inv) override {
        *found = v_found; *idx = v_idx; *slot = v_slot; *v = v_found ? verdict : MC_V_OK; *inv = v_found && verdict == MC_V_INVARIANT ? 0 : -1;
        return 0;
    }
    int fetch(uint64_t idx, uint8_t *out, uint32_t *prank, uint64_t *pidx, uint32_t *pslot) override {
        if (idx & (1ull << 63)) {  // (shard_loop.h FETCH_INIT) a violating initial state: rebuilt from its ordinal
            const uint64_t ord = idx & ~(1ull << 63);
            if (ord >= S::num_init(prm)) return MC_EBADCFG;
            S::init(prm, ord, WordRef{(uint64_t *)out, 1});
            *prank = rank; *pidx = 0xffffffffull; *pslot = 0xfffeu;
            return 0;
        }
        if (idx >= nstates()) return MC_EBADCFG;
        memcpy(out, &arena[idx * W], (size_t)W * 8);
        *prank = par[idx].rank; *pidx = par[idx].idx; *pslot = par[idx].slot;
        return 0;
    }
    int checkpoint(const char *path) override {
        if (!ck_ok || ck_levels.empty()) { mc_set_error_internal("shim checkpoint: needs a run that ended without an error"); return MC_EBADCFG; }
        FILE *f = fopen(path, "wb");
        if (!f) return MC_EBADCFG;
        const uint64_t hdr[10] = {0x314b434d494853ull, (uint64_t)W, rank, nranks, lo, hi, generated, dup, nstates(), ck_levels.size()};
        std::vector<uint64_t> fps;
        for (uint64_t v : seen.tab) if (v) fps.push_back(v);
        const uint64_t nf = fps.size();
        bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(ck_levels.data(), 8, ck_levels.size(), f) == ck_levels.size() &&
                  fwrite(arena.data(), 8, arena.size(), f) == arena.size() && fwrite(par.data(), sizeof(Par), par.size(), f) == par.size() &&
                  fwrite(&nf, 8, 1, f) == 1 && fwrite(fps.data(), 8, fps.size(), f) == fps.size();
        fclose(f);
        return ok ? 0 : MC_EBADCFG;
    }
    int restore(const char *path) override {
        FILE *f = fopen(path, "rb");
        if (!f) return MC_EPARSE;
        uint64_t hdr[10];
        W = S::words(prm);
        bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == 0x314b434d494853ull;
        if (ok && (hdr[1] != (uint64_t)W || hdr[2] != rank || hdr[3] != nranks)) { fclose(f); mc_set_error_internal("shim restore: another rank's file, or another spec / world size"); return MC_EBADCFG; }
        uint64_t nf = 0;
        std::vector<uint64_t> fps;
        if (ok) {
            ck_levels.assign(hdr[9], 0);
            arena.assign(hdr[8] * (uint64_t)W, 0);
            par.assign(hdr[8], Par{0, 0, 0});
            ok = fread(ck_levels.data(), 8, ck_levels.size(), f) == ck_levels.size() && fread(arena.data(), 8, arena.size(), f) == arena.size() &&
                 fread(par.data(), sizeof(Par), par.size(), f) == par.size() && fread(&nf, 8, 1, f) == 1;
            if (ok) { fps.assign(nf, 0); ok = fread(fps.data(), 8, nf, f) == nf; }
        }
        fclose(f);
        if (!ok) return MC_EPARSE;
        seen.clear();
        for (uint64_t v : fps) seen.insert(v);
        lo = hdr[4]; hi = hdr[5]; generated = hdr[6]; dup = hdr[7];
        verdict = MC_V_OK; v_found = false;
        sl[0].launched = sl[1].launched = false;
        ck_resume = true; ck_ok = false;
        return 0;
    }
    int begin() override {
        W = S::words(prm);
        dup = 0;
        ck_resume = ck_ok = false;
        arena.clear(); par.clear(); seen.clear(); generated = 0; verdict = MC_V_OK; v_found = false;
        uint64_t tmp[S::MAX_WORDS];
        for (uint64_t k = 0; k < S::num_init(prm); k++) {
            S::init(prm, k, WordRef{tmp, 1});
            const unsigned st = S::init_status(prm, CWordRef{tmp, 1});
            const uint64_t fp = (st & ST_OUT_OF_MODEL) ? 0 : S::fp_of(prm, CWordRef{tmp, 1});
            const bool mine = nranks <= 1 || (fp ? fp_owner(fp, nranks) == rank : rank == 0);
            if (!mine) continue;
            generated++;
            if (st & ST_INVARIANT) viol(MC_V_INVARIANT, k, 0xfffeu);  // (the ORDINAL of the initial state, as the engine reports it)
            if (fp && seen.insert(fp).second) push_state(tmp, rank, 0xffffffffull, 0xfffeu);
        }
        lo = 0; hi = nstates();
        return 0;
    }
    // mc_shard_begin_replicated: the same BFS on every rank until a level has >= min_frontier states, then a slice
    uint64_t dup = 0;
    int begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) override {
        W = S::words(prm);
        arena.clear(); par.clear(); seen.clear(); generated = 0; verdict = MC_V_OK; v_found = false;
        uint64_t tmp[S::MAX_WORDS];
        for (uint64_t k = 0; k < S::num_init(prm); k++) {
            S::init(prm, k, WordRef{tmp, 1});
            const unsigned st = S::init_status(prm, CWordRef{tmp, 1});
            generated++;
            if (st & ST_INVARIANT) viol(MC_V_INVARIANT, k, 0xfffeu);
            if (st & ST_OUT_OF_MODEL) continue;
            if (seen.insert(S::fp_of(prm, CWordRef{tmp, 1})).second) push_state(tmp, rank, 0xffffffffull, 0xfffeu);
        }
        uint64_t l = 0, h = nstates();
        uint32_t nl = 0;
        const uint32_t cap = *nlevels;
        auto push = [&](uint64_t n) { if (nl < cap) levels_out[nl] = n; nl++; };
        push(h);
        while (h > l && verdict == MC_V_OK && h - l < (min_frontier ? min_frontier : 1) && !(max_distinct && h >= max_distinct) &&
               !(max_levels && nl >= max_levels)) {
            for (uint64_t i = l; i < h; i++) {
                std::vector<uint64_t> cur(arena.begin() + (long)(i * W), arena.begin() + (long)((i + 1) * W));
                CWordRef s{cur.data(), 1};
                typename S::Local loc;
                S::load(prm, s, loc);
                const int ns = S::nslots(prm, loc);
                if (S::parent_status(prm, loc, s) & ST_INVARIANT) viol(MC_V_INVARIANT, i, 0xfffdu);
                uint64_t nsucc = 0;
                for (int slot = 0; slot < ns; slot++) {
                    uint64_t fp = 0;
                    const unsigned st = S::eval(prm, loc, s, slot, fp);
                    if (!(st & ST_ENABLED)) continue;
                    nsucc++; generated++;
                    if (st & ST_OVERFLOW) return MC_EOVERFLOW;
                    if (st & ST_ASSERT) { viol(MC_V_ASSERT, i, (uint32_t)slot); continue; }
                    if (st & ST_SPECERR) { viol(MC_V_SPECERR, i, (uint32_t)slot); continue; }
                    if (st & ST_INVARIANT) viol(MC_V_INVARIANT, i, (uint32_t)slot);
                    if (st & ST_OUT_OF_MODEL) continue;
                    if (seen.insert(fp).second) {
                        S::apply(prm, s, slot, WordRef{tmp, 1});
                        push_state(tmp, rank, i, (uint32_t)slot);
                    }
                }
                if (!nsucc && verdict == MC_V_OK) viol(MC_V_DEADLOCK, i, 0xffffu);
            }
            l = h;
            h = nstates();
            if (h > l) push(h - l);
        }
        if (nl > cap) return MC_EBADCFG;
        *nlevels = nl;
        const uint64_t base = h;
        if (verdict == MC_V_OK)
            for (uint64_t i = l; i < h; i++) {  // the states of the level whose fingerprint this rank owns
                std::vector<uint64_t> cur(arena.begin() + (long)(i * W), arena.begin() + (long)((i + 1) * W));
                if (nranks > 1 && fp_owner(S::fp_of(prm, CWordRef{cur.data(), 1}), nranks) != rank) continue;
                push_state(cur.data(), rank, i, 0xfffcu);  // SLOT_COPY: not a step
            }
        lo = base;
        hi = nstates();
        dup = rank == 0 ? hi - base : hi;
        if (rank != 0) generated = 0;
        return 0;
    }
    uint64_t level_size() override { return hi - lo; }
    int expand_launch(unsigned slot, uint64_t first, uint64_t count) override {
        Slot &q = sl[slot & 1];
        auto &fps = q.fps;
        auto &src = q.src;
        fps.assign(nranks, {});
        src.assign(nranks, {});
        q.launched = true;
        if (first + count > hi - lo) return MC_EBADCFG;
        for (uint64_t i = lo + first; i < lo + first + count; i++) {
            CWordRef s{&arena[i * W], 1};
            typename S::Local loc;
            S::load(prm, s, loc);
            const int ns = S::nslots(prm, loc);
            if (S::parent_status(prm, loc, s) & ST_INVARIANT) viol(MC_V_INVARIANT, i, 0xfffdu);
            uint64_t nsucc = 0;
            for (int slot = 0; slot < ns; slot++) {
                uint64_t fp = 0;
                const unsigned st = S::eval(prm, loc, s, slot, fp);
                if (!(st & ST_ENABLED)) continue;
                nsucc++; generated++;
                if (st & ST_OVERFLOW) return MC_EOVERFLOW;
                if (st & ST_ASSERT) { viol(MC_V_ASSERT, i, (uint32_t)slot); continue; }
                if (st & ST_SPECERR) { viol(MC_V_SPECERR, i, (uint32_t)slot); continue; }
                if (st & ST_INVARIANT) viol(MC_V_INVARIANT, i, (uint32_t)slot);
                if (st & ST_OUT_OF_MODEL) continue;
                const uint32_t o = fp_owner(fp, nranks);
                if (o == rank) {  // engine.hip, local-owner shortcut: probed at once, a new state joins this rank's own frontier
                    if (seen.insert(fp).second) {
                        uint64_t tmp[S::MAX_WORDS];
                        S::apply(prm, s, slot, WordRef{tmp, 1});
                        local_new.insert(local_new.end(), tmp, tmp + W);
                        local_par.push_back(Par{rank, i, (uint32_t)slot});
                    }
                    continue;
                }
                fps[o].push_back(fp);
                src[o].push_back({i, slot});
            }
            if (!nsucc && verdict == MC_V_OK) viol(MC_V_DEADLOCK, i, 0xffffu);
        }
        arena.insert(arena.end(), local_new.begin(), local_new.end());  // (after the loop: `s` points into the arena)
        par.insert(par.end(), local_par.begin(), local_par.end());
        local_new.clear();
        local_par.clear();
        return 0;
    }
    std::vector<uint64_t> local_new;
    std::vector<Par> local_par;
    int expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) override {
        Slot &q = sl[slot & 1];
        if (!q.launched) return MC_EBADCFG;
        q.launched = false;
        auto &fps = q.fps;
        auto &src = q.src;
        auto &pending = q.pending;
        auto &pend_off = q.pend_off;
        pending.clear();
        pend_off.assign(nranks + 1, 0);
        uint64_t k = 0;
        for (uint32_t o = 0; o < nranks; o++) {
            pend_off[o] = k;
            send_counts[o] = fps[o].size();
            if (k + fps[o].size() > send_cap) { mc_set_error_internal("shim expand: send buffer too small"); return MC_EROUTE; }
            for (size_t j = 0; j < fps[o].size(); j++) { send_fp[k++] = fps[o][j]; pending.push_back(src[o][j]); }
        }
        pend_off[nranks] = k;
        return 0;
    }
    int probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) override {
        for (uint64_t i = 0; i < n; i++) answers[i] = seen.insert(recv_fp[i]).second ? 1 : 0;
        return 0;
    }
    // exchange format: per owner a whole number of 64-state blocks, word-major inside a block
    int materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) override {
        auto &pending = sl[slot & 1].pending;
        auto &pend_off = sl[slot & 1].pend_off;
        uint64_t *out = (uint64_t *)send_states;
        uint64_t blk0 = 0;
        auto &mp = moved_par[slot & 1];
        mp.clear();
        for (uint32_t o = 0; o < nranks; o++) {
            uint64_t k = 0;
            for (uint64_t i = pend_off[o]; i < pend_off[o + 1]; i++) {
                if (!answers_back[i]) continue;
                if ((blk0 + k / 64 + 1) * 64 > send_cap) return MC_EARENA;  // (the loop sizes the buffer for the upper bound and repeats)
                mp.push_back((pending[i].parent << 16) | (uint64_t)(unsigned)pending[i].slot);
                S::apply(prm, CWordRef{&arena[pending[i].parent * W], 1}, pending[i].slot,
                         WordRef{out + (blk0 + k / 64) * (uint64_t)W * 64 + k % 64, 64});
                k++;
            }
            send_counts[o] = k;
            blk0 += (k + 63) / 64;
        }
        return 0;
    }
    int keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) override {
        auto &pending = sl[slot & 1].pending;
        std::vector<uint64_t> tmp(W);
        *n_new = 0;
        for (size_t i = 0; i < pending.size(); i++) {
            if (!answers_back[i]) continue;
            S::apply(prm, CWordRef{&arena[pending[i].parent * W], 1}, pending[i].slot, WordRef{tmp.data(), 1});
            push_state(tmp.data(), rank, pending[i].parent, (uint32_t)pending[i].slot);
            (*n_new)++;
        }
        return 0;
    }
    int ingest(const uint8_t *recv_states, uint64_t n) override {  // one source's bucket
        const uint64_t *in = (const uint64_t *)recv_states;
        for (uint64_t j = 0; j < n; j++) {
            for (int w = 0; w < W; w++) arena.push_back(in[(j / 64) * (uint64_t)W * 64 + (uint64_t)w * 64 + j % 64]);
            par.push_back(Par{rank, 0xfffffffeull, 0});  // produced on another rank: ingest_parents fills it in
        }
        return 0;
    }
    uint64_t end_level() override { lo = hi; hi = nstates(); return hi - lo; }
    void counters(uint64_t *g, uint64_t *d, int32_t *v) override { *g = generated; *d = nstates() - dup; *v = verdict; }
    void check_frontier() override {  // mc_shard_check_frontier: check-on-expand invariants of the unexpanded local frontier
        for (uint64_t i = lo; i < hi && verdict == MC_V_OK; i++) {
            CWordRef s{&arena[i * W], 1};
            typename S::Local loc;
            S::load(prm, s, loc);
            if (S::parent_status(prm, loc, s) & ST_INVARIANT) viol(MC_V_INVARIANT, i, 0xfffdu);
        }
    }
};

extern "C" {
void *shim_shard_create(const mc_spec_desc *d, uint32_t rank, uint32_t nranks) {
    ShimShardBase *e = nullptr;
    dispatch_spec(d, [&](auto spec, const auto &prm) {
        auto *x = new ShimShard<decltype(spec)>();
        x->prm = prm; x->rank = rank; x->nranks = nranks ? nranks : 1; x->W = decltype(spec)::words(prm);
        e = x;
        return 0;
    });
    return e;
}
void shim_shard_destroy(void *e) { delete (ShimShardBase *)e; }
int shim_shard_begin(void *e) { return ((ShimShardBase *)e)->begin(); }
int shim_shard_checkpoint(void *e, const char *path) { return ((ShimShardBase *)e)->checkpoint(path); }
int shim_shard_restore(void *e, const char *path) { return ((ShimShardBase *)e)->restore(path); }
int shim_shard_begin_replicated(void *e, uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out,
                                uint32_t *nlevels) {
    return ((ShimShardBase *)e)->begin_replicated(min_frontier, max_distinct, max_levels, levels_out, nlevels);
}
int shim_shard_level_size(void *e, uint64_t *n) { *n = ((ShimShardBase *)e)->level_size(); return 0; }
int shim_shard_expand_launch(void *e, uint32_t slot, uint64_t first, uint64_t count) {
    return ((ShimShardBase *)e)->expand_launch(slot, first, count);
}
int shim_shard_expand_finish(void *e, uint32_t slot, uint64_t *send_fp, uint64_t cap, uint64_t *counts) {
    return ((ShimShardBase *)e)->expand_finish(slot, send_fp, cap, counts);
}
int shim_shard_probe(void *e, const uint64_t *fp, uint64_t n, uint8_t *ans) { return ((ShimShardBase *)e)->probe(fp, n, ans); }
int shim_shard_materialise(void *e, uint32_t slot, const uint8_t *ans, uint8_t *states, uint64_t cap, uint64_t *counts) {
    return ((ShimShardBase *)e)->materialise(slot, ans, states, cap, counts);
}
int shim_shard_ingest(void *e, const uint8_t *states, uint64_t n) { return ((ShimShardBase *)e)->ingest(states, n); }
int shim_shard_keep(void *e, uint32_t slot, const uint8_t *ans, uint64_t *n) { return ((ShimShardBase *)e)->keep(slot, ans, n); }
int shim_shard_end_level(void *e, uint64_t *n) { *n = ((ShimShardBase *)e)->end_level(); return 0; }
int shim_shard_counters(void *e, uint64_t *g, uint64_t *d, int32_t *v) { ((ShimShardBase *)e)->counters(g, d, v); return 0; }
int shim_shard_check_frontier(void *e) { ((ShimShardBase *)e)->check_frontier(); return 0; }
}

// ------------------------------------------------------------------------------------------
// The product's level loop (tla_rust_amd/csrc/shard_loop.h — the same source libtlamc.so compiles) over the host emulation
// above: what the multi-rank CPU tests run, with a transport the test supplies (torch.distributed gloo through callbacks).
#include "../../tla_rust_amd/csrc/shard_loop.h"

static thread_local std::string g_shim_error;
extern "C" void mc_set_error_internal(const char *msg) { g_shim_error = msg ? msg : ""; }
extern "C" const char *shim_last_error() { return g_shim_error.c_str(); }
extern "C" const char *mc_strerror(int code) {
    switch (code) {
        case MC_OK: return "ok";
        case MC_EBADCFG: return "bad configuration";
        case MC_EHIP: return "HIP error";
        case MC_EOVERFLOW: return "packed-state slot overflow";
        case MC_ETABLEFULL: return "seen-set full";
        case MC_EARENA: return "arena exhausted";
        case MC_EROUTE: return "exchange bucket full";
        case MC_ERCCL: return "exchange failure";
        case MC_ESTATE: return "call sequence error";
        default: return "error";
    }
}

struct ShimOps {
    ShimShardBase *s;
    uint32_t P;
    std::vector<uint64_t> pack_counts[2];
    uint64_t route_max = 0, route_lvl = 0, route_sum = 0;  // what AbiOps reads back from the in-band counts of the HIP engine
    uint64_t chunk_limit() const { return 0; }
    size_t state_bytes() const { return s->state_bytes(); }
    bool traced() const { return true; }
    void record(int, int) {}
    void wait(int, int) {}
    void clear_counts(uint64_t *buf, uint32_t n, uint64_t cap) { for (uint32_t t = 0; t < n; t++) buf[(uint64_t)t * cap] = 0; }
    void clear_bytes(void *p, size_t n, int) { memset(p, 0, n); }
    int resume(uint64_t *lv, uint32_t *n) {
        const uint32_t cap = *n;
        *n = 0;
        if (!s->ck_resume) return 0;
        if (s->ck_levels.size() > cap) return MC_EBADCFG;
        for (size_t k = 0; k < s->ck_levels.size(); k++) lv[k] = s->ck_levels[k];
        *n = (uint32_t)s->ck_levels.size();
        s->ck_resume = false;
        return 0;
    }
    int note_levels(const uint64_t *lv, uint32_t n, int32_t verdict) { s->ck_levels.assign(lv, lv + n); s->ck_ok = verdict == MC_V_OK || verdict == MC_V_BUDGET; return 0; }
    int begin() { return s->begin(); }
    int begin_replicated(uint64_t mf, uint64_t md, uint64_t ml, uint64_t *lv, uint32_t *n) { s->ck_resume = s->ck_ok = false; return s->begin_replicated(mf, md, ml, lv, n); }
    int level_size(uint64_t *n) { *n = s->level_size(); return 0; }
    // like the engine, a launch into a slot invalidates what the slot's previous round left pending: a loop that launches round
    // r+1 before it has issued the keep of round r-1 (same slot) fails here as it does on the GPU
    int expand_launch(uint32_t slot, uint64_t first, uint64_t count, uint64_t) { pack_counts[slot & 1].clear(); return s->expand_launch(slot, first, count); }
    int expand_finish(uint32_t slot, uint64_t *fp, uint64_t cap, uint64_t *counts) { return s->expand_finish(slot, fp, cap, counts); }
    // fixed-capacity rounds emulated on the variable-size step calls: the in-band layout of include/tlamc.h mc_shard_*_pack
    int expand_pack(uint32_t slot, uint64_t *send_fp, uint64_t cap) {
        std::vector<uint64_t> tmp((size_t)P * cap), counts(P);
        int rc = s->expand_finish(slot, tmp.data(), tmp.size(), counts.data());
        if (rc) return rc;
        memset(send_fp, 0, (size_t)P * cap * 8);
        uint64_t off = 0;
        for (uint32_t t = 0; t < P; t++) {
            route_max = std::max(route_max, counts[t]);
            route_sum += counts[t];
            if (getenv("TLAMC_SHARD_DEBUG_ROUNDS") && counts[t]) fprintf(stderr, "[shard]   pack slot %u owner %u: %llu entries, cap %llu\n", slot, t, (unsigned long long)counts[t], (unsigned long long)cap);
        }
        for (uint32_t t = 0; t < P; t++) {
            if (counts[t] + 1 > cap) { mc_set_error_internal("an exchange bucket is full"); return MC_EROUTE; }
            send_fp[(uint64_t)t * cap] = counts[t];
            memcpy(send_fp + (uint64_t)t * cap + 1, tmp.data() + off, counts[t] * 8);
            off += counts[t];
        }
        pack_counts[slot & 1] = counts;
        return 0;
    }
    int probe(const uint64_t *fp, uint64_t n, uint8_t *ans) { return s->probe(fp, n, ans); }
    int probe_pack(const uint64_t *recv, uint64_t cap, uint8_t *ans) {
        memset(ans, 0, (size_t)P * cap);
        // like k_probe_packed: the sources are walked INTERLEAVED, 256 entries at a time (walked one after the other, the lower
        // ranks would win every same-round tie and keep the state)
        uint64_t most = 0;
        for (uint32_t q = 0; q < P; q++) most = std::max(most, recv[(uint64_t)q * cap] < cap ? recv[(uint64_t)q * cap] : 0);
        for (uint64_t b = 0; b < most; b += 256)
            for (uint32_t q = 0; q < P; q++) {
                const uint64_t n = recv[(uint64_t)q * cap] < cap ? recv[(uint64_t)q * cap] : 0;
                if (b < n) s->probe(recv + (uint64_t)q * cap + 1 + b, std::min<uint64_t>(256, n - b), ans + (uint64_t)q * cap + 1 + b);
            }
        return 0;
    }
    int keep_pack(uint32_t slot, const uint8_t *back, uint64_t cap) {
        const auto &counts = pack_counts[slot & 1];
        if (counts.size() != P) return MC_ESTATE;
        std::vector<uint8_t> flat;
        for (uint32_t t = 0; t < P; t++) {
            // answers outside the counts must be 0 (the HIP sender scans the whole packed range)
            if (back[(uint64_t)t * cap]) return MC_ESTATE;
            for (uint64_t j = 1 + counts[t]; j < cap; j++) if (back[(uint64_t)t * cap + j]) return MC_ESTATE;
            flat.insert(flat.end(), back + (uint64_t)t * cap + 1, back + (uint64_t)t * cap + 1 + counts[t]);
        }
        uint64_t n = 0;
        if (flat.empty()) flat.push_back(0);
        return s->keep(slot, flat.data(), &n);
    }
    int wait_keep(uint32_t) { return 0; }
    int keep(uint32_t slot, const uint8_t *back) { uint64_t n = 0; return s->keep(slot, back, &n); }
    int materialise_slot(uint32_t slot, const uint8_t *back, uint8_t *st, uint64_t cap, uint64_t *counts) { return s->materialise(slot, back, st, cap, counts); }
    int materialise_parents(uint32_t slot, uint64_t *out) { return s->materialise_parents(slot, out); }
    int ingest(const uint8_t *st, uint64_t n) { return s->ingest(st, n); }
    int ingest_parents(const uint64_t *pp, uint64_t n, uint32_t src) { return s->ingest_parents(pp, n, src); }
    int end_level(uint64_t *n) { *n = s->end_level(); route_lvl = route_max; route_max = 0; return 0; }
    int route_fill(uint64_t *mx, uint64_t *sum) { *mx = route_lvl; *sum = route_sum; return 0; }
    int counters(uint64_t *g, uint64_t *d, int32_t *v) { s->counters(g, d, v); return 0; }
    int check_frontier() { s->check_frontier(); return 0; }
    int violation(int32_t *f, uint64_t *i, uint32_t *sl, int32_t *v, int32_t *inv) { return s->violation(f, i, sl, v, inv); }
    int fetch(uint64_t idx, uint8_t *st, uint32_t *pr, uint64_t *pi, uint32_t *ps) { return s->fetch(idx, st, pr, pi, ps); }
};

extern "C" int shim_shard_run_transport(void *e, const mc_transport *t, const mc_shard_opts *o, mc_result *out) {
    ShimOps ops{(ShimShardBase *)e, t->world, {}};
    return mc_shard::run_restarting(ops, *t, *o, out);
}
extern "C" int shim_shard_trace_transport(void *e, const mc_transport *t, uint8_t *states_out, int32_t *slots_out, size_t *n_inout, int32_t *final_slot) {
    ShimOps ops{(ShimShardBase *)e, t->world, {}};
    mc_shard::Loop<ShimOps> loop(ops, *t);
    return loop.trace(states_out, slots_out, n_inout, final_slot);
}
// host evaluation of one (state, slot) pair for the tests' counterexample printing: the successor and the action id
extern "C" int shim_state_apply(const mc_spec_desc *d, const uint64_t *words, int slot, uint64_t *out) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        S::apply(prm, CWordRef{words, 1}, slot, WordRef{out, 1});
        return 0;
    });
}
// ANALYSIS AID (profiles/probe_cache_sim.py): what fraction of the seen-set probes would a direct-mapped cache of recently probed
// fingerprints answer?  BFS in arena order; per 64-parent block the wavefront's own 256-entry filter (engine.hip WFILT) first,
// then a global direct-mapped cache of 2^cache_log2 fingerprints, then the exact set.  out: candidates, wave-filter hits, cache
// hits, duplicates that reached the table, new states.
extern "C" int shim_probe_cache_sim(const mc_spec_desc *d, uint64_t max_levels, int cache_log2, uint64_t *out) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        const int W = S::words(prm);
        std::vector<uint64_t> cur, next, cache((size_t)1 << cache_log2, 0);
        const uint64_t cmask = ((uint64_t)1 << cache_log2) - 1;
        FpSet seen;
        uint64_t tmp[S::MAX_WORDS], filt[256];
        for (int q = 0; q < 5; q++) out[q] = 0;
        for (uint64_t k = 0; k < S::num_init(prm); k++) {
            S::init(prm, k, WordRef{tmp, 1});
            if (S::init_status(prm, CWordRef{tmp, 1}) & ST_OUT_OF_MODEL) continue;
            if (seen.insert(stored_fp<S>(prm, tmp)).second) next.insert(next.end(), tmp, tmp + W);
        }
        cur.swap(next);
        for (uint64_t level = 1; !cur.empty() && (!max_levels || level < max_levels); level++) {
            const uint64_t nstates = cur.size() / (size_t)W;
            for (uint64_t i = 0; i < nstates; i++) {
                if ((i & 63) == 0) memset(filt, 0, sizeof filt);
                CWordRef s{&cur[i * W], 1};
                typename S::Local loc;
                S::load(prm, s, loc);
                const int ns = S::nslots(prm, loc);
                for (int slot = 0; slot < ns; slot++) {
                    uint64_t fp = 0;
                    const unsigned st = S::eval(prm, loc, s, slot, fp);
                    if (!(st & ST_ENABLED) || (st & (ST_OVERFLOW | ST_ASSERT | ST_SPECERR | ST_OUT_OF_MODEL | ST_SELFLOOP))) continue;
                    out[0]++;
                    const unsigned h = (unsigned)(fp >> 20) & 255u;
                    if (filt[h] == fp) { out[1]++; continue; }
                    filt[h] = fp;
                    const uint64_t ch = (fp >> 24) & cmask;
                    if (cache[ch] == fp) { out[2]++; continue; }
                    cache[ch] = fp;
                    if (seen.insert(fp).second) {
                        out[4]++;
                        S::apply(prm, s, slot, WordRef{tmp, 1});
                        next.insert(next.end(), tmp, tmp + W);
                    } else {
                        out[3]++;
                    }
                }
            }
            cur.clear();
            cur.swap(next);
        }
        return 0;
    });
}
// ANALYSIS AID (profiles/probe_order_sim.py, round 6): how many seen-set look-ups does the wavefront's own duplicate filter (wfilt entries,
// direct-mapped, cleared per 64 parents) answer, as a function of the ORDER in which a level's states lie in the arena?  mode 0: parent-major
// (a parent's new states next to each other, in slot order: what a host BFS appends); mode 1: the device's — per workgroup of 128 parents
// the survivors sorted by action class, inside a class by wavefront, then slot-major (the order in which the probe batches confirmed them);
// mode 2: class-major like 1, but the survivors of one PARENT PAIR-GROUP of `group` parents kept together (group-major, then class).
// out: candidates, filter hits, duplicates that reached the table, new states.
template <class S, class = void>
struct ShimSlotClass { static int of(int) { return 0; } };
template <class S>
struct ShimSlotClass<S, decltype((void)S::NCLS)> { static int of(int slot) { return S::slot_class(slot); } };
extern "C" int shim_probe_order_sim(const mc_spec_desc *d, uint64_t max_levels, int mode, int wfilt, int group, uint64_t *out) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        const int W = S::words(prm);
        std::vector<uint64_t> cur, next;
        FpSet seen;
        uint64_t tmp[S::MAX_WORDS];
        std::vector<uint64_t> filt((size_t)wfilt);
        for (int q = 0; q < 4; q++) out[q] = 0;
        for (uint64_t k = 0; k < S::num_init(prm); k++) {
            S::init(prm, k, WordRef{tmp, 1});
            if (S::init_status(prm, CWordRef{tmp, 1}) & ST_OUT_OF_MODEL) continue;
            if (seen.insert(stored_fp<S>(prm, tmp)).second) next.insert(next.end(), tmp, tmp + W);
        }
        cur.swap(next);
        struct Surv { int cls, wave, slot, lane; std::vector<uint64_t> row; };
        for (uint64_t level = 1; !cur.empty() && (!max_levels || level < max_levels); level++) {
            const uint64_t nstates = cur.size() / (size_t)W;
            std::vector<Surv> wg;
            auto flush_wg = [&]() {
                if (mode == 1) std::stable_sort(wg.begin(), wg.end(), [](const Surv &a, const Surv &b) {
                    return std::make_tuple(a.cls, a.wave, a.slot, a.lane) < std::make_tuple(b.cls, b.wave, b.slot, b.lane); });
                else if (mode == 2) std::stable_sort(wg.begin(), wg.end(), [&](const Surv &a, const Surv &b) {
                    return std::make_tuple((a.wave * 64 + a.lane) / group, a.cls, a.slot, a.lane) < std::make_tuple((b.wave * 64 + b.lane) / group, b.cls, b.slot, b.lane); });
                for (auto &x : wg) next.insert(next.end(), x.row.begin(), x.row.end());
                wg.clear();
            };
            for (uint64_t i = 0; i < nstates; i++) {
                if ((i & 63) == 0) std::fill(filt.begin(), filt.end(), 0ull);
                if ((i & 127) == 0) flush_wg();
                CWordRef s{&cur[i * W], 1};
                typename S::Local loc;
                S::load(prm, s, loc);
                const int ns = S::nslots(prm, loc);
                for (int slot = 0; slot < ns; slot++) {
                    uint64_t fp = 0;
                    const unsigned st = S::eval(prm, loc, s, slot, fp);
                    if (!(st & ST_ENABLED) || (st & (ST_OVERFLOW | ST_ASSERT | ST_SPECERR | ST_OUT_OF_MODEL | ST_SELFLOOP))) continue;
                    out[0]++;
                    const unsigned h = (unsigned)(fp >> 20) & (unsigned)(wfilt - 1);
                    if (filt[h] == fp) { out[1]++; continue; }
                    filt[h] = fp;
                    if (seen.insert(fp).second) {
                        out[3]++;
                        S::apply(prm, s, slot, WordRef{tmp, 1});
                        wg.push_back(Surv{ShimSlotClass<S>::of(slot), (int)((i >> 6) & 1), slot, (int)(i & 63), std::vector<uint64_t>(tmp, tmp + W)});
                    } else {
                        out[2]++;
                    }
                }
            }
            flush_wg();
            cur.clear();
            cur.swap(next);
        }
        return 0;
    });
}
// the consistency checks shim_run applies to every reachable state (by-family evaluation == slot-by-slot evaluation, dense pairs,
// load_expand == load + guards), on ONE hand-made state: states no small model reaches (a Leader s5 of the 5-server model)
extern "C" long shim_state_mismatches(const mc_spec_desc *d, const uint64_t *words) {
    long bad = -1;
    dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        CWordRef s{words, 1};
        typename S::Local loc;
        S::load(prm, s, loc);
        const int ns = S::nslots(prm, loc);
        bad = (long)(FamCheck<S>::mismatches(prm, loc, s, ns) + DenseCheck<S>::mismatches(prm, loc, s));
        return 0;
    });
    return bad;
}
extern "C" int shim_state_format(const mc_spec_desc *d, const uint64_t *words, char *buf, size_t cap) {
    return dispatch_spec(d, [&](auto spec, const auto &prm) { return decltype(spec)::format(prm, words, buf, cap); });
}
extern "C" const char *shim_state_action_name(const mc_spec_desc *d, const uint64_t *words, int slot) {
    const char *name = "?";
    dispatch_spec(d, [&](auto spec, const auto &prm) {
        using S = decltype(spec);
        name = S::action_name(S::action_of(prm, words, slot));
        return 0;
    });
    return name;
}
