// spec_ssi.h — device lowering of examples/serializableSnapshotIsolation.tla (Cahill's serializable
// snapshot isolation; reference lines cited per action) under specs/MCssi.tla / MCssi.cfg:
// TxnId = {T1..Tn} (n <= 4), Key = {K1..Km} (m <= 3), NoLock a model value, INIT Init / NEXT Next.
//
// Packed state = 10 words (80 B):
//   word 0      additive fingerprint (raw sum of H(word w, salt_w), w = 1..9)
//   word 1      meta: Len(history)[0,6) | per txn t at bit 8+10t: holdingXLocks 3 | waitingForXLock 2
//               (key, 3 = NoLock) | inConflict 1 | outConflict 1 | holdingSIREADlocks 3
//   words 2..9  history, 4 events per word, 16 bits each: op 3 | txn 2 | key 2 | ver 2 | reason 3
//               (history is append-only, at most |TxnId| * (2|Key| + 2) = 32 events)
//
// The spec's helper operators all scan `history`; here ONE unrolled pass per parent (load())
// builds small index tables (begin / commit position per transaction, write / read position per
// (key, transaction), read versions, key masks) and every action and invariant is a table lookup.
// All 77 action slots have compile-time (transaction, key, choice) indices and are fully unrolled
// by the expand kernel.  Every CHOOSE over transactions is resolved in ascending order T1 < T2 < ..
// (Commit's AbortOpSeq :465-474); the other CHOOSEs are over singletons.
//
// SYMMETRY (the run-book makes Key and TxnId symmetry sets, :38-44).  TLC canonicalises a successor by trying
// every permutation; here the orbit representative is chosen so that NO search is needed: transactions are
// numbered in the order of their `begin` events and keys in the order of their first read / write event.  Every
// other label of a state is at its initial value (an unstarted transaction, an untouched key), so two states
// are in one orbit iff these relabelled forms are equal.  A parent stored in that form has started = {T1..Tj}
// and touched keys {K1..Ki}; an action introduces at most one new label, so its successor's representative is
// the same action taken with the lowest unused label: Begin(t), t unstarted, appends `begin T(j+1)`;
// StartWriteMayBlock(txn, k), k untouched, writes K(i+1).  The aliased slots still count as generated
// successors (TLC generates each and finds the duplicates through the fingerprints), the seen-set does the
// rest, and the fingerprint stays incremental.  (tests/ compare this, level by level, with a brute-force search
// over all |TxnId|! * |Key|! permutations.)
//
// Invariants (:59-79) are evaluated once per state when it is EXPANDED (parent_status), not per
// generated successor: every stored state is expanded exactly once, so the verdict and the length
// of the shortest counterexample are the same as TLC's check-on-generation.  A run that stops on a
// budget leaves its last level unexpanded: the engine evaluates parent_status on that frontier before
// it reports (CHECK_ON_EXPAND, k_check_frontier), so no counted state is ever left unchecked.
#pragma once
#include "mc_common.h"
#include <stdio.h>
#include <string.h>

// MC_SSI_COMMIT_FAMILY (default on): Commit(txn) is a family of its own, LAST in the wavefront's pair list, so the states that end in a
// `commit` lie together in the arena — and only the wavefronts of the next level that hold them evaluate the three invariants about
// committed transactions (parent_status_step); 0 = A/B: Commit rides with Begin / ChooseToAbort
#ifndef MC_SSI_COMMIT_FAMILY
#define MC_SSI_COMMIT_FAMILY 1
#endif
namespace mc {

struct SsiParams { int nt, nk, inv_mask, find, textbook, sym; };  // textbook = 1: examples/textbookSnapshotIsolation.tla; sym: cfg SYMMETRY

struct SpecSsi {
    using Params = SsiParams;
    static constexpr int NT = 4, NK = 3, HWORDS = 8, HCAP = 32;
#ifndef MC_SSI_STEP_STATUS
#define MC_SSI_STEP_STATUS 1   // (0 = A/B: the kernels evaluate every invariant of every stored state, as in rounds 1-5)
#endif
#if MC_SSI_STEP_STATUS
    static constexpr bool STEP_STATUS = true;      // the kernels evaluate parent_status_step (below)
#endif
    static constexpr bool CHECK_ON_EXPAND = true;  // invariants live in parent_status: the engine checks a run's last, unexpanded level too
    static constexpr int W_FP = 0, W_META = 1, W_H0 = 2;
    static constexpr int MAX_WORDS = 10;
    MC_HD static int words(const Params &) { return MAX_WORDS; }
    // slots per txn t (base 19 t): 0 Begin, 1 Commit, 2 ChooseToAbort, 3 FinishBlockedWrite, 4+k Read(k),
    // 7+4k+c StartWriteMayBlock(k) with c = index of the deadlock victim (c = 0 otherwise); slot 76 = termination
    static constexpr int PER_TXN = 4 + NK + NK * NT, TOTAL_SLOTS = NT * PER_TXN + 1;
    static constexpr int FIX_SLOTS = TOTAL_SLOTS;
    static constexpr int STAGE_WORDS = 0;
    MC_HD static int max_slots(const Params &) { return TOTAL_SLOTS; }
    enum : int { OP_BEGIN = 0, OP_READ = 1, OP_WRITE = 2, OP_COMMIT = 3, OP_ABORT = 4 };
    enum : int { R_VOLUNTARY = 0, R_FCW = 1, R_DEADLOCK = 2, R_COMMIT = 3, R_READ = 4, R_WRITE = 5 };
    enum : int { SA_BEGIN, SA_COMMIT, SA_ABORT, SA_READ, SA_WRITE, SA_FINISH, SA_TERMINATED };
    static constexpr unsigned NOLOCK = 3;

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        if (np < 2) return -1;
        o.nt = (int)p[0]; o.nk = (int)p[1];
        o.inv_mask = np > 2 ? (int)p[2] : 127;
        o.find = np > 3 ? (int)p[3] : 0;
        o.textbook = np > 4 ? (int)p[4] : 0;  // the same model without Cahill's variables (textbookSnapshotIsolation.tla:115)
        o.sym = np > 5 ? (int)p[5] & 3 : 0;   // cfg SYMMETRY: bit 0 Permutations(TxnId), bit 1 Permutations(Key) (:38-44)
        if (o.nt < 1 || o.nt > NT || o.nk < 1 || o.nk > NK || o.find < 0 || o.find > 7) return -1;
        return 0;
    }

    // ------------------------------------------------------------------ meta word accessors
    MC_HD static int m_len(uint64_t m) { return (int)(m & 63); }
    MC_HD static unsigned m_txn(uint64_t m, int t) { return (unsigned)(m >> (8 + 10 * t)) & 1023u; }
    MC_HD static unsigned t_xl(unsigned f) { return f & 7u; }
    MC_HD static unsigned t_wait(unsigned f) { return f >> 3 & 3u; }
    MC_HD static unsigned t_in(unsigned f) { return f >> 5 & 1u; }
    MC_HD static unsigned t_out(unsigned f) { return f >> 6 & 1u; }
    MC_HD static unsigned t_sir(unsigned f) { return f >> 7 & 7u; }
    MC_HD static unsigned mk_txn(unsigned xl, unsigned wait, unsigned in, unsigned out, unsigned sir) {
        return xl | wait << 3 | in << 5 | out << 6 | sir << 7;
    }
    MC_HD static uint64_t m_set_txn(uint64_t m, int t, unsigned f) { return bits_set(m, 8 + 10 * t, 10, f); }
    MC_HD static unsigned mk_event(int op, int txn, int key, int ver, int reason) {
        return (unsigned)op | (unsigned)txn << 3 | (unsigned)key << 5 | (unsigned)ver << 7 | (unsigned)reason << 9;
    }
    MC_HD static unsigned idx6(uint32_t tab, int t) { return tab >> (6 * t) & 63u; }

    // H: one word's contribution to the additive fingerprint.  Round 6: hmum (two 32 x 32 -> 64 multiply-accumulates) instead of fmix64
    // (two 64-bit multiplies = a dozen quarter-rate instructions): the kernels of this spec are bound by VALU issue, and a successor's
    // fingerprint is three to six of these
    MC_HD static uint64_t H(uint64_t x, uint64_t salt) { return hmum(x, salt); }
    // ------------------------------------------------------------------ Init :938-943
    MC_HD static uint64_t num_init(const Params &) { return 1; }
    MC_HD static uint64_t init_meta() {
        uint64_t m = 0;
        for (int t = 0; t < NT; t++) m = m_set_txn(m, t, mk_txn(0, NOLOCK, 0, 0, 0));
        return m;
    }
    MC_HD static void init(const Params &, uint64_t, WordRef out) {
        for (int w = 0; w < MAX_WORDS; w++) out.set(w, 0);
        out.set(W_META, init_meta());
        uint64_t fp = 0;
        for (int w = 1; w < MAX_WORDS; w++) fp += H(out.get(w), salt_of((unsigned)w));
        out.set(W_FP, fp);
    }
    template <class Ref>
    MC_HD static uint64_t fp_of(const Params &, Ref s) { return fp_nonzero(s.get(W_FP)); }
    template <class Ref>
    MC_HD static uint64_t fp_recompute(const Params &, Ref s) {
        uint64_t fp = 0;
        for (int w = 1; w < MAX_WORDS; w++) fp += H(s.get(w), salt_of((unsigned)w));
        return fp;
    }
    template <class Ref>
    MC_HD static unsigned init_status(const Params &, Ref) { return ST_ENABLED; }

    // ------------------------------------------------------------------ per-parent tables
    struct Local {
        uint64_t fp, meta;
        int n;
        unsigned started, committed, aborted;  // transaction masks
        uint32_t bidx, cidx;                   // 6 bits per txn: position of begin / commit (0 = none)
        uint32_t rkeys, wkeys;                 // 4 bits per txn: keys read / written
        uint32_t widx0, widx1, widx2;          // per key: 6 bits per txn, position of its write
        uint32_t ridx0, ridx1, ridx2;          // per key: position of its read
        uint32_t rver0, rver1, rver2;          // per key: 2 bits per txn, version read
        uint32_t abort_reasons;                // bit r: some transaction aborted with reason r
        bool wellformed;
        // the history's TAIL (parent_status_step): its last event, and whether it ends in `commit` followed by nothing but the aborts of
        // that commit's losers (reason "First Committer Wins")
        int tail_op, tail_t, tail_k;
        bool tail_commit;
    };
    MC_HD static uint32_t pick3(uint32_t a, uint32_t b, uint32_t c, int k) { return k == 0 ? a : k == 1 ? b : c; }
    MC_HD static uint32_t widx(const Local &l, int k) { return pick3(l.widx0, l.widx1, l.widx2, k); }
    MC_HD static uint32_t ridx(const Local &l, int k) { return pick3(l.ridx0, l.ridx1, l.ridx2, k); }
    MC_HD static uint32_t rver(const Local &l, int k) { return pick3(l.rver0, l.rver1, l.rver2, k); }

    template <class Ref>
    MC_HD static void load(const Params &, Ref s, Local &l) {
        l.fp = s.get(W_FP);
        l.meta = s.get(W_META);
        l.n = m_len(l.meta);
        l.started = l.committed = l.aborted = 0;
        l.bidx = l.cidx = l.rkeys = l.wkeys = 0;
        l.widx0 = l.widx1 = l.widx2 = l.ridx0 = l.ridx1 = l.ridx2 = l.rver0 = l.rver1 = l.rver2 = 0;
        l.abort_reasons = 0;
        l.wellformed = true;
        l.tail_op = -1; l.tail_t = 0; l.tail_k = 0; l.tail_commit = false;
        // one pass over the history; WellFormedTransactionsInHistory (:1146-1179) is checked on the way
#pragma unroll
        for (int w = 0; w < HWORDS; w++) {
            if (4 * w >= l.n) continue;
            const uint64_t word = s.get(W_H0 + w);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int i = 4 * w + q;
                if (i >= l.n) continue;
                const unsigned e = (unsigned)(word >> (16 * q)) & 0xffffu;
                const int op = (int)(e & 7u), t = (int)(e >> 3 & 3u), k = (int)(e >> 5 & 3u);
                const unsigned tb = 1u << t, pos = (unsigned)(i + 1);
                l.tail_op = op; l.tail_t = t; l.tail_k = k;
                l.tail_commit = op == OP_COMMIT || (l.tail_commit && op == OP_ABORT && (e >> 9 & 7u) == (unsigned)R_FCW);
                const bool fin = ((l.committed | l.aborted) & tb) != 0, st = (l.started & tb) != 0;
                if (fin) l.wellformed = false;  // nothing may follow commit / abort
                if (op == OP_BEGIN) {
                    if (st) l.wellformed = false;
                    l.started |= tb;
                    l.bidx |= pos << (6 * t);
                } else {
                    if (!st) l.wellformed = false;  // first operation must be begin
                    if (op == OP_COMMIT) { l.committed |= tb; l.cidx |= pos << (6 * t); }
                    else if (op == OP_ABORT) { l.aborted |= tb; l.abort_reasons |= 1u << (e >> 9 & 7u); }
                    else if (op == OP_READ) {
                        if (l.rkeys >> (4 * t + k) & 1u) l.wellformed = false;  // Bernstein's simplification
                        l.rkeys |= 1u << (4 * t + k);
                        const uint32_t a = pos << (6 * t), v = (e >> 7 & 3u) << (2 * t);
                        l.ridx0 |= k == 0 ? a : 0; l.ridx1 |= k == 1 ? a : 0; l.ridx2 |= k == 2 ? a : 0;
                        l.rver0 |= k == 0 ? v : 0; l.rver1 |= k == 1 ? v : 0; l.rver2 |= k == 2 ? v : 0;
                    } else {
                        if (l.wkeys >> (4 * t + k) & 1u) l.wellformed = false;
                        l.wkeys |= 1u << (4 * t + k);
                        const uint32_t a = pos << (6 * t);
                        l.widx0 |= k == 0 ? a : 0; l.widx1 |= k == 1 ? a : 0; l.widx2 |= k == 2 ? a : 0;
                    }
                }
            }
        }
    }
    MC_HD static int nslots(const Params &, const Local &) { return TOTAL_SLOTS; }

    MC_HD static bool is_active(const Local &l, int t) { return (l.started & ~(l.committed | l.aborted)) >> t & 1u; }  // :279
    // StartedAndCanDoPublicOperation :328-336
    MC_HD static bool can_do(const Local &l, int t) { return is_active(l, t) && t_wait(m_txn(l.meta, t)) == NOLOCK; }
    // LatestCommittedVersionOfKeyWhenTxnBegan :351-361 (-1 = {})
    MC_HD static int latest_version(const Params &p, const Local &l, int txn, int key) {
        const unsigned st = idx6(l.bidx, txn);
        int best = -1;
        unsigned bw = 0;
#pragma unroll
        for (int w = 0; w < NT; w++) {
            const unsigned wi = idx6(widx(l, key), w), ci = idx6(l.cidx, w);
            if (w < p.nt && wi && wi <= st && ci && ci <= st && wi > bw) { bw = wi; best = w; }
        }
        return best;
    }
    // VersionThatWouldBeReadBy :366-378
    MC_HD static int version_read_by(const Params &p, const Local &l, int txn, int key) {
        if (t_xl(m_txn(l.meta, txn)) >> key & 1u) return txn;
        return latest_version(p, l, txn, key);
    }
    // VersionIDsOfKeyNewerThanReadByTxn :384-399 (all later writes of key, whoever wrote them)
    MC_HD static unsigned newer_versions(const Params &p, const Local &l, int key, int ver) {
        const unsigned wv = idx6(widx(l, key), ver);
        unsigned m = 0;
#pragma unroll
        for (int w = 0; w < NT; w++) if (w < p.nt && idx6(widx(l, key), w) > wv) m |= 1u << w;
        return m;
    }
    // WritersCommittedToKeySinceTxnBegan :339-346
    MC_HD static unsigned writers_since(const Params &p, const Local &l, int txn, int key) {
        const unsigned st = idx6(l.bidx, txn);
        unsigned m = 0;
#pragma unroll
        for (int w = 0; w < NT; w++) {
            const unsigned ci = idx6(l.cidx, w);
            if (w < p.nt && ci && ci >= st && (l.wkeys >> (4 * w + key) & 1u)) m |= 1u << w;
        }
        return m;
    }
    // findConcurrentSIREADlockOwners :657-685
    MC_HD static unsigned siread_owners(const Params &p, const Local &l, int txn, int key) {
        const unsigned bt = idx6(l.bidx, txn);
        unsigned m = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const unsigned ci = idx6(l.cidx, t);
            if (t < p.nt && t != txn && (t_sir(m_txn(l.meta, t)) >> key & 1u) && (ci == 0 || ci > bt)) m |= 1u << t;
        }
        return m;
    }

    // ------------------------------------------------------------------ successor = new meta + appended events
    struct Delta {
        uint64_t meta;
        uint64_t ev;   // the appended events, 16 bits each, first one lowest (a step appends at most |TxnId| = 4: Commit + its losers)
        int nev;
    };
    MC_HD static void d_append(Delta &d, unsigned e) {
        d.ev |= (uint64_t)e << (16 * d.nev);
        d.nev++;
    }
    // internalAbort(txn, reason) :406-416
    MC_HD static void internal_abort(Delta &d, int txn, int reason) {
        d_append(d, mk_event(OP_ABORT, txn, 0, 0, reason));
        d.meta = m_set_txn(d.meta, txn, mk_txn(0, NOLOCK, 0, 0, 0));
    }
    // HelperWriteCanAcquireXLock :700-771
    MC_HD static void write_can_acquire(const Params &p, const Local &l, int txn, int key, Delta &d) {
        const unsigned owners = p.textbook ? 0u : siread_owners(p, l, txn, key);  // textbook SI: HelperWriteCanAcquireXLock :383-386
        bool danger = false;
#pragma unroll
        for (int o = 0; o < NT; o++) if ((owners >> o & 1u) && ((l.committed >> o & 1u) || t_in(m_txn(l.meta, o)))) danger = true;  // :726-728
        if (owners && danger) { internal_abort(d, txn, R_WRITE); return; }
        // snapshotIsolationWriteAction :688-691
        d_append(d, mk_event(OP_WRITE, txn, key, 0, 0));
        unsigned f = m_txn(d.meta, txn);
        f = mk_txn(t_xl(f) | 1u << key, NOLOCK, owners ? 1u : t_in(f), t_out(f), t_sir(f));
        d.meta = m_set_txn(d.meta, txn, f);
#pragma unroll
        for (int o = 0; o < NT; o++) if (owners >> o & 1u) d.meta |= 1ull << (8 + 10 * o + 6);  // outConflict'[o] = TRUE
    }

    // The next-state relation by ACTION FAMILY (each a straight-line function the by-pairs kernel, engine_pairs.h, runs 64 lanes wide;
    // compute() below dispatches a slot to its family for the slot-by-slot kernel, the writers and the host).  A family function
    // returns false when the guard fails, true with the delta otherwise; finish_delta() closes an enabled delta.
    //   family 0: Begin(txn) :423-426 | Commit(txn) :429-491 | ChooseToAbort(txn) :494-496 | LegitimateTermination :963,996
    //   family 1: Read(txn, key) :525-626
    //   family 2: StartWriteMayBlock(txn, key) :883-911 (choice c of the deadlock victim) | FinishBlockedWrite(txn) :923-927
    MC_HD static bool compute_simple(const Params &p, const Local &l, int slot, Delta &d, int &action) {
        if (slot == NT * PER_TXN) {  // LegitimateTermination /\ UNCHANGED allvars :963,996
            action = SA_TERMINATED;
            const unsigned all = (1u << p.nt) - 1u;
            return ((l.committed | l.aborted) & all) == all;
        }
        const int txn = slot / PER_TXN, sub = slot % PER_TXN;
        if (txn >= p.nt) return false;
        const unsigned me = m_txn(l.meta, txn);
        if (sub == 0) {  // Begin(txn) :423-426
            action = SA_BEGIN;
            if (l.started >> txn & 1u) return false;
            // SYMMETRY over TxnId: the representative of the successor's orbit begins the lowest unstarted transaction
            d_append(d, mk_event(OP_BEGIN, (p.sym & 1) ? (int)__builtin_popcount(l.started) : txn, 0, 0, 0));
        } else if (sub == 1) {  // Commit(txn) :429-491
            action = SA_COMMIT;
            if (!can_do(l, txn)) return false;
            if (!p.textbook && t_in(me) && t_out(me)) internal_abort(d, txn, R_COMMIT);
            else {
                d_append(d, mk_event(OP_COMMIT, txn, 0, 0, 0));
                d.meta = m_set_txn(d.meta, txn, mk_txn(0, t_wait(me), t_in(me), t_out(me), t_sir(me)));  // drop X locks only
#pragma unroll
                for (int b = 0; b < NT; b++) {  // LoserTxns :461-462, AbortOpSeq in ascending order :465-474
                    const unsigned fb = m_txn(l.meta, b), wb = t_wait(fb);
                    if (b < p.nt && wb != NOLOCK && (t_xl(me) >> wb & 1u)) {
                        d_append(d, mk_event(OP_ABORT, b, 0, 0, R_FCW));
                        d.meta = m_set_txn(d.meta, b, mk_txn(0, NOLOCK, 0, 0, 0));
                    }
                }
            }
        } else {  // ChooseToAbort(txn) :494-496
            action = SA_ABORT;
            if (!can_do(l, txn)) return false;
            internal_abort(d, txn, R_VOLUNTARY);
        }
        return true;
    }
    MC_HD static bool compute_read(const Params &p, const Local &l, int slot, Delta &d, int &action) {  // Read(txn, key) :525-626
        const int txn = slot / PER_TXN, key = slot % PER_TXN - 4;
        action = SA_READ;
        if (txn >= p.nt || key >= p.nk || !can_do(l, txn) || (l.rkeys >> (4 * txn + key) & 1u)) return false;
        const int ver = version_read_by(p, l, txn, key);
        if (ver < 0) return false;  // no version yet: in particular an untouched key is never read, so Read needs no SYMMETRY alias
        const unsigned newer = newer_versions(p, l, key, ver);
        bool danger = false;
#pragma unroll
        for (int x = 0; x < NT; x++) if ((newer >> x & 1u) && (l.committed >> x & 1u) && t_out(m_txn(l.meta, x))) danger = true;
        if (p.textbook) d_append(d, mk_event(OP_READ, txn, key, ver, 0));  // textbookSnapshotIsolation.tla:365-378
        else if (danger) internal_abort(d, txn, R_READ);
        else {
            d_append(d, mk_event(OP_READ, txn, key, ver, 0));
            unsigned lockers = 0;
#pragma unroll
            for (int x = 0; x < NT; x++) if (x < p.nt && x != txn && (t_xl(m_txn(l.meta, x)) >> key & 1u)) lockers |= 1u << x;
#pragma unroll
            for (int x = 0; x < NT; x++) if ((newer | lockers) >> x & 1u) d.meta |= 1ull << (8 + 10 * x + 5);  // inConflict'[x] = TRUE
            d.meta |= (uint64_t)(1u << key) << (8 + 10 * txn + 7);                                                // SIREAD lock
            if (newer | lockers) d.meta |= 1ull << (8 + 10 * txn + 6);                                            // outConflict'[txn]
        }
        return true;
    }
    MC_HD static bool compute_write(const Params &p, const Local &l, int slot, Delta &d, int &action) {
        const int txn = slot / PER_TXN, sub = slot % PER_TXN;
        if (txn >= p.nt) return false;
        const unsigned me = m_txn(l.meta, txn);
        unsigned anylocked = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) anylocked |= t_xl(m_txn(l.meta, t));
        if (sub == 3) {  // FinishBlockedWrite(txn) :923-927
            action = SA_FINISH;
            const unsigned key = t_wait(me);
            if (key == NOLOCK) return false;
            if (anylocked >> key & 1u) return false;
            write_can_acquire(p, l, txn, (int)key, d);
            return true;
        }
        // StartWriteMayBlock(txn, key) :883-911, choice c of the deadlock victim
        const int key = (sub - 4 - NK) / NT, c = (sub - 4 - NK) % NT;
        action = SA_WRITE;
        if (key >= p.nk || !can_do(l, txn) || (t_xl(me) >> key & 1u)) return false;
        if (p.sym & 2) {  // SYMMETRY over Key: writing an untouched key = writing the lowest untouched key
            unsigned ks = l.rkeys | l.wkeys;
            ks = (ks | ks >> 4 | ks >> 8 | ks >> 12) & 7u;
            if (!(ks >> key & 1u)) {  // untouched: nobody wrote, locked or SIREAD-locked it, so :897-911 reduces to the plain write
                if (c) return false;
                write_can_acquire(p, l, txn, (int)__builtin_popcount(ks), d);
                return true;
            }
        }
        if (writers_since(p, l, txn, key)) {  // lost First Committer Wins :897-905 (waitingForXLock unchanged = NoLock)
            if (c) return false;
            d_append(d, mk_event(OP_ABORT, txn, 0, 0, R_FCW));
            d.meta = m_set_txn(d.meta, txn, mk_txn(0, NOLOCK, 0, 0, 0));
        } else if (anylocked >> key & 1u) {  // HelperWriteConflictsWithXLock :774-880
            // follow "waits for the holder of" edges from txn (at most one per transaction)
            unsigned path = 1u << txn;  // members of pathThatCyclesFromTxnToTxn, if it cycles
            int from = txn;
            unsigned want = (unsigned)key;
            bool cycle = false;
#pragma unroll
            for (int step = 0; step < NT; step++) {
                int to = -1;
#pragma unroll
                for (int t = 0; t < NT; t++) if (to < 0 && is_active(l, t) && (t_xl(m_txn(l.meta, t)) >> want & 1u)) to = t;
                if (to < 0) break;                 // dead end: no cycle
                if (to == txn) { cycle = true; break; }
                path |= 1u << to;
                from = to;
                want = t_wait(m_txn(l.meta, from));
                if (want == NOLOCK) break;
            }
            (void)from;
            if (!cycle) {
                if (c) return false;
                d.meta = m_set_txn(d.meta, txn, mk_txn(t_xl(me), (unsigned)key, t_in(me), t_out(me), t_sir(me)));
            } else {  // \E to_abort \in Range(path) :851: c-th member in ascending order
                int victim = -1, seen = 0;
#pragma unroll
                for (int t = 0; t < NT; t++) if (path >> t & 1u) { if (seen == c) victim = t; seen++; }
                if (victim < 0) return false;
                d_append(d, mk_event(OP_ABORT, victim, 0, 0, R_DEADLOCK));
                if (victim == txn) d.meta = m_set_txn(d.meta, txn, mk_txn(0, t_wait(me), 0, 0, 0));
                else {
                    d.meta = m_set_txn(d.meta, victim, mk_txn(0, NOLOCK, 0, 0, 0));
                    d.meta = m_set_txn(d.meta, txn, mk_txn(t_xl(me), (unsigned)key, t_in(me), t_out(me), t_sir(me)));
                }
            }
        } else {
            if (c) return false;
            write_can_acquire(p, l, txn, key, d);
        }
        return true;
    }
    MC_HD static void delta_init(const Local &l, Delta &d) { d.meta = l.meta; d.nev = 0; d.ev = 0; }
    MC_HD static unsigned finish_delta(const Local &l, Delta &d) {
        if (l.n + d.nev > HCAP) return ST_ENABLED | ST_OVERFLOW;
        d.meta = (d.meta & ~63ull) | (uint64_t)(l.n + d.nev);
        return ST_ENABLED;
    }
    MC_HD static constexpr int slot_family(int slot) {
        if (slot == NT * PER_TXN) return 0;
        const int sub = slot % PER_TXN;
        if (MC_SSI_COMMIT_FAMILY && sub == 1) return 3;
        return sub < 3 ? 0 : sub == 3 ? 2 : sub < 4 + NK ? 1 : 2;
    }
    MC_HD static unsigned compute(const Params &p, const Local &l, int slot, Delta &d, int &action) {
        delta_init(l, d);
        const int f = slot_family(slot);
        const bool en = (f == 0 || f == 3) ? compute_simple(p, l, slot, d, action) : f == 1 ? compute_read(p, l, slot, d, action) : compute_write(p, l, slot, d, action);
        return en ? finish_delta(l, d) : 0u;
    }

    // the (at most two) history words the appended events land in: positions n .. n + nev - 1, four 16-bit events per word
    template <class Ref>
    MC_HD static void touched(const Local &l, Ref s, const Delta &d, int &wa, uint64_t &olda, uint64_t &newa, uint64_t &newb) {
        wa = l.n >> 2;
        olda = (d.nev && wa < HWORDS) ? s.get(W_H0 + wa) : 0;
        const int sh = 16 * (l.n & 3);
        newa = olda | d.ev << sh;
        newb = sh ? d.ev >> (64 - sh) : 0;
    }
    template <class Ref>
    MC_HD static unsigned eval(const Params &p, Local &l, Ref s, int slot, uint64_t &fp) {
        Delta d;
        int action;
        const unsigned st = compute(p, l, slot, d, action);
        if (!(st & ST_ENABLED)) return 0;
        if (st & ST_OVERFLOW) { fp = 1; return st; }
        int wa;
        uint64_t olda, newa, newb;
        touched(l, s, d, wa, olda, newa, newb);
        uint64_t f = l.fp + H(d.meta, salt_of(W_META)) - H(l.meta, salt_of(W_META));
        if (newa != olda) f += H(newa, salt_of((unsigned)(W_H0 + wa))) - H(olda, salt_of((unsigned)(W_H0 + wa)));
        if (newb) f += H(newb, salt_of((unsigned)(W_H0 + wa + 1))) - H(0, salt_of((unsigned)(W_H0 + wa + 1)));
        fp = fp_nonzero(f);
        return st;
    }
    template <class Ref>
    MC_HD static unsigned apply(const Params &p, Ref s, int slot, WordRef out) {
        Local l;
        load(p, s, l);
        Delta d;
        int action;
        const unsigned st = compute(p, l, slot, d, action);
        for (int w = 0; w < MAX_WORDS; w++) out.set(w, s.get(w));
        if (!(st & ST_ENABLED) || (st & ST_OVERFLOW)) return st;
        int wa;
        uint64_t olda, newa, newb;
        touched(l, s, d, wa, olda, newa, newb);
        out.set(W_META, d.meta);
        if (d.nev) out.set(W_H0 + wa, newa);
        if (newb) out.set(W_H0 + wa + 1, newb);
        uint64_t f = 0;
        for (int w = 1; w < MAX_WORDS; w++) f += H(out.get(w), salt_of((unsigned)w));
        out.set(W_FP, f);
        return st;
    }

    // ------------------------------------------------------------------ by-pairs protocol (engine_pairs.h: k_expand_pairs)
    // The fused kernel of round 6.  Lane = parent builds the tables once (load), checks the invariants (parent_status) and computes a
    // mask of the slots whose guard holds (guards: exact up to the deadlock-victim choices, see there); the enabled (parent, slot) PAIRS of
    // the wavefront's 64 parents are then laid out family by family in LDS and evaluated 64 at a time — every lane busy, one family's
    // code path — from the parent's Summary (the tables compute_* reads) and its row, both staged in LDS.  The same evaluation runs a
    // second time for the pairs the seen-set accepted and writes their rows (the successor is parent + new meta + <= 2 history words).
    static constexpr int PAIR_FAMILIES = MC_SSI_COMMIT_FAMILY ? 4 : 3;
#ifndef MC_SSI_BLIND
#define MC_SSI_BLIND 0
#endif
    static constexpr bool BLIND_INSERT = MC_SSI_BLIND != 0;  // 85 % of the candidates of this search are new states (G / D = 1.19 on the 4 x 3 model)
    static constexpr int PAIR_ROUNDS = NT;        // a wavefront whose pairs overflow its list works transaction by transaction
    static constexpr int PAIR_ROUND_SLOTS = PER_TXN + 1;  // ... at most this many slots per parent and round
    struct Summary {
        uint32_t masks;  // started | committed << 4 | aborted << 8
        uint32_t bidx, cidx, rkeys, wkeys, widx0, widx1, widx2;
    };
    MC_HD static void summarize(const Local &l, Summary &q) {
        q.masks = l.started | l.committed << 4 | l.aborted << 8;
        q.bidx = l.bidx; q.cidx = l.cidx; q.rkeys = l.rkeys; q.wkeys = l.wkeys;
        q.widx0 = l.widx0; q.widx1 = l.widx1; q.widx2 = l.widx2;
    }
    // the fields of Local the family functions read (the read positions / versions are the invariants' only)
    MC_HD static void local_of(const Summary &q, uint64_t fp, uint64_t meta, Local &l) {
        l.fp = fp; l.meta = meta; l.n = m_len(meta);
        l.started = q.masks & 15u; l.committed = q.masks >> 4 & 15u; l.aborted = q.masks >> 8 & 15u;
        l.bidx = q.bidx; l.cidx = q.cidx; l.rkeys = q.rkeys; l.wkeys = q.wkeys;
        l.widx0 = q.widx0; l.widx1 = q.widx1; l.widx2 = q.widx2;
        l.ridx0 = l.ridx1 = l.ridx2 = l.rver0 = l.rver1 = l.rver2 = 0; l.abort_reasons = 0; l.wellformed = true;
        l.tail_op = -1; l.tail_t = 0; l.tail_k = 0; l.tail_commit = false;
    }
    // slots of family f / of round r (transaction r; the termination slot rides with transaction 0) as 128-bit masks (lo: slots 0..63,
    // hi: 64..76); constexpr, so that the kernel's masks are immediates
    struct SlotMask { uint64_t lo, hi; };
    MC_HD static constexpr SlotMask family_mask(int f) {
        SlotMask m{0, 0};
        for (int s = 0; s < TOTAL_SLOTS; s++) {
            const int sub = s % PER_TXN;
            if (slot_family(s) == f) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64); }
        }
        return m;
    }
    MC_HD static constexpr SlotMask round_mask(int r) {
        SlotMask m{0, 0};
        for (int s = 0; s < TOTAL_SLOTS; s++)
            if (s == NT * PER_TXN ? r == 0 : s / PER_TXN == r) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64); }
        return m;
    }
    // GUARDS: bit s set <= compute(slot s) may be enabled; EXACT (bit s set <=> enabled) for every slot but the deadlock-victim choices
    // c >= 1 of StartWriteMayBlock, which are offered whenever the key's holder is itself waiting (the cycle search stays in compute_write)
    MC_HD static void guards(const Params &p, const Local &l, uint64_t &lo, uint64_t &hi) {
        lo = hi = 0;
        auto put = [&](int s) { if (s < 64) lo |= 1ull << s; else hi |= 1ull << (s - 64); };
        const unsigned all = (1u << p.nt) - 1u;
        if (((l.committed | l.aborted) & all) == all) put(NT * PER_TXN);
        unsigned anylocked = 0;
#pragma unroll
        for (int t = 0; t < NT; t++) anylocked |= t_xl(m_txn(l.meta, t));
        unsigned ks = l.rkeys | l.wkeys;  // keys touched (SYMMETRY over Key)
        ks = (ks | ks >> 4 | ks >> 8 | ks >> 12) & 7u;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (t >= p.nt) continue;
            const int b = t * PER_TXN;
            const unsigned me = m_txn(l.meta, t);
            if (!(l.started >> t & 1u)) put(b + 0);
            const unsigned wk = t_wait(me);
            if (wk != NOLOCK && !(anylocked >> wk & 1u)) put(b + 3);
            if (!can_do(l, t)) continue;
            put(b + 1);
            put(b + 2);
#pragma unroll
            for (int k = 0; k < NK; k++) {
                if (k >= p.nk) continue;
                if (!(l.rkeys >> (4 * t + k) & 1u) && version_read_by(p, l, t, k) >= 0) put(b + 4 + k);
                if (t_xl(me) >> k & 1u) continue;
                put(b + 4 + NK + NT * k);  // c = 0: always some successor (plain write / abort / block / first victim)
                if ((p.sym & 2) && !(ks >> k & 1u)) continue;
                if (!(anylocked >> k & 1u)) continue;
                bool holder_waits = false;  // a cycle through k's holder needs the holder to wait for a lock itself
#pragma unroll
                for (int o = 0; o < NT; o++) if (is_active(l, o) && (t_xl(m_txn(l.meta, o)) >> k & 1u) && t_wait(m_txn(l.meta, o)) != NOLOCK) holder_waits = true;
                if (holder_waits)
                    for (int c = 1; c < p.nt; c++) put(b + 4 + NK + NT * k + c);
            }
        }
    }
    MC_HD static bool guard_is_exact(int slot) {  // every slot but the victim choices c >= 1 of StartWriteMayBlock
        if (slot == NT * PER_TXN) return true;
        const int sub = slot % PER_TXN;
        return sub < 4 + NK || (sub - 4 - NK) % NT == 0;
    }
    // PAIR BASE: what every successor's fingerprint shares — the parent's sum without the terms of the three words a step can change
    // (meta, the history word the next event lands in, the one after it).  A pair then adds three terms instead of taking three away
    // and adding three; the parent's lane computes the base once, it waits in LDS beside the Summary.
    static constexpr int W_PAIR_BASE = W_FP;  // word of the STAGED row (k_expand_pairs' LDS copy) that holds the base instead of the parent's own sum
    template <class Ref>
    MC_HD static uint64_t pair_base(const Params &, const Local &l, Ref row) {
        const int wa = l.n >> 2;
        uint64_t b = l.fp - H(l.meta, salt_of(W_META));
        if (wa < HWORDS) b -= H(row.get(W_H0 + wa), salt_of((unsigned)(W_H0 + wa)));
        if (wa + 1 < HWORDS) b -= H(0, salt_of((unsigned)(W_H0 + wa + 1)));   // (append-only: the word behind the last event's is empty)
        return b;
    }
    // what a pair's evaluation hands to the writer: the successor's meta word and the (at most two) history words the events land in
    struct PairOut {
        uint64_t raw_fp, meta, newa, newb;  // raw_fp: word 0 of the successor (the raw additive sum; the seen-set key is fp_nonzero of it)
        int wa, nev;
    };
    // evaluation of one (parent, slot) pair of family F from the parent's Summary + row (any Ref with get(w)); fp = successor's fingerprint
    template <int F, class Ref>
    MC_HD static unsigned eval_pair(const Params &p, const Summary &q, Ref row, int slot, uint64_t &fp, PairOut &o) {
        const uint64_t base = row.get(W_PAIR_BASE);
        Local l;
        local_of(q, 0, row.get(W_META), l);
        Delta d;
        delta_init(l, d);
        int action;
        const bool en = (F == 0 || F == 3) ? compute_simple(p, l, slot, d, action) : F == 1 ? compute_read(p, l, slot, d, action) : compute_write(p, l, slot, d, action);
        if (!en) return 0;
        const unsigned st = finish_delta(l, d);
        if (st & ST_OVERFLOW) { fp = 1; return st; }
        const int wa = l.n >> 2, sh = 16 * (l.n & 3);
        const uint64_t olda = wa < HWORDS ? row.get(W_H0 + wa) : 0;
        o.wa = wa;
        o.nev = d.nev;
        o.meta = d.meta;
        o.newa = olda | d.ev << sh;
        o.newb = sh ? d.ev >> (64 - sh) : 0;
        uint64_t f = base + H(d.meta, salt_of(W_META));
        if (wa < HWORDS) f += H(o.newa, salt_of((unsigned)(W_H0 + wa)));
        if (wa + 1 < HWORDS) f += H(o.newb, salt_of((unsigned)(W_H0 + wa + 1)));
        o.raw_fp = f;
        fp = fp_nonzero(f);
        return st;
    }
    template <class Ref>
    MC_HD static void write_pair(const Params &, Ref row, const PairOut &o, WordRef out) {
        out.set(W_FP, o.raw_fp);
        out.set(W_META, o.meta);
#pragma unroll
        for (int w = 0; w < HWORDS; w++) {
            uint64_t v = row.get(W_H0 + w);
            if (o.nev && w == o.wa) v = o.newa;
            if (o.newb && w == o.wa + 1) v = o.newb;
            out.set(W_H0 + w, v);
        }
    }

    // ------------------------------------------------------------------ invariants, per expanded state
    MC_HD static bool has_cycle(const unsigned *adj_in) {  // FindAllNodesInAnyCycle(edges) /= {}  :1040-1060
        unsigned r0 = adj_in[0], r1 = adj_in[1], r2 = adj_in[2], r3 = adj_in[3];
        // Warshall over the 4 nodes: after step k, r_a holds everything reachable from a through nodes <= k
        if (r1 & 1u) r1 |= r0;
        if (r2 & 1u) r2 |= r0;
        if (r3 & 1u) r3 |= r0;
        if (r0 & 2u) r0 |= r1;
        if (r2 & 2u) r2 |= r1;
        if (r3 & 2u) r3 |= r1;
        if (r0 & 4u) r0 |= r2;
        if (r1 & 4u) r1 |= r2;
        if (r3 & 4u) r3 |= r2;
        if (r0 & 8u) r0 |= r3;
        if (r1 & 8u) r1 |= r3;
        if (r2 & 8u) r2 |= r3;
        return (r0 & 1u) || (r1 & 2u) || (r2 & 4u) || (r3 & 8u);
    }
    // ---- the invariants in pieces (parent_status: all of them; parent_status_step: those the last step can have changed)
    // CorrectnessOfHoldingXLocks :1302-1321, CorrectnessOfWaitingForXLock :1324-1330 (the lock variables) and well-formedness (from load)
    MC_HD static unsigned inv_locks(const Params &p, const Local &l) {
        if ((p.inv_mask & 1) && !l.wellformed) return ST_INVARIANT | (0u << 8);
        if (p.inv_mask & 2) {  // CorrectnessOfHoldingXLocks :1302-1321
            bool ok = true;
            unsigned seen = 0;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const unsigned xl = t_xl(m_txn(l.meta, t));
                if (seen & xl) ok = false;
                seen |= xl;
                const unsigned wk = l.wkeys >> (4 * t) & 7u;
                if (is_active(l, t) ? xl != wk : xl != 0) ok = false;
            }
            if (!ok) return ST_INVARIANT | (1u << 8);
        }
        if (p.inv_mask & 4) {  // CorrectnessOfWaitingForXLock :1324-1330
#pragma unroll
            for (int t = 0; t < NT; t++) if (t < p.nt && t_wait(m_txn(l.meta, t)) != NOLOCK && !is_active(l, t)) return ST_INVARIANT | (2u << 8);
        }
        return 0;
    }
    // CorrectReadView :1215-1268 for ONE read event: txn read key (the caller knows it did)
    MC_HD static bool read_view_ok(const Params &p, const Local &l, int txn, int key) {
        bool ok = true;
        const unsigned ir = idx6(ridx(l, key), txn), itxnb = idx6(l.bidx, txn);
        const int ver = (int)(rver(l, key) >> (2 * txn) & 3u);
        if (ver != txn) {  // only committed reads
            const unsigned irfc = idx6(l.cidx, ver);
            if (!irfc || !(irfc < itxnb)) ok = false;
        }
        const unsigned iwkv = idx6(widx(l, key), ver);
#pragma unroll
        for (int w = 0; w < NT; w++) {  // only up-to-date reads
            const unsigned wi = idx6(widx(l, key), w), ci = idx6(l.cidx, w);
            if (wi > iwkv && wi <= itxnb && ci && ci <= itxnb) ok = false;
        }
        const unsigned iw = idx6(widx(l, key), txn);
        if (iw) {  // key both read and written by txn
            if (ir < iw) { if (ver != latest_version(p, l, txn, key)) ok = false; }
            else if (ver != txn) ok = false;
        }
        return ok;
    }
    // FirstCommitterWins :1271-1278, CahillSerializable :1379-1446, BernsteinSerializable :1505-1556: statements about COMMITTED transactions
    MC_HD static unsigned inv_committed(const Params &p, const Local &l) {
        if (p.inv_mask & 16) {  // FirstCommitterWins :1271-1278 with AreConcurrent :1118-1133
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++) {
                    if (a == b || !(l.committed >> a & 1u) || !(l.committed >> b & 1u)) continue;
                    const unsigned b1 = idx6(l.bidx, a), c1 = idx6(l.cidx, a), b2 = idx6(l.bidx, b), c2 = idx6(l.cidx, b);
                    const bool conc = b1 < b2 ? c1 > b2 : c2 > b1;  // both committed, both started
                    if (conc && ((l.wkeys >> (4 * a)) & (l.wkeys >> (4 * b)) & 7u)) return ST_INVARIANT | (4u << 8);
                }
        }
        if (p.inv_mask & 32) {  // CahillSerializable :1379-1446
            unsigned adj[4] = {0, 0, 0, 0};
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < NT; b++) {
                    if (a == b || !(l.committed >> a & 1u) || !(l.committed >> b & 1u)) continue;
#pragma unroll
                    for (int x = 0; x < NK; x++) {
                        const unsigned aw = idx6(widx(l, x), a), bw = idx6(widx(l, x), b);
                        const bool ww = aw && bw && aw < bw;
                        const bool wr = aw && (l.rkeys >> (4 * b + x) & 1u) && idx6(l.cidx, a) < idx6(l.bidx, b);
                        const bool rw = (l.rkeys >> (4 * a + x) & 1u) && bw && idx6(l.bidx, a) < idx6(l.cidx, b);
                        if (ww || wr || rw) adj[a] |= 1u << b;
                    }
                }
            if (has_cycle(adj)) return ST_INVARIANT | (5u << 8);
        }
        if (p.inv_mask & 64) {  // BernsteinSerializable :1505-1556
            unsigned adj[4] = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < NT; r++)
#pragma unroll
                for (int x = 0; x < NK; x++) {
                    if (!(l.rkeys >> (4 * r + x) & 1u) || !(l.committed >> r & 1u)) continue;
                    const int j = (int)(rver(l, x) >> (2 * r) & 3u);  // r read x from j: rk[xj] with k = r
                    if (!(l.committed >> j & 1u)) continue;
                    if (j != r) adj[j] |= 1u << r;  // BernsteinSG: writer -> reader
#pragma unroll
                    for (int i = 0; i < NT; i++) {  // version-order edges
                        if (i == j || i == r || j == r || !(l.committed >> i & 1u)) continue;
                        const unsigned xi = idx6(widx(l, x), i), xj = idx6(widx(l, x), j);
                        if (!xi || !xj) continue;
                        if (xi < xj) adj[i] |= 1u << j; else adj[r] |= 1u << i;
                    }
                }
            if (has_cycle(adj)) return ST_INVARIANT | (6u << 8);
        }
        return 0;
    }
    MC_HD static unsigned inv_find(const Params &p, const Local &l) {
        if (p.find >= 1 && p.find <= 6) {  // ~AtLeastNTxnsAbortedDueToReason(1, r) :1579-1582
            if (l.abort_reasons >> (p.find - 1) & 1u) return ST_INVARIANT | (7u << 8);
        } else if (p.find == 7) {  // ~AtLeastNTxnsAreWaitingForLocks(2) :1578
            int n = 0;
#pragma unroll
            for (int t = 0; t < NT; t++) n += (t < p.nt && t_wait(m_txn(l.meta, t)) != NOLOCK) ? 1 : 0;
            if (n >= 2) return ST_INVARIANT | (7u << 8);
        }
        return 0;
    }
    // returns ST_INVARIANT | id << 8, or 0: EVERY invariant of the state, whatever its history (known-answer tests, the host)
    template <class Ref>
    MC_HD static unsigned parent_status(const Params &p, const Local &l, Ref) {
        if (const unsigned r = inv_locks(p, l)) return r;
        if (p.inv_mask & 8) {  // CorrectReadView :1215-1268
            bool ok = true;
#pragma unroll
            for (int txn = 0; txn < NT; txn++) {
#pragma unroll
                for (int key = 0; key < NK; key++) {
                    if (!(l.rkeys >> (4 * txn + key) & 1u)) continue;
                    if (!read_view_ok(p, l, txn, key)) ok = false;
                }
            }
            if (!ok) return ST_INVARIANT | (3u << 8);
        }
        if (const unsigned r = inv_committed(p, l)) return r;
        return inv_find(p, l);
    }
    // The same verdict for a state REACHED BY THE SEARCH, from what its last step can have changed (round 6; the kernels call this one:
    // engine_kernels.h stored_state_status).  Every state the search stores was generated from a state that was expanded — i.e. checked —
    // before (BFS: a violation ends the run at the end of its level), and the history is append-only, so the state's history is its
    // predecessor's, which satisfied every invariant, plus the events of ONE step at its end:
    //   * well-formedness and the two lock invariants read the lock variables: evaluated always (a few instructions);
    //   * CorrectReadView is a statement per read event, whose terms are fixed from then on — the version read, commits and writes BEFORE
    //     the reader began (a later commit lies after it) — except "key both read and written by txn", which a later WRITE of the same
    //     (txn, key) completes: it can only have changed for the (txn, key) of a last event that is a READ or a WRITE;
    //   * FirstCommitterWins and the two serialization graphs speak about COMMITTED transactions only, whose begin / commit positions,
    //     read and write sets are final: they can only have changed if the last step committed somebody — the history ends in `commit`,
    //     possibly followed by the First-Committer-Wins aborts of that commit's losers (taken conservatively: the same tail can also be
    //     a commit followed by an unrelated lost write, where re-evaluating is merely redundant);
    //   * a step that appends nothing (a write that blocks) re-evaluates its predecessor's tail: redundant, never wrong.
    // tests/_shim (PairCheck) compares this with parent_status on every state the lowering visits, violating models included.
    template <class Ref>
    MC_HD static unsigned parent_status_step(const Params &p, const Local &l, Ref) {
        if (const unsigned r = inv_locks(p, l)) return r;
        if ((p.inv_mask & 8) && (l.tail_op == OP_READ || l.tail_op == OP_WRITE) && (l.rkeys >> (4 * l.tail_t + l.tail_k) & 1u) &&
            !read_view_ok(p, l, l.tail_t, l.tail_k))
            return ST_INVARIANT | (3u << 8);
        if (l.tail_commit)
            if (const unsigned r = inv_committed(p, l)) return r;
        return inv_find(p, l);
    }

    // ------------------------------------------------------------------ host side
    static int action_of(const Params &p, const uint64_t *parent, int slot) {
        Local l;
        load(p, CWordRef{parent, 1}, l);
        Delta d;
        int action = -1;
        compute(p, l, slot, d, action);
        return action;
    }
    static const char *action_name(int a) {
        static const char *nm[] = {"Begin", "Commit", "ChooseToAbort", "Read", "StartWriteMayBlock", "FinishBlockedWrite", "Terminated"};
        return a >= 0 && a < 7 ? nm[a] : a < 0 ? "Initial predicate" : "?";
    }
    static int format(const Params &p, const uint64_t *w, char *buf, size_t cap) {
        static const char *reason[] = {"voluntary", "forced by First Committer Wins", "forced by deadlock-prevention",
                                       "in attempted commit, to preserve serializability",
                                       "in attempted read, to preserve serializability",
                                       "in attempted write, to preserve serializability", "?", "?"};
        size_t k = 0;
        auto put = [&](const char *fmt, auto... a) {
            if (k < cap) { int n = snprintf(buf + k, cap - k, fmt, a..., 0); if (n > 0) k += (size_t)n; if (k > cap) k = cap; }
        };
        const uint64_t meta = w[W_META];
        const int n = m_len(meta);
        put("/\\ history = <<");
        for (int i = 0; i < n; i++) {
            const unsigned e = (unsigned)(w[W_H0 + (i >> 2)] >> (16 * (i & 3))) & 0xffffu;
            const int op = (int)(e & 7u), t = (int)(e >> 3 & 3u) + 1, key = (int)(e >> 5 & 3u) + 1, ver = (int)(e >> 7 & 3u) + 1;
            if (i) put(", ");
            if (op == OP_BEGIN) put("[op |-> \"begin\", txnid |-> T%d]", t);
            else if (op == OP_COMMIT) put("[op |-> \"commit\", txnid |-> T%d]", t);
            else if (op == OP_ABORT) put("[op |-> \"abort\", reason |-> \"%s\", txnid |-> T%d]", reason[e >> 9 & 7u], t);
            else if (op == OP_READ) put("[key |-> K%d, op |-> \"read\", txnid |-> T%d, ver |-> T%d]", key, t, ver);
            else put("[key |-> K%d, op |-> \"write\", txnid |-> T%d]", key, t);
        }
        put(">>");
        auto keyset = [&](unsigned m) {
            put("{");
            bool first = true;
            for (int q = 0; q < p.nk; q++) if (m >> q & 1u) { put("%sK%d", first ? "" : ", ", q + 1); first = false; }
            put("}");
        };
        auto per_txn = [&](const char *title, auto f) {
            put("\n/\\ %s = (", title);
            for (int t = 0; t < p.nt; t++) { put("%sT%d :> ", t ? " @@ " : "", t + 1); f(m_txn(meta, t)); }
            put(")");
        };
        per_txn("holdingXLocks", [&](unsigned f) { keyset(t_xl(f)); });
        per_txn("waitingForXLock", [&](unsigned f) { if (t_wait(f) == NOLOCK) put("NoLock"); else put("K%d", (int)t_wait(f) + 1); });
        per_txn("inConflict", [&](unsigned f) { put("%s", t_in(f) ? "TRUE" : "FALSE"); });
        per_txn("outConflict", [&](unsigned f) { put("%s", t_out(f) ? "TRUE" : "FALSE"); });
        per_txn("holdingSIREADlocks", [&](unsigned f) { keyset(t_sir(f)); });
        return (int)k;
    }
};

}  // namespace mc
