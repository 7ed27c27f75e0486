// spec_raft.h — device lowering of examples/raft.tla (reference examples/raft.tla:110-507) under
// the model wrapper specs/MCraft.tla (+ specs/MCraft.cfg) of this repo.
//
// Packed state = WORDS 64-bit words (layout below).  Round 3 layout (W 288 -> 168 bytes for the bench model, 944 -> 264 for
// five servers): a LOG is 2 or 3 bits per client-request VALUE (the term of the entry with that value, 0 = no such entry) instead
// of a length and a list of entries — 6 bits instead of 33 for three values — so that the scalars AND the log of a server share
// one word, voterLog[i] is one word, committedLog rides in the globals word, an election record takes two words instead of 1 + n
// and allLogs four logs per word (SURVEY.md Appendix B's budget).  Every byte of W is paid three times per distinct state
// (expand reads it, materialise reads and writes it).
//
// The three set-valued history variables (messages with its monotone key set, elections, allLogs) are kept as UNORDERED slot
// arrays: equality of TLA+ values is decided by an additive (multiset) fingerprint
//     fp = sum_w Hw(logical field w) + sum_m (count(m) + 1) * H(key(m), SALT_M) + sum_e He(e) + sum_l H(log, SALT_A)
// which does not depend on slot order, so a successor is "parent with <= 2 message slots
// changed, <= 1 election appended, <= NS logs appended, one server's words and the globals
// rewritten" and its fingerprint is the parent's plus the differences — O(delta), not O(W).
// The hashed FIELDS are logical (globals, committedLog, scalars of server i, log[i], voterLog[i][j]: one salt each), whatever
// physical word they share.  Chosen so that the common deltas cost ONE hash each (H = hmum, mc_common.h):
//   * the bag is a multiset: a message contributes (count + 1) * H(key) — the "+ 1" because a key whose count fell to 0
//     stays in the bag (note 1 below) — so Send / Discard / Duplicate / Drop change the sum by +-H(key) (a new key: 2 H(key));
//   * of the globals only clientRequests and committedLogDecrease are hashed: the slot counts nMsgs / nElec / nAll are
//     functions of the three sets, which are hashed element by element;
//   * an empty voterLog entry contributes 0 (Restart / Timeout clear them, a vote sets one).
//
// Semantic notes that change counts — all kept:
//  0. raft.tla:392-393 vs :402/:75: the "already done" branch is enabled only when
//     m.mcommitIndex = commitIndex[i].
//  1. the bag is a map msg -> {0,1,2}; zero-count keys stay (raft.tla:117-129).
//  2. UpdateTerm / return-to-follower / conflict / append do not consume the message.
//  3. RequestVote(i,i) is allowed (raft.tla:209-217); AppendEntries needs i /= j (:223).
//  4. voterLog[i] @@ (j :> mlog) keeps an existing entry (raft.tla:343-344).
//  5. committedLog' = <<>> unless newCommitIndex > 1 (raft.tla:296-300).
//  6. committedLogDecrease' uses lazy \/ (raft.tla:302-303).
//  7. allLogs' = allLogs \cup {log[i]} over UNPRIMED logs (raft.tla:493).
#pragma once
#include "mc_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace mc {

// -------------------------------------------------------------------------- log encoding
// An entry is [term, value] with value = clientRequests when ClientRequest appended it (raft.tla:264-274): the counter only
// grows, so every value is created exactly once, every log is a sequence of entries with strictly increasing values, and a log
// is the map value -> term of its entry (0: none).  Log word: digit v (1-based value) at bits tb * (v - 1), tb bits each (tb = 2
// while MaxTerm <= 3, else 3); Len = number of non-zero digits; the k-th entry is the k-th non-zero digit.  An append that
// would break the order (a digit at or above the new value is set) cannot happen in raft.tla; it is reported (ST_OVERFLOW ->
// MC_EOVERFLOW), never stored wrongly.  Entries INSIDE messages keep the 6-bit form term | value << 3.
namespace rlog {
constexpr int LCAP = 5;   // values 1..5 at most (MaxClientRequests <= 6)
constexpr int LB_MAX = 15;
// (a log fits 15 bits: 32-bit arithmetic throughout — half the instructions and registers of the 64-bit words it is stored in)
MC_HD unsigned digit(uint32_t l, int v, int tb) { return (l >> (tb * (v - 1))) & ((1u << tb) - 1u); }
MC_HD int len(uint32_t l, int tb) {
    int n = 0;
#pragma unroll
    for (int v = 1; v <= LCAP; v++) n += digit(l, v, tb) != 0;
    return n;
}
MC_HD int eterm(unsigned e) { return (int)(e & 7u); }
MC_HD int evalue(unsigned e) { return (int)(e >> 3); }
MC_HD unsigned mk_entry(int term, int value) { return (unsigned)term | ((unsigned)value << 3); }
// k-th entry (1-based, k <= Len) as term | value << 3
MC_HD unsigned entry(uint32_t l, int k, int tb) {
    unsigned e = 0;
#pragma unroll
    for (int v = 1; v <= LCAP; v++) {
        const unsigned d = digit(l, v, tb);
        if (d && --k == 0) e = mk_entry((int)d, v);
    }
    return e;
}
MC_HD int last_term(uint32_t l, int tb) {                                                            // raft.tla:113
    int t = 0;
#pragma unroll
    for (int v = 1; v <= LCAP; v++) { const unsigned d = digit(l, v, tb); t = d ? (int)d : t; }
    return t;
}
MC_HD bool can_append(uint32_t l, unsigned e, int tb) { return evalue(e) >= 1 && evalue(e) <= LCAP && (l >> (tb * (evalue(e) - 1))) == 0; }
MC_HD uint32_t append(uint32_t l, unsigned e, int tb) { return l | ((uint32_t)eterm(e) << (tb * (evalue(e) - 1))); }  // Append
MC_HD uint32_t prefix(uint32_t l, int k, int tb) {  // SubSeq(l, 1, k)
    uint32_t out = 0;
#pragma unroll
    for (int v = 1; v <= LCAP; v++) {
        const unsigned d = digit(l, v, tb);
        if (d && k > 0) { out |= d << (tb * (v - 1)); --k; }
    }
    return out;
}
MC_HD uint32_t drop_last(uint32_t l, int tb) { return prefix(l, len(l, tb) - 1, tb); }
}  // namespace rlog

enum : int { R_FOLLOWER = 0, R_CANDIDATE = 1, R_LEADER = 2 };
enum : int { M_RVREQ = 0, M_RVRESP = 1, M_AEREQ = 2, M_AERESP = 3 };
enum : int { RA_RESTART, RA_TIMEOUT, RA_REQUESTVOTE, RA_BECOMELEADER, RA_CLIENTREQUEST, RA_ADVANCECOMMIT,
             RA_APPENDENTRIES, RA_RECEIVE, RA_DUPLICATE, RA_DROP };

struct RaftParams {
    int n, max_client_requests, max_term, max_log_len, max_msgs, inv_mask;
    int cm, ce, ca;  // capacities of the messages / elections / allLogs slot arrays (runtime: they size W)
    int max_keys;    // MaxMsgKeys of specs/MCraft.tla: Cardinality(DOMAIN messages) <= max_keys (0 = unbounded)
    int tb, lb;      // bits per log digit (2: MaxTerm <= 3, else 3) and per log (tb * (MaxClientRequests - 1) <= 15)
};

// A small array that is guaranteed to live in registers: explicit scalar members and
// compare-select access (no alloca, so nothing is ever indexed dynamically in scratch memory).
// With a compile-time index the chains fold away.
template <int N, class T = uint64_t>
struct RegArr {
    static_assert(N >= 1 && N <= 8, "RegArr holds 1..8 words");
    T a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    MC_HD T get(int i) const {
        T r = a0;
        if (N > 1) r = i == 1 ? a1 : r;
        if (N > 2) r = i == 2 ? a2 : r;
        if (N > 3) r = i == 3 ? a3 : r;
        if (N > 4) r = i == 4 ? a4 : r;
        if (N > 5) r = i == 5 ? a5 : r;
        if (N > 6) r = i == 6 ? a6 : r;
        if (N > 7) r = i == 7 ? a7 : r;
        return r;
    }
    MC_HD void set(int i, T v) {
        a0 = i == 0 ? v : a0;
        if (N > 1) a1 = i == 1 ? v : a1;
        if (N > 2) a2 = i == 2 ? v : a2;
        if (N > 3) a3 = i == 3 ? v : a3;
        if (N > 4) a4 = i == 4 ? v : a4;
        if (N > 5) a5 = i == 5 ? v : a5;
        if (N > 6) a6 = i == 6 ? v : a6;
        if (N > 7) a7 = i == 7 ? v : a7;
    }
};

template <int NS>
struct SpecRaft {
    using Params = RaftParams;
    // ---------------------------------------------------------------- word layout (physical)
    static constexpr int W_FP = 0;     // additive fingerprint of this state (raw sum)
    static constexpr int W_GLOB = 1;   // clientRequests[0,3) decrease[3] nMsgs[8,16) nElec[16,20) nAll[24,32) | committedLog[32,47)
    MC_HD static constexpr int W_SRV(int i) { return 2 + 2 * i; }   // scalars of server i [0, LOGSH) | log[i] [LOGSH, LOGSH + 15)
    MC_HD static constexpr int W_VL(int i) { return 3 + 2 * i; }    // voterLog[i][j] at bits [lb * j, lb * (j + 1))
    static constexpr int W_MSG0 = 2 + 2 * NS;                       // messages[cm]: 32 bits each, two per word (slot k in half k & 1 of word k >> 1)
    static constexpr int EL_WORDS = 2;                              // election: record word | evoterLog (packed like W_VL)
    MC_HD static int msg_words(const Params &p) { return (p.cm + 1) >> 1; }
    MC_HD static int W_EL0(const Params &p) { return W_MSG0 + msg_words(p); }                 // elections[ce][2]
    MC_HD static int W_ALL0(const Params &p) { return W_MSG0 + msg_words(p) + p.ce * EL_WORDS; }  // allLogs[ca]: four 16-bit slots per word
    MC_HD static int all_words(const Params &p) { return (p.ca + 3) >> 2; }
    MC_HD static int words(const Params &p) { return W_MSG0 + msg_words(p) + p.ce * EL_WORDS + all_words(p); }
    static constexpr int MAX_WORDS = W_MSG0 + 32 + 8 * EL_WORDS + 16;
    static constexpr int LOGSH = 16 + 6 * NS;  // 34 (3 servers) / 46 (5 servers): the scalars end here
    static constexpr uint64_t SVMASK = (1ull << LOGSH) - 1ull;
    static_assert(LOGSH + rlog::LB_MAX <= 64, "scalars + log of a server share one word");
    // LOGICAL fields: what the fingerprint hashes, one salt each (the numbering of the one-field-per-word layout of rounds 1-2)
    static constexpr int F_GLOB = 1, F_CLOG = 2;
    MC_HD static constexpr int F_SV(int i) { return 3 + i * (2 + NS); }
    MC_HD static constexpr int F_LOG(int i) { return F_SV(i) + 1; }
    MC_HD static constexpr int F_VLOG(int i, int j) { return F_SV(i) + 2 + j; }
    // slots: Restart NS | Timeout NS | RequestVote NS^2 | BecomeLeader NS | ClientRequest NS |
    //        AdvanceCommitIndex NS | AppendEntries NS^2 | per message k: Receive, Duplicate, Drop
    static constexpr int FIX = 5 * NS + 2 * NS * NS;
    // classes of slots whose successors are built by the same branch of compute(): the in-wave tail of the expand kernel sorts a
    // workgroup's new states by class before it writes them (engine.hip SlotClasses)
    static constexpr int NCLS = 10;
    MC_HD static int slot_class(int slot) {
        if (slot >= FIX) return 7 + (slot - FIX) % 3;  // Receive / Duplicate / Drop
        if (slot < 2 * NS) return slot < NS ? 0 : 1;   // Restart, Timeout
        if (slot < 2 * NS + NS * NS) return 2;         // RequestVote
        if (slot < 5 * NS + NS * NS) return 3 + (slot - (2 * NS + NS * NS)) / NS;  // BecomeLeader, ClientRequest, AdvanceCommitIndex
        return 6;                                      // AppendEntries
    }
    static constexpr int FIX_SLOTS = FIX;  // slots whose action and server indices are compile-time constants
    static constexpr int DENSE_SLOTS = 2 * NS;  // Restart(i), Timeout(i): enabled for (nearly) every state — the by-family kernel
                                                // evaluates them inline, lane = parent (engine.hip k_expand_family)
    static constexpr int STAGE_WORDS = 16; // message slots of each parent the expand kernel stages in LDS
    template <class Ref>
    MC_HD static void stage_range(const Params &, Ref s, int &lo, int &n) { lo = W_MSG0; n = (g_nm(s.get(W_GLOB)) + 1) >> 1; }
    MC_HD static int max_slots(const Params &p) { return FIX + 3 * p.cm; }
    static constexpr uint64_t SALT_M = 0x8f1bbcdc8f1bbcdcull, SALT_E = 0xca62c1d6ca62c1d6ull, SALT_A = 0x5a8279995a827999ull;

    // ---- field access through a state reference (Ref::get = physical word)
    template <class Ref> MC_HD static uint64_t rd_glob(Ref s) { return s.get(W_GLOB) & 0xffffffffull; }
    template <class Ref> MC_HD static uint64_t rd_clog(Ref s) { return s.get(W_GLOB) >> 32; }
    template <class Ref> MC_HD static uint64_t rd_sv(Ref s, int i) { return s.get(W_SRV(i)) & SVMASK; }
    template <class Ref> MC_HD static uint64_t rd_log(Ref s, int i) { return s.get(W_SRV(i)) >> LOGSH; }
    MC_HD static uint64_t pack_srv(uint64_t sv, uint64_t log) { return sv | (log << LOGSH); }
    MC_HD static uint64_t pack_glob(uint64_t glob, uint64_t clog) { return (glob & 0xffffffffull) | (clog << 32); }
    MC_HD static uint64_t vl_get(uint64_t word, int j, const Params &p) { return (word >> (p.lb * j)) & ((1ull << p.lb) - 1ull); }
    MC_HD static uint64_t vl_set(uint64_t word, int j, uint64_t log, const Params &p) { return word | (log << (p.lb * j)); }  // (the entry was empty)
    template <class Ref> MC_HD static uint64_t rd_msg(Ref s, int k) { return (s.get(W_MSG0 + (k >> 1)) >> (32 * (k & 1))) & 0xffffffffull; }
    MC_HD static uint64_t half_of(uint64_t word, int k) { return (word >> (32 * (k & 1))) & 0xffffffffull; }
    MC_HD static uint64_t set_half(uint64_t word, int k, uint64_t m) { return (k & 1) ? (word & 0xffffffffull) | (m << 32) : (word & ~0xffffffffull) | m; }
    template <class Ref> MC_HD static uint64_t rd_all(const Params &p, Ref s, int a) { return (s.get(W_ALL0(p) + (a >> 2)) >> (16 * (a & 3))) & 0xffffull; }

    // server scalars: term[0,3) state[3,5) votedFor[5,8) votesGranted[8,13) commitIndex[13,16)
    //                 nextIndex[j] 3 bits at 16+3j (1 .. Len + 1 <= 6), matchIndex[j] 3 bits at 16+3NS+3j
    MC_HD static int sv_term(uint64_t v) { return (int)(v & 7); }
    MC_HD static int sv_state(uint64_t v) { return (int)(v >> 3 & 3); }
    MC_HD static int sv_voted(uint64_t v) { return (int)(v >> 5 & 7); }       // 0 = Nil, j+1
    MC_HD static unsigned sv_granted(uint64_t v) { return (unsigned)(v >> 8 & 31); }
    MC_HD static int sv_commit(uint64_t v) { return (int)(v >> 13 & 7); }
    MC_HD static int sv_next(uint64_t v, int j) { return (int)(v >> (16 + 3 * j) & 7); }
    MC_HD static int sv_match(uint64_t v, int j) { return (int)(v >> (16 + 3 * NS + 3 * j) & 7); }
    MC_HD static uint64_t sv_set_term(uint64_t v, int x) { return bits_set(v, 0, 3, (uint64_t)x); }
    MC_HD static uint64_t sv_set_state(uint64_t v, int x) { return bits_set(v, 3, 2, (uint64_t)x); }
    MC_HD static uint64_t sv_set_voted(uint64_t v, int x) { return bits_set(v, 5, 3, (uint64_t)x); }
    MC_HD static uint64_t sv_set_granted(uint64_t v, unsigned x) { return bits_set(v, 8, 5, x); }
    MC_HD static uint64_t sv_set_commit(uint64_t v, int x) { return bits_set(v, 13, 3, (uint64_t)x); }
    MC_HD static uint64_t sv_set_next(uint64_t v, int j, int x) { return bits_set(v, 16 + 3 * j, 3, (uint64_t)x); }
    MC_HD static uint64_t sv_set_match(uint64_t v, int j, int x) { return bits_set(v, 16 + 3 * NS + 3 * j, 3, (uint64_t)x); }
    MC_HD static uint64_t sv_reset_leader_vars(uint64_t v, int next) {        // nextIndex = next, matchIndex = 0 for all j
        for (int j = 0; j < NS; j++) v = sv_set_match(sv_set_next(v, j, next), j, 0);
        return v;
    }
    // globals
    MC_HD static int g_creq(uint64_t g) { return (int)(g & 7); }
    MC_HD static int g_decr(uint64_t g) { return (int)(g >> 3 & 1); }
    MC_HD static int g_nm(uint64_t g) { return (int)(g >> 8 & 255); }
    MC_HD static int g_ne(uint64_t g) { return (int)(g >> 16 & 15); }
    MC_HD static int g_na(uint64_t g) { return (int)(g >> 24 & 255); }
    // message word: count[0,2) type[2,4) term[4,7) src[7,10) dst[10,13) payload[13,..)
    MC_HD static int m_count(uint64_t m) { return (int)(m & 3); }
    MC_HD static int m_type(uint64_t m) { return (int)(m >> 2 & 3); }
    MC_HD static int m_term(uint64_t m) { return (int)(m >> 4 & 7); }
    MC_HD static int m_src(uint64_t m) { return (int)(m >> 7 & 7); }
    MC_HD static int m_dst(uint64_t m) { return (int)(m >> 10 & 7); }
    MC_HD static uint64_t m_head(int type, int term, int src, int dst) {
        return ((uint64_t)type << 2) | ((uint64_t)term << 4) | ((uint64_t)src << 7) | ((uint64_t)dst << 10);
    }
    // RVReq : lastLogTerm[13,16) lastLogIndex[16,19)
    // RVResp: granted[13] mlog[14,27)
    // AEReq : prevIdx[13,16) commitIdx[16,19) mlog[19,32) — mprevLogTerm and mentries are FUNCTIONS of (mlog, mprevLogIndex)
    //         (raft.tla:225-244: prevLogTerm = log[i][prevLogIndex].term, entries = SubSeq(log[i], nextIndex, lastEntry), mlog =
    //         log[i]), so the record is determined by the fields kept: 32 bits per message, two per word
    // AEResp: success[13] matchIndex[14,17)
    MC_HD static uint64_t mk_rvreq(int term, int llt, int lli, int src, int dst) {
        return m_head(M_RVREQ, term, src, dst) | ((uint64_t)llt << 13) | ((uint64_t)lli << 16);
    }
    MC_HD static uint64_t mk_rvresp(int term, int granted, uint64_t mlog, int src, int dst) {
        return m_head(M_RVRESP, term, src, dst) | ((uint64_t)granted << 13) | (mlog << 14);
    }
    MC_HD static uint64_t mk_aereq(int term, int pidx, int cidx, uint64_t mlog, int src, int dst) {
        return m_head(M_AEREQ, term, src, dst) | ((uint64_t)pidx << 13) | ((uint64_t)cidx << 16) | (mlog << 19);
    }
    // the derived fields of an AppendEntriesRequest
    MC_HD static int ae_pidx(uint64_t m) { return (int)(m >> 13 & 7); }
    MC_HD static int ae_cidx(uint64_t m) { return (int)(m >> 16 & 7); }
    MC_HD static uint32_t ae_mlog(uint64_t m) { return (uint32_t)(m >> 19) & 0x1fffu; }
    MC_HD static int ae_pterm(uint64_t m, int tb) { const int p = ae_pidx(m); return p > 0 ? rlog::eterm(rlog::entry(ae_mlog(m), p, tb)) : 0; }
    MC_HD static int ae_nent(uint64_t m, int tb) { return ae_pidx(m) + 1 <= rlog::len(ae_mlog(m), tb) ? 1 : 0; }
    MC_HD static unsigned ae_ent(uint64_t m, int tb) { return ae_nent(m, tb) ? rlog::entry(ae_mlog(m), ae_pidx(m) + 1, tb) : 0u; }
    MC_HD static uint32_t rv_mlog(uint64_t m) { return (uint32_t)(m >> 14) & 0x1fffu; }
    MC_HD static uint64_t mk_aeresp(int term, int success, int midx, int src, int dst) {
        return m_head(M_AERESP, term, src, dst) | ((uint64_t)success << 13) | ((uint64_t)midx << 14);
    }
    // election word 0: eterm[0,3) eleader[3,6) evotes[6,11) elog[11,26); word 1: evoterLog, packed like voterLog[i]
    MC_HD static uint64_t helec(const RegArr<EL_WORDS> &ew) { return hmum(ew.get(1) + hmum(ew.get(0), SALT_E), SALT_E + 1ull); }

    // contribution of logical field f holding x
    static constexpr uint64_t GLOB_HASHED = 0xffull;  // clientRequests[0,3) decrease[3]
    MC_HD static uint64_t hvlog(uint64_t x, int i, int j) { return x ? hmum(x, salt_of((unsigned)F_VLOG(i, j))) : 0ull; }
    MC_HD static uint64_t hfield(int f, uint64_t x) { return hmum(x, salt_of((unsigned)f)); }
    // sum over the fields of the header (everything but the three slot arrays)
    template <class Ref>
    MC_HD static uint64_t hheader(const Params &prm, Ref s) {
        uint64_t fp = hfield(F_GLOB, rd_glob(s) & GLOB_HASHED) + hfield(F_CLOG, rd_clog(s));
        for (int i = 0; i < NS; i++) {
            fp += hfield(F_SV(i), rd_sv(s, i)) + hfield(F_LOG(i), rd_log(s, i));
            const uint64_t vw = s.get(W_VL(i));
            for (int j = 0; j < NS; j++) fp += hvlog(vl_get(vw, j, prm), i, j);
        }
        return fp;
    }
    // contribution of one message slot: (count + 1) * H(key)
    MC_HD static uint64_t hkey(uint64_t mword) { return hmum(mword >> 2, SALT_M); }
    MC_HD static uint64_t hmsg(uint64_t mword) { return hkey(mword) * (uint64_t)(m_count(mword) + 1); }

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        if (np < 5) return -1;
        o.n = (int)p[0]; o.max_client_requests = (int)p[1]; o.max_term = (int)p[2];
        o.max_log_len = (int)p[3]; o.max_msgs = (int)p[4];
        o.inv_mask = np > 5 ? (int)p[5] : 1;
        o.cm = np > 6 && p[6] > 0 ? (int)p[6] : 40;
        o.ce = np > 7 && p[7] > 0 ? (int)p[7] : 4;
        o.ca = np > 8 && p[8] > 0 ? (int)p[8] : 16;
        o.max_keys = np > 9 && p[9] > 0 ? (int)p[9] : 0;
        if (o.cm > 64 || o.ce > 8 || o.ca > 64 || o.max_keys > 255) return -1;
        if (o.n != NS) return -1;
        if (o.max_client_requests < 1 || o.max_client_requests - 1 > rlog::LCAP || o.max_client_requests > 7) return -1;
        if (o.max_term < 1 || o.max_term > 6) return -1;  // term MaxTerm+1 must still fit 3 bits
        if (o.max_log_len < 0 || o.max_msgs < 0) return -1;
        o.tb = o.max_term <= 3 ? 2 : 3;                       // an entry's term is a leader's currentTerm <= MaxTerm
        o.lb = o.tb * (o.max_client_requests - 1);           // values 1 .. MaxClientRequests - 1
        if (o.lb < o.tb) o.lb = o.tb;
        if (o.lb > 13 || NS * o.lb > 64) return -1;  // a message is 32 bits (AppendEntriesRequest: 19 + lb); voterLog[i] is one word
        return 0;
    }

    // ---------------------------------------------------------------- Init   raft.tla:156-179
    MC_HD static uint64_t num_init(const Params &) { return 1; }
    MC_HD static void init(const Params &prm, uint64_t, WordRef out) {
        for (int w = 0; w < words(prm); w++) out.set(w, 0);
        uint64_t sv = sv_reset_leader_vars(sv_set_term(0, 1), 1);  // currentTerm 1, Follower, Nil, {}, 0, next 1, match 0
        for (int i = 0; i < NS; i++) out.set(W_SRV(i), pack_srv(sv, 0));
        out.set(W_GLOB, pack_glob(1, 0));  // clientRequests = 1, committedLog = <<>>
        out.set(W_FP, hheader(prm, out));
    }
    template <class Ref>
    MC_HD static uint64_t fp_of(const Params &, Ref s) { return fp_nonzero(s.get(W_FP)); }
    // full recomputation of the fingerprint (tests: must equal the incrementally maintained one)
    template <class Ref>
    MC_HD static uint64_t fp_recompute(const Params &prm, Ref s) {
        uint64_t fp = hheader(prm, s);
        const uint64_t g = rd_glob(s);
        for (int k = 0; k < g_nm(g); k++) fp += hmsg(rd_msg(s, k));
        for (int e = 0; e < g_ne(g); e++) {
            RegArr<EL_WORDS> ew;
#pragma unroll
            for (int q = 0; q < EL_WORDS; q++) ew.set(q, s.get(W_EL0(prm) + e * EL_WORDS + q));
            fp += helec(ew);
        }
        for (int a = 0; a < g_na(g); a++) fp += hmum(rd_all(prm, s, a), SALT_A);
        return fp;
    }
    template <class Ref>
    MC_HD static unsigned init_status(const Params &, Ref) { return ST_ENABLED; }

    // Per-parent SIGNATURES of the message keys: one byte per message slot k < SIG_SLOTS (7 bits folded from the key; 0x80 =
    // no message in the slot), slot k in byte k & 3 of word k >> 2.  Send(m) must find m's slot in the bag: comparing the 16
    // bytes at once (SWAR) names the candidate slot, and ONE load verifies it — instead of a scan of the whole bag, whose
    // loads the evaluating lane had to take one after the other from L2 (up to MaxMsgKeys round trips per batch of pairs).
    static constexpr int SIG_SLOTS = 16;
    static constexpr uint32_t SIG_EMPTY = 0x80808080u;
    struct Sigs { uint32_t w0, w1, w2, w3; };  // plain data: it lives in LDS (Summary)
    MC_HD static Sigs sigs_empty() { return Sigs{SIG_EMPTY, SIG_EMPTY, SIG_EMPTY, SIG_EMPTY}; }
    MC_HD static uint32_t key_sig(uint64_t word) {
        uint32_t h = (uint32_t)(word >> 2) ^ (uint32_t)(word >> 34);
        h ^= h >> 16;
        return (h ^ (h >> 7) ^ (h >> 14)) & 0x7fu;
    }
    // bit 8 * j + h set: byte j of word h equals sg, i.e. slot 4 * h + j is a candidate
    MC_HD static uint32_t sig_match(const Sigs &g, uint32_t sg) {
        uint32_t v = sg | (sg << 8);
        v |= v << 16;
        uint32_t m = 0, t, y;
        t = g.w0 ^ v; y = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t; m |= (~y & 0x80808080u) >> 7;
        t = g.w1 ^ v; y = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t; m |= (~y & 0x80808080u) >> 6;
        t = g.w2 ^ v; y = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t; m |= (~y & 0x80808080u) >> 5;
        t = g.w3 ^ v; y = ((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t; m |= (~y & 0x80808080u) >> 4;
        return m;
    }
    MC_HD static int sig_slot_of_bit(int b) { return ((b & 7) << 2) | (b >> 3); }

    // ---------------------------------------------------------------- per-parent cache
    struct Local {
        uint64_t fp, glob, clog;
        RegArr<NS> sv;
        RegArr<NS, uint32_t> log;   // (15 bits each)
        int nm, inflight;
        Sigs sig;              // signatures of the message keys (slots < SIG_SLOTS)
        unsigned addmask;      // servers whose log is not yet in allLogs (raft.tla:493), first occurrence only
        int nadd;              // popcount(addmask)
        uint64_t add_fp;       // sum of their contributions
        // (precomputing H(glob), H(sv[i]), ... once per parent was measured SLOWER — +10 % expand,
        //  +40 % materialise: most successors touch few words, and the registers cost occupancy)
        int cache_k;           // message slot whose H(word) is cached (Receive / Duplicate / Drop share it)
        uint64_t cache_hm;
        RegArr<NS> vlh;        // sum_j H(voterLog[i][j]) per server and ...
        unsigned vany;         // ... bit i: voterLog[i] has an entry (dense Restart / Timeout pairs; filled when WANT_FP)
        uint64_t dig;          // 11 bits per server: state[0,2) currentTerm[2,5) LastTerm(log)[5,8) Len(log)[8,11) — all that
                               // RequestVote(i, j) reads of server i (raft.tla:209-217): a quarter of all pairs needs no arena word
    };
    MC_HD static uint32_t digest_of(uint64_t sv, uint32_t lg, int tb) {
        return (uint32_t)sv_state(sv) | (uint32_t)sv_term(sv) << 2 | (uint32_t)rlog::last_term(lg, tb) << 5 | (uint32_t)rlog::len(lg, tb) << 8;
    }
    // WANT_FP = false: the caller never computes a fingerprint from this cache (k_materialise with a known one)
    template <bool WANT_FP = true, class Ref>
    MC_HD static void load(const Params &prm, Ref s, Local &l) {
        l.fp = s.get(W_FP);
        const uint64_t gw = s.get(W_GLOB);
        l.glob = gw & 0xffffffffull;
        l.clog = gw >> 32;
        #pragma unroll
        for (int i = 0; i < NS; i++) { const uint64_t x = s.get(W_SRV(i)); l.sv.set(i, x & SVMASK); l.log.set(i, (uint32_t)(x >> LOGSH)); }
        l.nm = g_nm(l.glob);
        l.inflight = 0;
        l.sig = sigs_empty();
        // four slots per round: the four loads are in flight together (a slot index is clamped into the bag, never past it)
        for (int k0 = 0; k0 < l.nm; k0 += 4) {
            const bool v1 = k0 + 1 < l.nm, v2 = k0 + 2 < l.nm, v3 = k0 + 3 < l.nm;
            const uint64_t p0 = s.get(W_MSG0 + (k0 >> 1)), p1 = s.get(W_MSG0 + (v2 ? (k0 >> 1) + 1 : (k0 >> 1)));  // (k0 is a multiple of 4)
            const uint64_t x0 = p0 & 0xffffffffull, x1 = p0 >> 32, x2 = p1 & 0xffffffffull, x3 = p1 >> 32;
            l.inflight += m_count(x0) + (v1 ? m_count(x1) : 0) + (v2 ? m_count(x2) : 0) + (v3 ? m_count(x3) : 0);
            uint32_t w = key_sig(x0) | 0x80808000u;
            if (k0 + 1 < l.nm) w = (w & ~0x0000ff00u) | (key_sig(x1) << 8);
            if (k0 + 2 < l.nm) w = (w & ~0x00ff0000u) | (key_sig(x2) << 16);
            if (k0 + 3 < l.nm) w = (w & ~0xff000000u) | (key_sig(x3) << 24);
            if (k0 == 0) l.sig.w0 = w;
            if (k0 == 4) l.sig.w1 = w;
            if (k0 == 8) l.sig.w2 = w;
            if (k0 == 12) l.sig.w3 = w;
        }
        // allLogs' = allLogs \cup {log[i] : i \in Server} — the same for every successor of this state
        const unsigned present = all_present(prm, s, l, g_na(l.glob));
        l.vany = 0;
        if (WANT_FP) {
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const uint64_t vw = s.get(W_VL(i));
                if (vw) l.vany |= 1u << i;
                uint64_t h = 0;
#pragma unroll
                for (int j = 0; j < NS; j++) h += hvlog(vl_get(vw, j, prm), i, j);
                l.vlh.set(i, h);
            }
        }
        finish_local<WANT_FP>(prm, l, present);
    }
    // bit i: log[i] is an element of allLogs (four 16-bit slots per word)
    template <class Ref>
    MC_HD static unsigned all_present(const Params &prm, Ref s, const Local &l, int na) {
        unsigned present = 0;
        const int wall = W_ALL0(prm);
        for (int a0 = 0; a0 < na; a0 += 4) {
            const uint64_t y = s.get(wall + (a0 >> 2));
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (a0 + u < na) {
                    const uint64_t x = (y >> (16 * u)) & 0xffffull;
#pragma unroll
                    for (int i = 0; i < NS; i++) if (x == l.log.get(i)) present |= 1u << i;
                }
        }
        return present;
    }
    template <bool WANT_FP>
    MC_HD static void finish_local(const Params &prm, Local &l, unsigned present) {
        l.cache_k = -1;
        l.cache_hm = 0;
        l.addmask = 0;
        l.nadd = 0;
        l.add_fp = 0;
        l.dig = 0;
#pragma unroll
        for (int i = 0; i < NS; i++) l.dig |= (uint64_t)digest_of(l.sv.get(i), l.log.get(i), prm.tb) << (11 * i);
#pragma unroll
        for (int i = 0; i < NS; i++) {
            bool skip = (present >> i & 1) != 0;
#pragma unroll
            for (int q = 0; q < i; q++) skip |= l.log.get(q) == l.log.get(i);  // an equal log of a lower server is added instead
            if (!skip) {
                l.addmask |= 1u << i;
                l.nadd++;
                if (WANT_FP) l.add_fp += hmum(l.log.get(i), SALT_A);
            }
        }
    }
    MC_HD static int nslots(const Params &, const Local &l) { return FIX + 3 * l.nm; }

    template <class Ref>
    MC_HD static unsigned parent_status(const Params &, const Local &, Ref) { return 0; }  // invariants are checked per successor

    // ---------------------------------------------------------------- successor delta
    struct Delta {
        uint64_t glob, clog;
        int srv;                 // the one server whose words change (-1: none)
        uint64_t sv, log;        // its new scalars / log
        uint64_t osv, olog;      // ... and the old ones
        bool pre;                // srv is a compile-time server: use the parent's precomputed hashes
        int vmode;               // 0 voterLog[srv] unchanged, 1 cleared, 2 one entry set
        int vj; uint64_t vlog;
        int nmop;                // message slots rewritten: op A (send) and op B (discard / dup)
        int midxA, midxB;        // midx == nm: appended
        uint64_t moldA, mnewA, moldB, mnewB;
        bool eadd;
        RegArr<EL_WORDS> ew;
        uint64_t ovl;            // the packed voterLog word of server srv (read when vmode != 0)
        int dinflight;
    };

    // Send(m) / WithMessage   raft.tla:117-121,138: increment (saturating at 2) or add with count 1
    template <class Ref>
    MC_HD static unsigned send(const Local &l, Ref s, const Params &prm, uint64_t key_word /*count bits zero*/, Delta &d) {
        int idx = l.nm;
        uint64_t old = 0;
        uint32_t cand = sig_match(l.sig, key_sig(key_word));  // slots whose signature equals the key's: almost always 0 or 1
        while (cand) {
            const int k = sig_slot_of_bit(__builtin_ctz(cand));
            cand &= cand - 1;
            const uint64_t x = rd_msg(s, k);
            if ((x >> 2) == (key_word >> 2)) { idx = k; old = x; cand = 0; }
        }
        for (int k = SIG_SLOTS; k < l.nm; k++) {  // bags beyond the signature words: scanned
            const uint64_t x = rd_msg(s, k);
            if ((x >> 2) == (key_word >> 2)) { idx = k; old = x; }
        }
        // op A; the discard of a Reply is op B (response key /= request key, so the slots differ)
        d.nmop |= 1;
        d.midxA = idx;
        d.moldA = old;
        if (idx < l.nm) {
            const int c = m_count(old);
            d.mnewA = c < 2 ? old + 1 : old;
            d.dinflight += c < 2 ? 1 : 0;
            return 0u;
        }
        d.mnewA = key_word | 1;
        d.dinflight += 1;
        // a new key beyond MaxMsgKeys leaves the model (checked by the caller's bookkeeping): not a capacity overflow
        if (prm.max_keys && idx >= prm.max_keys) return 0u;
        return idx >= prm.cm ? (unsigned)ST_OVERFLOW : 0u;
    }
    // Discard(m) / WithoutMessage on the slot the message was read from   raft.tla:125-129,142
    MC_HD static void discard(int k, uint64_t mword, Delta &d) {
        d.nmop |= 2;
        d.midxB = k;
        d.moldB = mword;
        const int c = m_count(mword);
        d.mnewB = c > 0 ? mword - 1 : mword;
        d.dinflight -= c > 0 ? 1 : 0;
    }

    MC_HD static bool in_quorum(unsigned set) { return 2 * __builtin_popcount(set) > NS; }  // raft.tla:110

    // compute the successor of `slot`; returns status bits (0 = not enabled).  When `slot` is a
    // compile-time constant (the unrolled FIX_SLOTS part of the expand kernel) every server index
    // below folds to a constant and the pick()s disappear.
    // MEM = true (k_materialise: `slot` differs per lane): server words are read from the state
    // itself instead of select-indexing the register copy, which would be demoted to scratch.
    template <bool MEM, class Ref>
    MC_HD static uint64_t srv_word(const Local &l, Ref s, int i) { if (MEM) return rd_sv(s, i); return l.sv.get(i); }
    template <bool MEM, class Ref>
    MC_HD static uint64_t log_word(const Local &l, Ref s, int i) { if (MEM) return rd_log(s, i); return l.log.get(i); }

    // Action FAMILIES: the expand-by-family kernel buckets enabled (state, slot) pairs of the SPARSE fixed slots per family in LDS
    // and evaluates 64 pairs of ONE family at a time, so the successor construction below runs with every lane busy and without
    // divergence between action types.  The first NFAM ids are the QUEUES: RequestVote(i, j), AppendEntries(i, j), and F_MISC for
    // the three rare kinds BecomeLeader / ClientRequest / AdvanceCommitIndex (one batch with divergent arithmetic instead of three
    // nearly empty ones).  Not queued: Restart(i) / Timeout(i) — enabled for nearly every state, evaluated by the parent's lane
    // (eval_dense) — and Receive / Duplicate / Drop, evaluated by the parent's lane per IN-FLIGHT message (eval, slot >= FIX: at
    // most MaxMsgs messages of a bag have a copy in flight).  FAM < 0 = any kind (eval, apply).
    enum : int { F_REQVOTE, F_APPEND, F_MISC, NFAM, F_BECOME = NFAM, F_CLIENT, F_ADVANCE };
#define MC_FAM(f) (FAM < 0 || FAM == (f) || (FAM == F_MISC && (f) >= NFAM))

    // The function is staged so that the memory round trips of a pair do not depend on its kind: (1) the message word of a
    // message slot, (2) the scalars and the log of the ONE server the action is about, (3) arithmetic only — a branch per kind,
    // which records the message it wants to send instead of sending it — (4) one Send: signature match + one verifying load.
    // A wavefront evaluating pairs of several kinds (F_MISC, k_materialise) walks the branches of (3) one after the other, but
    // its loads are issued once, before and after them.
    template <bool MEM = false, int FAM = -1, class Ref>
    MC_HD static unsigned compute(const Params &prm, const Local &l, Ref s, int slot, Delta &d, int &action) {
        d.glob = l.glob; d.clog = l.clog; d.srv = -1; d.sv = d.osv = 0; d.log = d.olog = 0; d.pre = true; d.vmode = 0; d.vj = 0; d.vlog = 0;
        d.nmop = 0; d.midxA = d.midxB = -1; d.moldA = d.mnewA = d.moldB = d.mnewB = 0;
        d.eadd = false; d.dinflight = 0; d.ovl = 0;
        d.ew = RegArr<EL_WORDS>();
        unsigned st = ST_ENABLED;
        const int tb = prm.tb;
        // ---- (1), (2): operands
        constexpr bool MSG_KINDS = FAM < 0;  // the message kinds are never queued
        int i = 0, j = 0, k = -1, kind = -1;
        uint64_t m = 0;
        if (slot >= FIX) {
            if (!MSG_KINDS) return 0;
            const int q = slot - FIX;
            k = q / 3; kind = q % 3;
            if (k >= l.nm) return 0;
            m = rd_msg(s, k);
            i = m_dst(m); j = m_src(m);
        } else if (slot < 2 * NS) {
            i = slot < NS ? slot : slot - NS;
        } else if (slot < 2 * NS + NS * NS) {
            const int q = slot - 2 * NS; i = q / NS; j = q % NS;
        } else if (slot < 5 * NS + NS * NS) {
            i = (slot - (2 * NS + NS * NS)) % NS;
        } else {
            const int q = slot - (5 * NS + NS * NS); i = q / NS; j = q % NS;
        }
        // (slot >= FIX: i differs per lane, the words are read from the read-only state, never select-indexed from the register copy)
        // Duplicate / Drop are about the message alone; a RequestVote pair evaluated by another lane (MEM) reads the digest
        constexpr bool BY_DIGEST = MEM && FAM == F_REQVOTE;
        // (scalars and log of a server are ONE physical word: one load)
        const uint64_t svw = slot >= FIX ? (kind == 0 ? s.get(W_SRV(i)) : 0ull) : (BY_DIGEST || !MEM) ? 0ull : s.get(W_SRV(i));
        const uint64_t svi = (slot >= FIX || MEM) ? svw & SVMASK : l.sv.get(i);
        const uint64_t lgi = (slot >= FIX || MEM) ? svw >> LOGSH : l.log.get(i);
        bool want_send = false;
        uint64_t skey = 0;
        // ---- (3): the action
        if (FAM < 0 && slot < NS) {  // Restart(i)   raft.tla:186-194 (always enabled)
            action = RA_RESTART;
            d.srv = i; d.osv = svi; d.olog = d.log = lgi;
            d.sv = sv_reset_leader_vars(sv_set_commit(sv_set_granted(sv_set_state(d.osv, R_FOLLOWER), 0), 0), 1);
            d.vmode = 1; d.ovl = s.get(W_VL(i));
        } else if (FAM < 0 && slot >= NS && slot < 2 * NS) {  // Timeout(i)   raft.tla:197-206
            action = RA_TIMEOUT;
            const int stt = sv_state(svi);
            if (!(stt == R_FOLLOWER || stt == R_CANDIDATE)) return 0;
            const int nt = sv_term(svi) + 1;
            if (nt > prm.max_term) st |= ST_OUT_OF_MODEL;
            d.srv = i; d.osv = svi; d.olog = d.log = lgi;
            d.sv = sv_set_granted(sv_set_voted(sv_set_term(sv_set_state(svi, R_CANDIDATE), nt & 7), 0), 0);
            d.vmode = 1; d.ovl = s.get(W_VL(i));
        } else if (MC_FAM(F_REQVOTE) && slot >= 2 * NS && slot < 2 * NS + NS * NS) {  // RequestVote(i, j)   raft.tla:209-217
            action = RA_REQUESTVOTE;
            const unsigned dg = BY_DIGEST ? (unsigned)(l.dig >> (11 * i)) & 2047u : (unsigned)digest_of(svi, lgi, tb);
            if ((int)(dg & 3u) != R_CANDIDATE) return 0;
            want_send = true;
            skey = mk_rvreq((int)(dg >> 2 & 7u), (int)(dg >> 5 & 7u), (int)(dg >> 8 & 7u), i, j);
        } else if (MC_FAM(F_BECOME) && slot >= 2 * NS + NS * NS && slot < 3 * NS + NS * NS) {  // BecomeLeader(i)   raft.tla:247-261
            action = RA_BECOMELEADER;
            if (sv_state(svi) != R_CANDIDATE || !in_quorum(sv_granted(svi))) return 0;
            d.srv = i; d.osv = svi; d.olog = d.log = lgi;
            d.sv = sv_reset_leader_vars(sv_set_state(svi, R_LEADER), rlog::len(lgi, tb) + 1);
            const uint64_t ew0 = (uint64_t)sv_term(svi) | ((uint64_t)i << 3) | ((uint64_t)sv_granted(svi) << 6) | (lgi << 11);
            d.ew.set(0, ew0);
            d.ew.set(1, s.get(W_VL(i)));
            // elections \cup {...}: a set — an identical record changes nothing
            bool present = false;
            const int ne = g_ne(l.glob), wel = W_EL0(prm);
            for (int e = 0; e < ne; e++) {
                bool eq = true;
#pragma unroll
                for (int q = 0; q < EL_WORDS; q++) eq &= s.get(wel + e * EL_WORDS + q) == d.ew.get(q);
                present |= eq;
            }
            if (!present) {
                d.eadd = true;
                if (ne >= prm.ce) st |= ST_OVERFLOW;
                d.glob += 1ull << 16;
            }
        } else if (MC_FAM(F_CLIENT) && slot >= 3 * NS + NS * NS && slot < 4 * NS + NS * NS) {  // ClientRequest(i)   raft.tla:264-274
            action = RA_CLIENTREQUEST;
            const int creq = g_creq(l.glob);
            if (sv_state(svi) != R_LEADER || !(creq < prm.max_client_requests)) return 0;
            if (!rlog::can_append(lgi, rlog::mk_entry(sv_term(svi), creq), tb)) return ST_ENABLED | ST_OVERFLOW;
            d.srv = i; d.osv = d.sv = svi; d.olog = lgi;
            d.log = rlog::append(lgi, rlog::mk_entry(sv_term(svi), creq), tb);
            d.glob += 1;  // clientRequests' = clientRequests + 1
            if (rlog::len(d.log, tb) > prm.max_log_len) st |= ST_OUT_OF_MODEL;
        } else if (MC_FAM(F_ADVANCE) && slot >= 4 * NS + NS * NS && slot < 5 * NS + NS * NS) {  // AdvanceCommitIndex(i)   raft.tla:280-305
            action = RA_ADVANCECOMMIT;
            const uint64_t lg = lgi;
            if (sv_state(svi) != R_LEADER) return 0;
            int maxAgree = 0;
            const int lglen = rlog::len(lg, tb);
            for (int index = 1; index <= lglen; index++) {
                unsigned agree = 1u << i;  // Agree(index) == {i} \cup {k : matchIndex[i][k] >= index}
#pragma unroll
                for (int kk = 0; kk < NS; kk++) if (sv_match(svi, kk) >= index) agree |= 1u << kk;
                if (in_quorum(agree)) maxAgree = index;
            }
            const int nci = (maxAgree > 0 && rlog::eterm(rlog::entry(lg, maxAgree, tb)) == sv_term(svi)) ? maxAgree : sv_commit(svi);
            uint64_t ncl = 0;
            if (nci > 1) {
                if (nci > lglen) return ST_ENABLED | ST_SPECERR;  // log[i][j] out of domain
                ncl = rlog::prefix(lg, nci, tb);
            }
            const int lc = rlog::len(l.clog, tb);
            bool decr = nci < lc;  // lazy \/ : the \E is evaluated only when nci >= Len(committedLog)
            // \E index \in 1..Len(committedLog) : committedLog[index] /= newCommittedLog[index] — here Len(newCommittedLog) >= lc
            if (!decr && lc > 0) decr = rlog::prefix(ncl, lc, tb) != l.clog;
            d.srv = i; d.osv = svi; d.olog = d.log = lg; d.sv = sv_set_commit(svi, nci);
            d.clog = ncl;
            d.glob = bits_set(d.glob, 3, 1, decr ? 1 : 0);
        } else if (MC_FAM(F_APPEND) && slot >= 5 * NS + NS * NS && slot < FIX) {  // AppendEntries(i, j)   raft.tla:222-244
            action = RA_APPENDENTRIES;
            const uint64_t lg = lgi;
            if (i == j || sv_state(svi) != R_LEADER) return 0;
            const int next = sv_next(svi, j), prevIdx = next - 1;
            int prevTerm = 0;
            const int lglen = rlog::len(lg, tb);
            if (prevIdx > 0 && prevIdx > lglen) return ST_ENABLED | ST_SPECERR;  // log[i][prevLogIndex] out of domain
            (void)prevTerm;  // (prevLogTerm and the entries are functions of (log[i], prevLogIndex): not stored in the message word)
            const int lastEntry = lglen < next ? lglen : next;  // Min({Len(log[i]), nextIndex[i][j]})
            const int ci = sv_commit(svi) < lastEntry ? sv_commit(svi) : lastEntry;
            want_send = true;
            skey = mk_aereq(sv_term(svi), prevIdx, ci, lg, i, j);
        } else if (MSG_KINDS && slot >= FIX) {
            const int cnt = m_count(m);
            if (kind == 1) {  // DuplicateMessage(m), m \in SingleMessage(messages)   raft.tla:134-135,471-473
                action = RA_DUPLICATE;
                if (cnt != 1) return 0;
                d.nmop = 2; d.midxB = k; d.moldB = m; d.mnewB = m + 1; d.dinflight = 1;
            } else if (kind == 2) {  // DropMessage(m), m \in ValidMessage(messages)   raft.tla:131-132,476-478
                action = RA_DROP;
                if (cnt == 0) return 0;
                discard(k, m, d);
            } else if (kind == 0) {  // Receive(m), m \in ValidMessage(messages)   raft.tla:449-464
                action = RA_RECEIVE;
                if (cnt == 0) return 0;
                const int mterm = m_term(m), type = m_type(m);
                const uint64_t lg = lgi;
                const int term = sv_term(svi);
                d.srv = i; d.osv = d.sv = svi; d.olog = d.log = lg; d.pre = false;
                if (mterm > term) {  // UpdateTerm   raft.tla:434-440 (message not consumed)
                    d.sv = sv_set_voted(sv_set_state(sv_set_term(svi, mterm), R_FOLLOWER), 0);
                    if (mterm > prm.max_term) st |= ST_OUT_OF_MODEL;
                } else if (type == M_RVREQ) {  // HandleRequestVoteRequest   raft.tla:313-332
                    const int llt = (int)(m >> 13 & 7), lli = (int)(m >> 16 & 7), lt = rlog::last_term(lg, tb);
                    const bool logOk = llt > lt || (llt == lt && lli >= rlog::len(lg, tb));
                    const bool grant = mterm == term && logOk && (sv_voted(svi) == 0 || sv_voted(svi) == j + 1);
                    if (grant) d.sv = sv_set_voted(svi, j + 1);
                    want_send = true;  // Reply(response, m)
                    skey = mk_rvresp(term, grant ? 1 : 0, lg, i, j);
                    discard(k, m, d);
                } else if (type == M_RVRESP) {
                    if (mterm == term) {  // HandleRequestVoteResponse   raft.tla:336-349
                        if (m >> 13 & 1) {
                            const unsigned vg = sv_granted(svi);
                            d.sv = sv_set_granted(svi, vg | (1u << j));
                            if (!(vg >> j & 1)) {  // voterLog[i] @@ (j :> m.mlog): existing entry wins
                                d.vmode = 2; d.vj = j; d.vlog = rv_mlog(m); d.ovl = s.get(W_VL(i));
                            }
                        }
                    }  // else DropStaleResponse   raft.tla:443-446
                    discard(k, m, d);
                } else if (type == M_AEREQ) {  // HandleAppendEntriesRequest   raft.tla:355-417
                    const int pidx = ae_pidx(m), pterm = ae_pterm(m, tb), nent = ae_nent(m, tb);
                    const unsigned ent = ae_ent(m, tb);
                    const int mci = ae_cidx(m), stt = sv_state(svi), len = rlog::len(lg, tb);
                    const bool logOk = pidx == 0 || (pidx > 0 && pidx <= len && pterm == rlog::eterm(rlog::entry(lg, pidx, tb)));
                    if (mterm < term || (mterm == term && stt == R_FOLLOWER && !logOk)) {  // reject   :361-373
                        want_send = true;
                        skey = mk_aeresp(term, 0, 0, i, j);
                        discard(k, m, d);
                    } else if (stt == R_CANDIDATE) {  // return to follower state   :374-378 (mterm = term here)
                        d.sv = sv_set_state(svi, R_FOLLOWER);
                    } else if (stt == R_FOLLOWER && logOk) {  // accept request   :379-416
                        const int index = pidx + 1;
                        if (nent == 0 || (len >= index && rlog::eterm(rlog::entry(lg, index, tb)) == rlog::eterm(ent))) {
                            // already done with request   :384-402; commitIndex' assigned AND UNCHANGED
                            if (mci != sv_commit(svi)) return 0;
                            want_send = true;
                            skey = mk_aeresp(term, 1, pidx + nent, i, j);
                            discard(k, m, d);
                        } else if (len >= index) {  // conflict: remove 1 entry   :403-410
                            d.log = rlog::drop_last(lg, tb);
                        } else if (len == pidx) {  // no conflict: append entry   :411-416
                            if (!rlog::can_append(lg, ent, tb)) return ST_ENABLED | ST_OVERFLOW;
                            d.log = rlog::append(lg, ent, tb);
                            if (rlog::len(d.log, tb) > prm.max_log_len) st |= ST_OUT_OF_MODEL;
                        } else {
                            return 0;
                        }
                    } else {
                        return 0;  // e.g. a Leader receiving AppendEntries of its own term
                    }
                } else {  // AppendEntriesResponse
                    if (mterm == term) {  // HandleAppendEntriesResponse   raft.tla:421-431
                        if (m >> 13 & 1) {
                            const int mi = (int)(m >> 14 & 7);
                            d.sv = sv_set_match(sv_set_next(svi, j, mi + 1), j, mi);
                        } else {
                            const int nx = sv_next(svi, j) - 1;
                            d.sv = sv_set_next(svi, j, nx > 1 ? nx : 1);  // Max({nextIndex[i][j] - 1, 1})
                        }
                    }  // else DropStaleResponse
                    discard(k, m, d);
                }
            } else {
                return 0;
            }
        } else {
            return 0;
        }
        // ---- (4): Send(m) / Reply(response, m): one place, whatever the kind
        if (want_send) st |= send(l, s, prm, skey, d);
        // bookkeeping shared by every action: counts in the globals word
        if ((d.nmop & 1) && d.midxA >= l.nm) {
            d.glob += 1ull << 8;
            if (prm.max_keys && l.nm + 1 > prm.max_keys) st |= ST_OUT_OF_MODEL;  // StateConstraint: Cardinality(DOMAIN messages) <= MaxMsgKeys
        }
        if (l.nadd) {
            if (g_na(l.glob) + l.nadd > prm.ca) st |= ST_OVERFLOW;
            d.glob += (uint64_t)l.nadd << 24;
        }
        if (l.inflight + d.dinflight > prm.max_msgs) st |= ST_OUT_OF_MODEL;
        // invariants on the successor (the parent satisfies them, so only the changed server matters)
        if ((prm.inv_mask & 1) && d.srv >= 0 && sv_state(d.sv) == R_LEADER) {  // NoTwoLeaders   raft.tla:500-507
#pragma unroll
            for (int jj = 0; jj < NS; jj++)
                if (jj != d.srv && sv_state(srv_word<MEM>(l, s, jj)) == R_LEADER && sv_term(srv_word<MEM>(l, s, jj)) == sv_term(d.sv)) st |= ST_INVARIANT;
        }
        if ((prm.inv_mask & 2) && !(st & ST_INVARIANT) && g_decr(d.glob)) st |= ST_INVARIANT | (1u << 8);  // CommittedLogStable
        return st;
    }

    // GENERATED-ONLY successors (round 4): enabled, so TLC counts them, but never stored — decided from the parent's in-flight
    // count alone, without evaluating the action.  StateConstraint bounds the copies in flight by MaxMsgs (specs/MCraft.tla), every
    // stored parent satisfies it, and an action that only ADDS a copy to a parent that already holds MaxMsgs leaves the model:
    //   DuplicateMessage(m) of a message in flight once (raft.tla:471-473): always one copy more;
    //   RequestVote(i, j) of a candidate (raft.tla:209-217): Send adds a copy of its request — unless the key is saturated at two
    //   copies (raft.tla:117-121), which a stored parent can only hold when MaxMsgs >= 2: the shortcut is taken for MaxMsgs = 1.
    // Neither changes a server or a global, so no invariant can fire on the discarded successor.  73 % of the bench model's parents
    // hold a message in flight: three quarters of the RequestVote pairs and every DuplicateMessage were evaluated for nothing
    // (profiles/r04m_phase_profile_t3.json: RequestVote batches 12 %, inline message actions 8 % of a wavefront's time).
    // tests/_shim checks on every (state, slot) of the CPU lowering tests that a slot named here IS enabled and NOT storable.
    static constexpr bool GENERATED_ONLY = true;
    MC_HD static bool fixed_generated_only(const Params &prm, int inflight, int slot) {  // for a slot whose guard bit is set
        return prm.max_msgs == 1 && inflight >= 1 && slot >= 2 * NS && slot < 2 * NS + NS * NS;
    }
    template <class Ref>
    MC_HD static bool message_generated_only(const Params &prm, const Local &l, Ref s, int slot) {  // slot >= FIX
        const int q = slot - FIX, k = q / 3;
        return q % 3 == 1 && l.inflight >= prm.max_msgs && k < l.nm && m_count(rd_msg(s, k)) == 1;
    }

    // ---------------------------------------------------------------- expand-by-family interface
    // guards: cheap and EXACT as to the family (compute<MEM, FAM> still decides whether the action is enabled)
    struct Guards {
        uint64_t fixed, fixed_hi;  // bit s: fixed slot s (< FIX) may be enabled.  FIX = 5 n + 2 n^2 is 33 for three servers and 75 for
                                   // five: two words (round 2 kept one — AppendEntries(i, j) of a leader s4 / s5, slots 65 .. 74, was
                                   // never queued by the by-family kernel; found by the 15-level golden of the 5-server model)
        uint32_t infl;    // load_expand: bit k — message slot k < GUARD_SLOTS holds a message with a copy in flight (count > 0)
    };
    static constexpr int GUARD_SLOTS = 16;
    MC_HD static unsigned inflight_slots(const Guards &g) { return g.infl; }
    MC_HD static void fixed_set(Guards &g, int slot) { if (slot < 64) g.fixed |= 1ull << slot; else g.fixed_hi |= 1ull << (slot - 64); }
    MC_HD static bool fixed_bit(const Guards &g, int slot) { return ((slot < 64 ? g.fixed >> slot : g.fixed_hi >> (slot - 64)) & 1ull) != 0; }
    MC_HD static void fixed_clear_dense(Guards &g) { g.fixed &= ~((1ull << DENSE_SLOTS) - 1ull); }
    static_assert(FIX <= 128 && DENSE_SLOTS < 64, "Guards holds 128 fixed slots");
    MC_HD static int fixed_family(int slot) {
        // (slots < 2 NS are the dense pairs: never asked)
        return slot < 2 * NS + NS * NS ? F_REQVOTE : slot < 5 * NS + NS * NS ? F_MISC /* BecomeLeader, ClientRequest, AdvanceCommitIndex */ : F_APPEND;
    }
    MC_HD static void guards(const Params &prm, const Local &l, Guards &g) {
        g.fixed = g.fixed_hi = 0;
        g.infl = 0;
        const int creq = g_creq(l.glob);
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const uint64_t sv = l.sv.get(i);
            const int st = sv_state(sv);
            fixed_set(g, i);                                                               // Restart(i)
            if (st == R_FOLLOWER || st == R_CANDIDATE) fixed_set(g, NS + i);              // Timeout(i)
            if (st == R_CANDIDATE) {
#pragma unroll
                for (int j = 0; j < NS; j++) fixed_set(g, 2 * NS + i * NS + j);           // RequestVote(i, j)
                if (in_quorum(sv_granted(sv))) fixed_set(g, 2 * NS + NS * NS + i);        // BecomeLeader(i)
            }
            if (st == R_LEADER) {
                if (creq < prm.max_client_requests) fixed_set(g, 3 * NS + NS * NS + i);   // ClientRequest(i)
                fixed_set(g, 4 * NS + NS * NS + i);                                       // AdvanceCommitIndex(i)
#pragma unroll
                for (int j = 0; j < NS; j++) if (j != i) fixed_set(g, 5 * NS + NS * NS + i * NS + j);  // AppendEntries(i, j)
            }
        }
    }
    // Phase A of the by-family expand kernel (lane = parent, rows of the arena block coalesced): load() and guards() in one, with
    // the loads of the parent's row issued in TWO groups — header (fingerprint, globals, two words per server), then up to 12
    // message slots + the allLogs words: two trips to HBM instead of eight dependent ones — and with what the kernel's push loop
    // needs of the bag — which messages have a copy in flight: their Receive / Duplicate / Drop are evaluated inline — kept as
    // one bit per slot.  Same Local / Guards as load() + guards().
    template <class Ref>
    MC_HD static void load_expand(const Params &prm, Ref s, Local &l, Guards &g) {
        // ---- trip 1: header (2 + 2 NS loads in flight)
        l.fp = s.get(W_FP);
        const uint64_t gw = s.get(W_GLOB);
        l.glob = gw & 0xffffffffull;
        l.clog = gw >> 32;
        uint64_t vw[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const uint64_t x = s.get(W_SRV(i));
            l.sv.set(i, x & SVMASK);
            l.log.set(i, (uint32_t)(x >> LOGSH));
            vw[i] = s.get(W_VL(i));
        }
        l.vany = 0;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            uint64_t h = 0;
            if (vw[i]) l.vany |= 1u << i;
#pragma unroll
            for (int j = 0; j < NS; j++) h += hvlog(vl_get(vw[i], j, prm), i, j);
            l.vlh.set(i, h);
        }
        l.nm = g_nm(l.glob);
        guards(prm, l, g);
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);  // the loads of trip 2 stay below
#endif
        // ---- trip 2: up to PRE message slots and the first allLogs word (slots up to the capacity are part of the row whatever
        //      nMsgs says; an index is clamped into the capacity, wave-uniformly)
        constexpr int PRE = 12;
        uint64_t mw[PRE];
        {
            const int mwords = msg_words(prm);
#pragma unroll
            for (int q = 0; q < PRE / 2; q++) {  // (an index past the capacity is clamped into it, wave-uniformly)
                const uint64_t x = s.get(W_MSG0 + (q < mwords ? q : mwords - 1));
                mw[2 * q] = x & 0xffffffffull;
                mw[2 * q + 1] = x >> 32;
            }
        }
        const int wall = W_ALL0(prm);
        const uint64_t al0 = s.get(wall);
        // ---- arithmetic (bags beyond PRE keys / sets beyond 4 logs: the tail loads)
        l.inflight = 0;
        uint32_t sw[4] = {SIG_EMPTY, SIG_EMPTY, SIG_EMPTY, SIG_EMPTY};
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < l.nm) {
                const uint64_t x = k < PRE ? mw[k < PRE ? k : 0] : rd_msg(s, k);
                l.inflight += m_count(x);
                sw[k >> 2] = (sw[k >> 2] & ~(0xffu << (8 * (k & 3)))) | (key_sig(x) << (8 * (k & 3)));
                if (m_count(x) > 0) g.infl |= 1u << k;
            }
        }
        l.sig = Sigs{sw[0], sw[1], sw[2], sw[3]};
        for (int k = 16; k < l.nm; k++) l.inflight += m_count(rd_msg(s, k));
        unsigned present = 0;
        const int na = g_na(l.glob);
#pragma unroll
        for (int a = 0; a < 4; a++)
            if (a < na) {
                const uint64_t x = (al0 >> (16 * a)) & 0xffffull;
#pragma unroll
                for (int i = 0; i < NS; i++) if (x == l.log.get(i)) present |= 1u << i;
            }
        for (int a = 4; a < na; a++) {
            const uint64_t x = rd_all(prm, s, a);
#pragma unroll
            for (int i = 0; i < NS; i++) if (x == l.log.get(i)) present |= 1u << i;
        }
        finish_local<true>(prm, l, present);
    }
    // what a lane evaluating a pair needs to know about the pair's parent beyond the words it reads itself (computed once by the
    // parent's lane, kept in LDS: 40 B per parent): the fingerprint every successor starts from, the globals (the low 32 bits of
    // their word), the key signatures and the allLogs' bookkeeping.  committedLog (the high half of that word) is read from the
    // arena by the one kind that needs it (AdvanceCommitIndex).
    struct Summary {
        uint64_t base_fp;  // fp + add_fp
        Sigs sig;
        uint32_t packed;   // nm[0,8) inflight[8,16) nadd[16,20) addmask[20,28)
        uint32_t glob;
        uint64_t dig;      // Local::dig
    };
    // bit 31 of `packed` is free: the by-family kernel marks there a parent that got a successor from another lane's family batch
    MC_HD static uint32_t *succ_word(Summary &q) { return &q.packed; }
    MC_HD static void summarize(const Local &l, Summary &q) {
        q.base_fp = l.fp + l.add_fp; q.sig = l.sig; q.glob = (uint32_t)l.glob; q.dig = l.dig;
        q.packed = (uint32_t)l.nm | (uint32_t)l.inflight << 8 | (uint32_t)l.nadd << 16 | l.addmask << 20;
    }
    // FAM = the queue the pair was taken from: only kinds that read or write committedLog load it
    template <int FAM, class Ref>
    MC_HD static void local_of_summary(const Summary &q, Ref s, Local &l) {
        l.fp = q.base_fp; l.add_fp = 0; l.glob = q.glob; l.sig = q.sig; l.dig = q.dig;
        l.clog = (FAM < 0 || FAM == F_MISC) ? rd_clog(s) : 0ull;  // other kinds copy it: d.clog = l.clog, never hashed
        l.nm = (int)(q.packed & 255u); l.inflight = (int)(q.packed >> 8 & 255u); l.nadd = (int)(q.packed >> 16 & 15u);
        l.addmask = q.packed >> 20 & 255u;
        l.cache_k = -1; l.cache_hm = 0;
    }
    // evaluate one (parent, slot) pair of family FAM; the parent's words are read through `s`
    template <int FAM, class Ref>
    MC_HD static unsigned eval_pair(const Params &prm, const Summary &q, Ref s, int slot, uint64_t &fp) {
        Local l;
        local_of_summary<FAM>(q, s, l);
        Delta d;
        int action;
        const unsigned st = compute<true, FAM>(prm, l, s, slot, d, action);
        if (!(st & ST_ENABLED)) return 0;
        if (st & (ST_OUT_OF_MODEL | ST_OVERFLOW | ST_SPECERR)) { fp = 1; return st; }
        if (is_self_loop(l, s, d)) { fp = 1; return st | ST_SELFLOOP; }  // (the successor is the parent: no fingerprint needed)
        fp = fp_nonzero(delta_fp(prm, l, s, d));
        return st;
    }

    template <class Ref>
    MC_HD static uint64_t delta_fp(const Params &prm, Local &l, Ref s, const Delta &d) {
        (void)s;
        const int lbits = prm.lb;
        uint64_t fp = l.fp + l.add_fp;
        if ((d.glob ^ l.glob) & GLOB_HASHED) fp += hfield(F_GLOB, d.glob & GLOB_HASHED) - hfield(F_GLOB, l.glob & GLOB_HASHED);
        if (d.clog != l.clog) fp += hfield(F_CLOG, d.clog) - hfield(F_CLOG, l.clog);
        if (d.srv >= 0) {
            const int i = d.srv;
            if (d.sv != d.osv) fp += hfield(F_SV(i), d.sv) - hfield(F_SV(i), d.osv);
            if (d.log != d.olog) fp += hfield(F_LOG(i), d.log) - hfield(F_LOG(i), d.olog);
            if (d.vmode == 1) {
                if (d.ovl) {
#pragma unroll
                    for (int j = 0; j < NS; j++) fp -= hvlog((d.ovl >> (lbits * j)) & ((1ull << lbits) - 1ull), i, j);
                }
            } else if (d.vmode == 2) {
                fp += hvlog(d.vlog, i, d.vj);  // the entry was empty (compute: "existing entry wins")
            }
        }
        if (d.nmop & 1) {  // Send: a new key enters with count 1 (multiplicity 2), a known one gains a copy unless saturated
            if (d.midxA >= l.nm) fp += 2 * hkey(d.mnewA);
            else if (d.mnewA != d.moldA) fp += hkey(d.mnewA);
        }
        if ((d.nmop & 2) && d.mnewB != d.moldB) {  // the message the slot was read from: one copy less (Discard / Drop) or more (Duplicate)
            if (l.cache_k != d.midxB) { l.cache_k = d.midxB; l.cache_hm = hkey(d.moldB); }
            fp += d.mnewB > d.moldB ? l.cache_hm : 0ull - l.cache_hm;
        }
        if (d.eadd) fp += helec(d.ew);
        return fp;
    }

    // the successor equals its parent (Restart of a freshly restarted server, AdvanceCommitIndex without progress, ...: about
    // one generated successor in nine): no fingerprint, no probe — the parent is in the seen-set
    template <class Ref>
    MC_HD static bool is_self_loop(const Local &l, Ref s, const Delta &d) {
        if (l.nadd || d.nmop || d.eadd || d.glob != l.glob || d.clog != l.clog) return false;
        if (d.srv < 0) return true;
        if (d.sv != d.osv || d.log != d.olog) return false;
        if (d.vmode == 2) return d.vlog == 0;
        if (d.vmode == 1) return d.ovl == 0;
        return true;
    }

    template <class Ref>
    MC_HD static unsigned eval(const Params &prm, Local &l, Ref s, int slot, uint64_t &fp) {
        Delta d;
        int action;
        const unsigned st = compute(prm, l, s, slot, d, action);
        if (!(st & ST_ENABLED)) return 0;
        // a successor outside the CONSTRAINT is generated and invariant-checked (done in compute) but
        // never stored, so its fingerprint is not needed
        if (st & (ST_OUT_OF_MODEL | ST_OVERFLOW | ST_SPECERR)) { fp = 1; return st; }
        if (is_self_loop(l, s, d)) { fp = fp_nonzero(l.fp); return st | ST_SELFLOOP; }  // the parent's own fingerprint
        fp = fp_nonzero(delta_fp(prm, l, s, d));
        return st;
    }
    // Restart(i) and Timeout(i) of ONE server, evaluated together by the parent's own lane (the dense slots of k_expand_family:
    // half of all successors).  Both rewrite the scalars of server i and clear voterLog[i], so H(old scalars) and the voterLog
    // terms are computed once for the two, and nothing else of the state changes (raft.tla:186-194, 197-206).  The results are
    // exactly eval(slot = i) and eval(slot = NS + i) — tests/_shim compares them on every state of every lowering test.
    static constexpr int DENSE_PAIRS = NS;
    MC_HD static int dense_slot(int i, int second) { return second ? NS + i : i; }
    // (the server index is a template parameter: a run-time index into the register copy `l.sv` would be turned into an indexed
    //  load of a stack array by the compiler — scratch memory — however uniform it is)
    template <class Ref>
    MC_HD static void eval_dense(const Params &prm, const Local &l, Ref s, int i, unsigned &stR, uint64_t &fpR, unsigned &stT, uint64_t &fpT) {
        if (i == 0) eval_dense_i<0>(prm, l, s, stR, fpR, stT, fpT);
        if (NS > 1 && i == 1) eval_dense_i<1 % NS>(prm, l, s, stR, fpR, stT, fpT);
        if (NS > 2 && i == 2) eval_dense_i<2 % NS>(prm, l, s, stR, fpR, stT, fpT);
        if (NS > 3 && i == 3) eval_dense_i<3 % NS>(prm, l, s, stR, fpR, stT, fpT);
        if (NS > 4 && i == 4) eval_dense_i<4 % NS>(prm, l, s, stR, fpR, stT, fpT);
        static_assert(NS <= 5, "eval_dense dispatches over at most 5 servers");
    }
    template <int i, class Ref>
    MC_HD static void eval_dense_i(const Params &prm, const Local &l, Ref s, unsigned &stR, uint64_t &fpR, unsigned &stT, uint64_t &fpT) {
        const uint64_t osv = l.sv.get(i);
        // bookkeeping shared by every action (compute): allLogs' additions, the in-flight bound, CommittedLogStable
        unsigned stc = ST_ENABLED;
        uint64_t glob = l.glob;
        if (l.nadd) {
            if (g_na(l.glob) + l.nadd > prm.ca) stc |= ST_OVERFLOW;
            glob += (uint64_t)l.nadd << 24;
        }
        if (l.inflight > prm.max_msgs) stc |= ST_OUT_OF_MODEL;
        if ((prm.inv_mask & 2) && g_decr(glob)) stc |= ST_INVARIANT | (1u << 8);
        (void)s;
        const uint64_t salt = salt_of((unsigned)F_SV(i));
        const uint64_t base = l.fp + l.add_fp - l.vlh.get(i) - hmum(osv, salt);
        const bool still = !l.nadd && !(l.vany >> i & 1u);  // nothing but the scalars can differ from the parent
        // Restart(i): always enabled
        const uint64_t svR = sv_reset_leader_vars(sv_set_commit(sv_set_granted(sv_set_state(osv, R_FOLLOWER), 0), 0), 1);
        stR = stc;
        if (stc & (ST_OUT_OF_MODEL | ST_OVERFLOW)) fpR = 1;
        else if (still && svR == osv) { fpR = fp_nonzero(l.fp); stR |= ST_SELFLOOP; }
        else fpR = fp_nonzero(base + hmum(svR, salt));
        // Timeout(i): a follower or a candidate
        const int stt = sv_state(osv), nt = sv_term(osv) + 1;
        stT = 0;
        fpT = 0;
        if (stt == R_FOLLOWER || stt == R_CANDIDATE) {
            const uint64_t svT = sv_set_granted(sv_set_voted(sv_set_term(sv_set_state(osv, R_CANDIDATE), nt & 7), 0), 0);
            stT = stc | (nt > prm.max_term ? (unsigned)ST_OUT_OF_MODEL : 0u);
            if (stT & (ST_OUT_OF_MODEL | ST_OVERFLOW)) fpT = 1;
            else fpT = fp_nonzero(base + hmum(svT, salt));  // never the parent: the term grows
        }
    }

    // re-evaluate `slot` on parent `s` and write the whole successor to `out`
    template <class Ref>
    MC_HD static unsigned apply(const Params &prm, Ref s, int slot, WordRef out) { return apply_copy_patch<false>(prm, s, slot, 0, out); }
    // ... when the expand kernel hands the successor's fingerprint over (fp_nonzero of the raw sum): the dozen hash terms of
    // delta_fp are not computed a second time.  The one ambiguous value (raw sum 0 or the substitute itself) is recomputed.
    static constexpr bool KNOWN_FP = true;
    template <class Ref>
    MC_HD static unsigned apply_known_fp(const Params &prm, Ref s, int slot, uint64_t fp_nz, WordRef out) {
        if (fp_nz == fp_nonzero(0)) return apply_copy_patch<false>(prm, s, slot, 0, out);
        return apply_copy_patch<true>(prm, s, slot, fp_nz, out);
    }
    // k_materialise's writer: COPY the parent row while reading it once — every group of loads is followed by the stores of the
    // same words (the arena is both source and destination, so the compiler keeps every load behind every earlier store: a loop
    // of "read a word, write it" would be one memory round trip per word), and the per-parent cache (in-flight count, key
    // signatures, allLogs' additions) is computed from the registers the copy passes through — then evaluate the action and
    // PATCH the handful of words it changes.
    template <bool KNOWN, class Ref>
    MC_HD static unsigned apply_copy_patch(const Params &prm, Ref s, int slot, uint64_t fp_known, WordRef out) {
        prefetch_row(s, words(prm));
        Local l;
        l.fp = KNOWN ? 0ull : s.get(W_FP);
        const uint64_t gw = s.get(W_GLOB);
        l.glob = gw & 0xffffffffull;
        l.clog = gw >> 32;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const uint64_t x = s.get(W_SRV(i));
            l.sv.set(i, x & SVMASK);
            l.log.set(i, (uint32_t)(x >> LOGSH));
        }
        l.nm = g_nm(l.glob);
        l.inflight = 0;
        uint32_t sw[4] = {SIG_EMPTY, SIG_EMPTY, SIG_EMPTY, SIG_EMPTY};
        const int mwords = msg_words(prm);
        for (int k0 = 0; k0 < prm.cm; k0 += 8) {
            uint64_t pw[4], x[8];
#pragma unroll
            for (int q = 0; q < 4; q++) pw[q] = s.get(W_MSG0 + (k0 + 2 * q < l.nm ? (k0 >> 1) + q : 0));  // word 0 exists whatever nMsgs is
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = half_of(pw[u >> 1], u);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u;
                if (k < l.nm) {
                    l.inflight += m_count(x[u]);
                    if (k0 < SIG_SLOTS) {
                        const uint32_t sg = key_sig(x[u]) << (8 * (u & 3)), mk = ~(0xffu << (8 * (u & 3)));
                        const int wi = (k0 >> 2) + (u >> 2);  // k0 is 0 or 8 here
                        if (wi == 0) sw[0] = (sw[0] & mk) | sg;
                        if (wi == 1) sw[1] = (sw[1] & mk) | sg;
                        if (wi == 2) sw[2] = (sw[2] & mk) | sg;
                        if (wi == 3) sw[3] = (sw[3] & mk) | sg;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)   // (slots >= nMsgs of a used word are zero in the parent already)
                if ((k0 >> 1) + q < mwords) out.set(W_MSG0 + (k0 >> 1) + q, k0 + 2 * q < l.nm ? pw[q] : 0ull);
        }
        l.sig = Sigs{sw[0], sw[1], sw[2], sw[3]};
        const int ne = g_ne(l.glob), wel = W_EL0(prm);
        for (int e = 0; e < prm.ce; e++) {
            uint64_t x[EL_WORDS];
#pragma unroll
            for (int q = 0; q < EL_WORDS; q++) x[q] = s.get(wel + (e < ne ? e : 0) * EL_WORDS + q);
#pragma unroll
            for (int q = 0; q < EL_WORDS; q++) out.set(wel + e * EL_WORDS + q, e < ne ? x[q] : 0ull);
        }
        // allLogs: four 16-bit slots per word; the successor's additions (the same for every successor of this parent,
        // raft.tla:493) go to slots na, na + 1, ...  The words are NOT kept in registers across the action (16 words = 32
        // VGPRs of a writer that already wants more than the 128 it gets inside the expand kernel: round 4): read here for
        // the membership test, read again (the CU's L1) when they are written at the end.
        const int na = g_na(l.glob), wall = W_ALL0(prm), naw = all_words(prm);
        const unsigned present = all_present(prm, s, l, na);
        l.vany = 0;
        finish_local<!KNOWN>(prm, l, present);
        // ---- the action
        Delta d;
        int action;
        const unsigned st = compute<true>(prm, l, s, slot, d, action);
        if (!(st & ST_ENABLED) || (st & (ST_OVERFLOW | ST_SPECERR))) {  // not a successor: the parent itself (never the case in k_materialise)
            out.set(W_FP, s.get(W_FP));
            out.set(W_GLOB, gw);
#pragma unroll
            for (int i = 0; i < NS; i++) { out.set(W_SRV(i), s.get(W_SRV(i))); out.set(W_VL(i), s.get(W_VL(i))); }
            for (int q = 0; q < naw; q++) out.set(wall + q, q * 4 < na ? s.get(wall + q) : 0ull);
            return st;
        }
        // ---- patch
        out.set(W_FP, KNOWN ? fp_known : delta_fp(prm, l, s, d));
        out.set(W_GLOB, pack_glob(d.glob, d.clog));
        {   // the servers' words: re-read from the parent row (the CU's L1: the row was asked for at the top) instead of being kept
            // in registers across the action — the writer has 120 VGPRs inside the expand kernel
            uint64_t sw2[NS], vw2[NS];
#pragma unroll
            for (int i = 0; i < NS; i++) { sw2[i] = s.get(W_SRV(i)); vw2[i] = s.get(W_VL(i)); }
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const bool me = i == d.srv;
                out.set(W_SRV(i), me ? pack_srv(d.sv, d.log) : sw2[i]);
                uint64_t v = vw2[i];
                if (me && d.vmode == 1) v = 0;
                if (me && d.vmode == 2) v = vl_set(v, d.vj, d.vlog, prm);
                out.set(W_VL(i), v);
            }
        }
        {   // the one or two message slots the action rewrites: read-modify-write of their (parent) words; both may share a word
            const int wa = d.midxA >> 1, wb = d.midxB >> 1;
            if (d.nmop & 1) {
                uint64_t x = 2 * wa < l.nm ? s.get(W_MSG0 + wa) : 0ull;
                x = set_half(x, d.midxA, d.mnewA);
                if ((d.nmop & 2) && wb == wa) x = set_half(x, d.midxB, d.mnewB);
                out.set(W_MSG0 + wa, x);
            }
            if ((d.nmop & 2) && !((d.nmop & 1) && wb == wa)) out.set(W_MSG0 + wb, set_half(s.get(W_MSG0 + wb), d.midxB, d.mnewB));
        }
        if (d.eadd) {
#pragma unroll
            for (int q = 0; q < EL_WORDS; q++) out.set(wel + ne * EL_WORDS + q, d.ew.get(q));
        }
        for (int q = 0; q < naw; q++) {  // one word at a time: the parent's slots of the word + the additions that fall into it
            uint64_t x = q * 4 < na ? s.get(wall + q) : 0ull;
            int pos = na;
#pragma unroll
            for (int i = 0; i < NS; i++)
                if (l.addmask >> i & 1) {
                    if (pos < prm.ca && (pos >> 2) == q) x |= (uint64_t)l.log.get(i) << (16 * (pos & 3));
                    pos++;
                }
            out.set(wall + q, x);
        }
        return st;
    }

    // The in-wave writer (round 4: the expand wavefront that found a new state writes it; engine.hip wave_write_survivors).  The
    // parent was expanded by this very workgroup: what its lane derived from the row — in-flight count, key signatures, allLogs'
    // additions — still lies in LDS (Summary), so the writer does NOT walk the message bag and allLogs a second time as
    // apply_copy_patch does for k_materialise.  It (1) copies the row verbatim, all words asked for before the first is used
    // (one round trip to L2 / the Infinity Cache), (2) evaluates the action, (3) overwrites the handful of words it changes;
    // stores of one wavefront to one address stay in order.  About half the instructions of apply_copy_patch.
    static constexpr bool SUMMARY_WRITER = true;
#if defined(MC_WRITER_TWICE) && MC_WRITER_TWICE   // A/B: round 4's form — copy the row, then overwrite the changed words
    template <class Ref>
    MC_HD static unsigned apply_summary_patch(const Params &prm, const Summary &q, Ref s, int slot, uint64_t fp_nz, WordRef out) {
        const int W = words(prm);
        static_assert(W_FP == 0 && W_GLOB == 1, "the two words every successor rewrites are the row's first two");
        {
            uint64_t t[16];
#pragma unroll
            for (int w0 = 0; w0 < MAX_WORDS; w0 += 16) {
                if (w0 < W) {
#pragma unroll
                    for (int u = 0; u < 16; u++) t[u] = s.get(w0 + u < W ? w0 + u : W - 1);
                    // (words 0 and 1 — fingerprint and globals — are rewritten by EVERY successor: not copied first, 2 of 17 stores less)
#pragma unroll
                    for (int u = 0; u < 16; u++) if (w0 + u < W && w0 + u > W_GLOB) out.set(w0 + u, t[u]);
                }
            }
        }
        Local l;
        local_of_summary<-1>(q, s, l);
        Delta d;
        int action;
        const unsigned st = compute<true>(prm, l, s, slot, d, action);
        if (!(st & ST_ENABLED) || (st & (ST_OVERFLOW | ST_SPECERR))) {  // not a successor: the parent itself (never the case for a survivor)
            out.set(W_FP, s.get(W_FP));
            out.set(W_GLOB, s.get(W_GLOB));
            return st;
        }
        // the fingerprint arrives with the survivor; the one ambiguous value (raw sum 0 or the substitute itself) is recomputed
        out.set(W_FP, fp_nz != fp_nonzero(0) ? fp_nz : delta_fp(prm, l, s, d));
        out.set(W_GLOB, pack_glob(d.glob, d.clog));
        if (d.srv >= 0) {
            out.set(W_SRV(d.srv), pack_srv(d.sv, d.log));
            if (d.vmode == 1) out.set(W_VL(d.srv), 0ull);
            else if (d.vmode == 2) out.set(W_VL(d.srv), vl_set(s.get(W_VL(d.srv)), d.vj, d.vlog, prm));
        }
        {   // the one or two message slots the action rewrites (both may share a word)
            const int wa = d.midxA >> 1, wb = d.midxB >> 1;
            if (d.nmop & 1) {
                uint64_t x = 2 * wa < l.nm ? s.get(W_MSG0 + wa) : 0ull;
                x = set_half(x, d.midxA, d.mnewA);
                if ((d.nmop & 2) && wb == wa) x = set_half(x, d.midxB, d.mnewB);
                out.set(W_MSG0 + wa, x);
            }
            if ((d.nmop & 2) && !((d.nmop & 1) && wb == wa)) out.set(W_MSG0 + wb, set_half(s.get(W_MSG0 + wb), d.midxB, d.mnewB));
        }
        if (d.eadd) {
            const int ne = g_ne(l.glob), wel = W_EL0(prm);
#pragma unroll
            for (int e = 0; e < EL_WORDS; e++) out.set(wel + ne * EL_WORDS + e, d.ew.get(e));
        }
        if (l.nadd) {  // allLogs' additions (the same for every successor of this parent) land in slots na, na + 1, ...
            const int na = g_na(l.glob), wall = W_ALL0(prm);
            const int qlo = na >> 2, qhi = (na + l.nadd - 1) >> 2;
            for (int qq = qlo; qq <= qhi && qq < all_words(prm); qq++) {
                uint64_t x = qq * 4 < na ? s.get(wall + qq) : 0ull;
                int pos = na;
#pragma unroll
                for (int i = 0; i < NS; i++)
                    if (l.addmask >> i & 1) {
                        if (pos < prm.ca && (pos >> 2) == qq) x |= (uint64_t)rd_log(s, i) << (16 * (pos & 3));
                        pos++;
                    }
                out.set(wall + qq, x);
            }
        }
        return st;
    }

#else
    // Round 5: EVERY WORD IS STORED ONCE.  Round 4's writer copied the row and then overwrote the words the action changes — stores whose
    // word index differs from lane to lane (which server, which message slot), i.e. 8 bytes here and 8 bytes there into lines the copy had
    // just written whole: the L2 handed 2.05 G write requests per step of the contract workload to the memory for 1.05 G sectors of rows
    // (profiles/r05g_request_mix.json).  Now the action is evaluated first, its handful of (word, value) changes stay in registers, and
    // the row is copied THROUGH them: a word is loaded, replaced if the successor changes it — the compare knows its class from the
    // (compile-time) word index: fingerprint, globals, server i, message word, election word, allLogs word — and stored, 64 lanes =
    // 512 contiguous bytes, once.
    template <class Ref>
    MC_HD static unsigned apply_summary_patch(const Params &prm, const Summary &q, Ref s, int slot, uint64_t fp_nz, WordRef out) {
        const int W = words(prm);
        static_assert(W_FP == 0 && W_GLOB == 1 && W_MSG0 == 2 + 2 * NS, "word classes by index: fingerprint, globals, NS x (server, voterLog), messages, elections, allLogs");
        Local l;
        local_of_summary<-1>(q, s, l);
        Delta d;
        int action;
        const unsigned st = compute<true>(prm, l, s, slot, d, action);
        const bool ok = (st & ST_ENABLED) && !(st & (ST_OVERFLOW | ST_SPECERR));  // (false: the parent itself is written — never the case for a survivor)
        // the fingerprint arrives with the survivor; the one ambiguous value (raw sum 0 or the substitute itself) is recomputed
        const uint64_t nfp = !ok ? 0ull : fp_nz != fp_nonzero(0) ? fp_nz : delta_fp(prm, l, s, d);
        const uint64_t nglob = ok ? pack_glob(d.glob, d.clog) : 0ull;
        const int srv = ok ? d.srv : -1;
        const uint64_t nsrv = srv >= 0 ? pack_srv(d.sv, d.log) : 0ull;
        const bool vch = srv >= 0 && d.vmode != 0;
        const uint64_t nvl = !vch || d.vmode == 1 ? 0ull : vl_set(s.get(W_VL(srv)), d.vj, d.vlog, prm);
        // the one or two message slots the action rewrites (both may share a word)
        const int wa = d.midxA >> 1, wb = d.midxB >> 1;
        bool hasA = false, hasB = false;
        uint64_t xa = 0, xb = 0;
        if (ok && (d.nmop & 1)) {
            xa = 2 * wa < l.nm ? s.get(W_MSG0 + wa) : 0ull;
            xa = set_half(xa, d.midxA, d.mnewA);
            if ((d.nmop & 2) && wb == wa) xa = set_half(xa, d.midxB, d.mnewB);
            hasA = true;
        }
        if (ok && (d.nmop & 2) && !((d.nmop & 1) && wb == wa)) { xb = set_half(s.get(W_MSG0 + wb), d.midxB, d.mnewB); hasB = true; }
        const int nmw = msg_words(prm), wel = W_MSG0 + nmw, wall = wel + prm.ce * EL_WORDS;
        const int e0 = ok && d.eadd ? g_ne(l.glob) * EL_WORDS : -1;   // election words e0, e0 + 1 (relative to wel)
        // allLogs' additions (the same for every successor of this parent) land in slots na, na + 1, ...: at most NS of them, 4 per word
        constexpr int ALLW = (NS + 3) / 4 + 1;
        int qlo = 0, qhi = -1;
        uint64_t xall[ALLW];
#pragma unroll
        for (int j = 0; j < ALLW; j++) xall[j] = 0;
        if (ok && l.nadd) {
            const int na = g_na(l.glob);
            qlo = na >> 2;
            qhi = (na + l.nadd - 1) >> 2;
            if (qhi >= all_words(prm)) qhi = all_words(prm) - 1;
#pragma unroll
            for (int j = 0; j < ALLW; j++) {
                const int qq = qlo + j;
                if (qq <= qhi) {
                    uint64_t x = qq * 4 < na ? s.get(wall + qq) : 0ull;
                    int pos = na;
#pragma unroll
                    for (int i = 0; i < NS; i++)
                        if (l.addmask >> i & 1) {
                            if (pos < prm.ca && (pos >> 2) == qq) x |= (uint64_t)rd_log(s, i) << (16 * (pos & 3));
                            pos++;
                        }
                    xall[j] = x;
                }
            }
        }
        uint64_t t[16];
#pragma unroll
        for (int w0 = 0; w0 < MAX_WORDS; w0 += 16) {
            if (w0 < W) {
#pragma unroll
                for (int u = 0; u < 16; u++) t[u] = s.get(w0 + u < W ? w0 + u : W - 1);
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int w = w0 + u;   // (a compile-time constant after unrolling)
                    if (w >= W) continue;
                    uint64_t v = t[u];
                    if (w == W_FP) { if (ok) v = nfp; }
                    else if (w == W_GLOB) { if (ok) v = nglob; }
                    else if (w < W_MSG0) {
                        const int i = (w - 2) >> 1;
                        if (((w - 2) & 1) == 0) { if (srv == i) v = nsrv; }
                        else if (vch && srv == i) v = nvl;
                    } else {
                        const int mw = w - W_MSG0;
                        if (mw < nmw) {
                            if (hasA && wa == mw) v = xa;
                            if (hasB && wb == mw) v = xb;
                        } else if (w < wall) {
                            const int e = w - wel;
#pragma unroll
                            for (int k = 0; k < EL_WORDS; k++) if (e == e0 + k && e0 >= 0) v = d.ew.get(k);
                        } else {
                            const int qq = w - wall;
#pragma unroll
                            for (int j = 0; j < ALLW; j++) if (qq == qlo + j && qq <= qhi) v = xall[j];
                        }
                    }
                    out.set(w, v);
                }
            }
        }
        return st;
    }
#endif

    // ---------------------------------------------------------------- host side: names and text
    static int action_of(const Params &prm, const uint64_t *parent, int slot) {
        Local l;
        CWordRef s{parent, 1};
        load(prm, s, l);
        Delta d;
        int action = -1;
        compute(prm, l, s, slot, d, action);
        return action;
    }
    static const char *action_name(int a) {
        static const char *nm[] = {"Restart", "Timeout", "RequestVote", "BecomeLeader", "ClientRequest",
                                   "AdvanceCommitIndex", "AppendEntries", "Receive", "DuplicateMessage", "DropMessage"};
        return a >= 0 && a < 10 ? nm[a] : a < 0 ? "Initial predicate" : "?";
    }

    struct Txt {
        char *b; size_t cap, k;
        void put(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
            va_list ap;
            va_start(ap, fmt);
            if (k < cap) { int w = vsnprintf(b + k, cap - k, fmt, ap); if (w > 0) k += (size_t)w; if (k > cap) k = cap; }
            va_end(ap);
        }
    };
    static void t_log(Txt &o, uint64_t lg, int tb) {
        const int n = rlog::len(lg, tb);
        if (!n) { o.put("<<>>"); return; }
        o.put("<<");
        for (int k = 1; k <= n; k++) { const unsigned e = rlog::entry(lg, k, tb); o.put("%s[term |-> %d, value |-> %d]", k > 1 ? ", " : "", rlog::eterm(e), rlog::evalue(e)); }
        o.put(">>");
    }
    static void t_servers(Txt &o, unsigned mask) {
        o.put("{");
        bool first = true;
        for (int j = 0; j < NS; j++) if (mask >> j & 1) { o.put("%ss%d", first ? "" : ", ", j + 1); first = false; }
        o.put("}");
    }
    static void t_vlog(Txt &o, unsigned dom, uint64_t packed, const Params &prm) {
        if (!dom) { o.put("<<>>"); return; }
        o.put("(");
        bool first = true;
        for (int j = 0; j < NS; j++) if (dom >> j & 1) { o.put("%ss%d :> ", first ? "" : " @@ ", j + 1); t_log(o, vl_get(packed, j, prm), prm.tb); first = false; }
        o.put(")");
    }
    static void t_msg(Txt &o, uint64_t m, int tb) {
        const int d = m_dst(m) + 1, sr = m_src(m) + 1, t = m_term(m);
        switch (m_type(m)) {
        case M_RVREQ:
            o.put("[mdest |-> s%d, mlastLogIndex |-> %d, mlastLogTerm |-> %d, msource |-> s%d, mterm |-> %d, mtype |-> RequestVoteRequest]",
                  d, (int)(m >> 16 & 7), (int)(m >> 13 & 7), sr, t);
            break;
        case M_RVRESP:
            o.put("[mdest |-> s%d, mlog |-> ", d); t_log(o, rv_mlog(m), tb);
            o.put(", msource |-> s%d, mterm |-> %d, mtype |-> RequestVoteResponse, mvoteGranted |-> %s]", sr, t, (m >> 13 & 1) ? "TRUE" : "FALSE");
            break;
        case M_AEREQ: {
            o.put("[mcommitIndex |-> %d, mdest |-> s%d, mentries |-> ", ae_cidx(m), d);
            const unsigned e = ae_ent(m, tb);
            if (ae_nent(m, tb)) o.put("<<[term |-> %d, value |-> %d]>>", rlog::eterm(e), rlog::evalue(e)); else o.put("<<>>");
            o.put(", mlog |-> "); t_log(o, ae_mlog(m), tb);
            o.put(", mprevLogIndex |-> %d, mprevLogTerm |-> %d, msource |-> s%d, mterm |-> %d, mtype |-> AppendEntriesRequest]",
                  ae_pidx(m), ae_pterm(m, tb), sr, t);
            break;
        }
        default:
            o.put("[mdest |-> s%d, mmatchIndex |-> %d, msource |-> s%d, msuccess |-> %s, mterm |-> %d, mtype |-> AppendEntriesResponse]",
                  d, (int)(m >> 14 & 7), sr, (m >> 13 & 1) ? "TRUE" : "FALSE", t);
        }
    }
    static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
    static void t_sorted(Txt &o, char **it, int n, const char *open, const char *sep, const char *close, const char *empty) {
        if (!n) { o.put("%s", empty); return; }
        qsort(it, (size_t)n, sizeof *it, cmp_str);
        o.put("%s", open);
        for (int i = 0; i < n; i++) { o.put("%s%s", i ? sep : "", it[i]); free(it[i]); }
        o.put("%s", close);
    }
    // canonical TLA+ text (same format as oracle/spec_raft.c:raft_print; sets sorted by text)
    static int format(const Params &prm, const uint64_t *w, char *buf, size_t cap) {
        static const char *stn[] = {"Follower", "Candidate", "Leader", "?"};
        Txt o{buf, cap, 0};
        char tmp[2048];
        char *it[64];
        const uint64_t g = w[W_GLOB] & 0xffffffffull;
        const int tb = prm.tb;
        o.put("/\\ messages = ");
        for (int k = 0; k < g_nm(g); k++) { const uint64_t mk_ = rd_msg(CWordRef{w, 1}, k); Txt e{tmp, sizeof tmp, 0}; t_msg(e, mk_, tb); e.put(" :> %d", m_count(mk_)); it[k] = strndup(tmp, e.k); }
        t_sorted(o, it, g_nm(g), "(", " @@ ", ")", "<<>>");
        o.put("\n/\\ elections = ");
        for (int x = 0; x < g_ne(g); x++) {
            Txt e{tmp, sizeof tmp, 0};
            const uint64_t *ew = w + W_EL0(prm) + x * EL_WORDS;
            const unsigned votes = (unsigned)(ew[0] >> 6 & 31);
            e.put("[eleader |-> s%d, elog |-> ", (int)(ew[0] >> 3 & 7) + 1); t_log(e, (ew[0] >> 11) & 0x7fffull, tb);
            e.put(", eterm |-> %d, evoterLog |-> ", (int)(ew[0] & 7)); t_vlog(e, votes, ew[1], prm);
            e.put(", evotes |-> "); t_servers(e, votes); e.put("]");
            it[x] = strndup(tmp, e.k);
        }
        t_sorted(o, it, g_ne(g), "{", ", ", "}", "{}");
        o.put("\n/\\ allLogs = ");
        for (int a = 0; a < g_na(g); a++) { Txt e{tmp, sizeof tmp, 0}; t_log(e, rd_all(prm, CWordRef{w, 1}, a), tb); it[a] = strndup(tmp, e.k); }
        t_sorted(o, it, g_na(g), "{", ", ", "}", "{}");
#define MC_PER_SERVER(title, expr)                                                             \
    o.put("\n/\\ " title " = (");                                                              \
    for (int i = 0; i < NS; i++) { const uint64_t sv = w[W_SRV(i)] & SVMASK; (void)sv; o.put("%ss%d :> ", i ? " @@ " : "", i + 1); expr; } \
    o.put(")");
        MC_PER_SERVER("currentTerm", o.put("%d", sv_term(sv)));
        MC_PER_SERVER("state", o.put("%s", stn[sv_state(sv)]));
        MC_PER_SERVER("votedFor", if (sv_voted(sv)) o.put("s%d", sv_voted(sv)); else o.put("Nil"));
        o.put("\n/\\ clientRequests = %d", g_creq(g));
        MC_PER_SERVER("log", t_log(o, w[W_SRV(i)] >> LOGSH, tb));
        MC_PER_SERVER("commitIndex", o.put("%d", sv_commit(sv)));
        o.put("\n/\\ committedLog = "); t_log(o, w[W_GLOB] >> 32, tb);
        o.put("\n/\\ committedLogDecrease = %s", g_decr(g) ? "TRUE" : "FALSE");
        MC_PER_SERVER("votesSent", o.put("FALSE"));
        MC_PER_SERVER("votesGranted", t_servers(o, sv_granted(sv)));
        MC_PER_SERVER("voterLog", t_vlog(o, sv_granted(sv), w[W_VL(i)], prm));
        MC_PER_SERVER("nextIndex", { o.put("("); for (int j = 0; j < NS; j++) o.put("%ss%d :> %d", j ? " @@ " : "", j + 1, sv_next(sv, j)); o.put(")"); });
        MC_PER_SERVER("matchIndex", { o.put("("); for (int j = 0; j < NS; j++) o.put("%ss%d :> %d", j ? " @@ " : "", j + 1, sv_match(sv, j)); o.put(")"); });
#undef MC_PER_SERVER
        return (int)o.k;
    }
};

}  // namespace mc
