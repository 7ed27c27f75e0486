// spec_paxos.h — device lowering of the reference's Paxos family:
//
//   examples/Paxos/Voting.tla:133-156  IncreaseMaxBal, VoteFor, Next        (model: MCVoting.tla + MCVoting.cfg)
//   examples/Paxos/Paxos.tla:93-176    Phase1a, Phase1b, Phase2a, Phase2b   (model: MCPaxos.tla + MCPaxos.cfg)
//
// with what the two cfg files ask TLC for: INVARIANT Inv (Voting.tla:160) / Inv1..Inv4 (= Inv!1..Inv!4 of Paxos.tla:192-208),
// PROPERTY C!Spec / V!Spec (the refinement step [C!Next]_chosen, [V!Next]_<<votes, maxBal>> checked on every generated
// transition), SYMMETRY Permutations(MCAcceptor) \cup Permutations(MCValue) (MCVoting.tla:10, MCPaxos.tla:12).
//
// Packed state: ONE 64-bit block per acceptor plus one word of acceptor-independent messages — the layout is chosen for the
// SYMMETRY reduction, which is then exact and cheap:
//   word 0      (Paxos)  bit b: [type |-> "1a", bal |-> b] \in msgs        bit nb + b*nv + v: [type |-> "2a", bal |-> b, val |-> v]
//   word 1 + a  (Paxos)  maxBal[a]+1 | maxVBal[a]+1 | maxVal[a] (0 = None) | 2b(a, b, v) bits | 1b(a, b, mbal, mval) bits
//               (Voting) maxBal[a]+1 | votes[a] as bits b*nv + v
// A permutation of Acceptor permutes the blocks, so the least image over Permutations(Acceptor) is the blocks SORTED; a
// permutation of Value is a fixed bit shuffle inside every word.  The representative of an orbit is the lexicographically
// least (word 0, sorted blocks) over the value permutations: nv! shuffles and one small sort instead of na!·nv! images.
// (TLC keeps the first state of an orbit it meets; the counts do not depend on the choice: the parity tests compare with a brute
// force over all na!·nv! images and with first-met representatives.)
//
// Slots = TLC's enumeration of the action's witnesses (every witness of a bounded \E inside an action is one generated
// successor, also when the quantified formula has no primed variable):
//   Voting  IncreaseMaxBal(a, b);  VoteFor(a, b, v) once per quorum Q and per witness c \in -1..b-1 of ShowsSafeAt(Q, b, v)
//   Paxos   Phase1a(b);  Phase2a(b, v) once per quorum Q and per witness (Q1bv = {}, or m \in Q1bv);  Phase1b(a) per 1a
//           message;  Phase2b(a) per 2a message
#pragma once
#include "mc_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace mc {

struct PaxosParams {
    int kind, na, nv, nb, inv_mask, sym, prop, mut, nq;
    unsigned quorum[8];
    int bb, vb, o_2b, o_1b;  // field widths and offsets inside an acceptor block
};

struct SpecPaxos {
    using Params = PaxosParams;
    static constexpr int XA = 4, XV = 3, XB = 4;
    static constexpr int MAX_WORDS = 1 + XA, FIX_SLOTS = 0, STAGE_WORDS = 0;
    // the INVARIANTs are evaluated once per STORED state, when it is expanded (parent_status), like TLC evaluates them once per
    // new state — not once per generated successor (7 x as many here); the engine checks a run's last, unexpanded level too.
    // The PROPERTY is a predicate of a TRANSITION and stays with the successor (eval).
    static constexpr bool CHECK_ON_EXPAND = true;
    static constexpr bool SLICE_SLOTS = true;  // up to 255 witness slots per state, each a few hundred instructions: slice small frontiers
    struct Local { uint64_t w[MAX_WORDS]; };
    struct View { unsigned votes[XA]; int maxBal[XA]; };  // what Voting's operators read: votes[a] as bits b*nv + v

    MC_HD static int words(const Params &p) { return 1 + p.na; }
    MC_HD static int n_voting_slots(const Params &p) { return p.na * p.nb + p.na * p.nb * p.nv * p.nq * p.nb; }
    MC_HD static int n_p2a_wit(const Params &p) { return 1 + p.na * p.nb; }
    MC_HD static int max_slots(const Params &p) {
        if (p.kind == 1) return n_voting_slots(p);
        return p.nb + p.nb * p.nv * p.nq * n_p2a_wit(p) + p.na * p.nb + p.na * p.nb * p.nv;
    }

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        if (np < 4) return -1;
        memset(&o, 0, sizeof o);
        o.kind = (int)p[0]; o.na = (int)p[1]; o.nv = (int)p[2]; o.nb = (int)p[3];
        o.inv_mask = np > 4 ? (int)p[4] : (o.kind ? 1 : 15);
        o.sym = np > 5 ? (int)p[5] & 3 : 0;
        o.prop = np > 6 ? (int)p[6] & 1 : 1;
        o.mut = np > 6 ? (int)p[6] >> 1 & 1 : 0;   // negative control: Phase2a without its quorum conjunct (specs/paxos/MCPaxosBad.tla)
        o.nq = np > 7 ? (int)p[7] : 0;
        if (o.kind < 0 || o.kind > 1 || o.na < 1 || o.na > XA || o.nv < 1 || o.nv > XV || o.nb < 1 || o.nb > XB || o.nq < 0 ||
            o.nq > 8 || (o.nq > 0 && np < 8u + (unsigned)o.nq)) return -1;
        if (o.nq == 0) {  // all majorities of minimal size (MCVoting.tla:8, MCPaxos.tla:9)
            const int need = o.na / 2 + 1;
            for (unsigned m = 1; m < (1u << o.na); m++)
                if (__builtin_popcount(m) == need) {
                    if (o.nq == 8) return -1;
                    o.quorum[o.nq++] = m;
                }
        } else {
            for (int q = 0; q < o.nq; q++) {
                o.quorum[q] = (unsigned)p[8 + q];
                if (!o.quorum[q] || o.quorum[q] >> o.na) return -1;
            }
        }
        if (o.sym & 1)  // Permutations(Acceptor) must map Quorum onto itself (TLC assumes it): every transposition is checked
            for (int i = 0; i < o.na; i++)
                for (int j = i + 1; j < o.na; j++)
                    for (int q = 0; q < o.nq; q++) {
                        unsigned m = o.quorum[q];
                        const unsigned bi = m >> i & 1u, bj = m >> j & 1u;
                        m = (m & ~(1u << i | 1u << j)) | bj << i | bi << j;
                        bool found = false;
                        for (int r = 0; r < o.nq; r++) found |= o.quorum[r] == m;
                        if (!found) return -1;
                    }
        o.bb = o.nb + 1 <= 2 ? 1 : o.nb + 1 <= 4 ? 2 : 3;
        o.vb = o.nv + 1 <= 2 ? 1 : 2;
        if (o.kind == 1) {
            o.o_2b = o.bb;  // votes
            o.o_1b = o.o_2b + o.nb * o.nv;
            if (o.o_1b > 32) return -1;
        } else {
            o.o_2b = 2 * o.bb + o.vb;
            o.o_1b = o.o_2b + o.nb * o.nv;
            if (o.o_1b + o.nb * (o.nb + 1) * (o.nv + 1) > 64 || o.nb + o.nb * o.nv > 64) return -1;
        }
        return 0;
    }

    // ------------------------------------------------------------------ block accessors
    MC_HD static int b_maxBal(const Params &p, uint64_t blk) { return (int)bits_get(blk, 0, p.bb) - 1; }
    MC_HD static int b_maxVBal(const Params &p, uint64_t blk) { return (int)bits_get(blk, p.bb, p.bb) - 1; }
    MC_HD static int b_maxVal(const Params &p, uint64_t blk) { return (int)bits_get(blk, 2 * p.bb, p.vb) - 1; }
    MC_HD static unsigned b_votes(const Params &p, uint64_t blk) { return (unsigned)bits_get(blk, p.o_2b, p.nb * p.nv); }
    MC_HD static int i_1b(const Params &p, int b, int mb, int mv) { return p.o_1b + (b * (p.nb + 1) + mb) * (p.nv + 1) + mv; }
    MC_HD static void view_of(const Params &p, const uint64_t *w, View &v) {
        for (int a = 0; a < XA; a++) {
            v.votes[a] = a < p.na ? b_votes(p, w[1 + a]) : 0u;
            v.maxBal[a] = a < p.na ? b_maxBal(p, w[1 + a]) : -1;
        }
    }

    // ------------------------------------------------------------------ Voting's operators on a View
    MC_HD static bool voted(const Params &p, const View &s, int a, int b, int v) { return s.votes[a] >> (b * p.nv + v) & 1u; }  // :51
    MC_HD static bool did_not_vote_at(const Params &p, const View &s, int a, int b) {                                           // :66
        return (s.votes[a] >> (b * p.nv) & ((1u << p.nv) - 1u)) == 0;
    }
    MC_HD static unsigned chosen(const Params &p, const View &s) {  // :56-62
        unsigned m = 0;
        for (int v = 0; v < p.nv; v++)
            for (int b = 0; b < p.nb; b++)
                for (int q = 0; q < p.nq; q++) {
                    bool all = true;
                    for (int a = 0; a < p.na; a++)
                        if ((p.quorum[q] >> a & 1u) && !voted(p, s, a, b, v)) all = false;
                    if (all) m |= 1u << v;
                }
        return m;
    }
    MC_HD static bool none_other_choosable_at(const Params &p, const View &s, int b, int v) {  // :76-78 with CannotVoteAt :68-69
        for (int q = 0; q < p.nq; q++) {
            bool all = true;
            for (int a = 0; a < p.na; a++)
                if ((p.quorum[q] >> a & 1u) && !(voted(p, s, a, b, v) || (s.maxBal[a] > b && did_not_vote_at(p, s, a, b)))) all = false;
            if (all) return true;
        }
        return false;
    }
    MC_HD static bool safe_at(const Params &p, const View &s, int b, int v) {  // :84
        for (int c = 0; c < b; c++)
            if (!none_other_choosable_at(p, s, c, v)) return false;
        return true;
    }
    // is c \in -1..(b-1) a witness of ShowsSafeAt(Q, b, v)?  (:111-115)
    MC_HD static bool shows_safe_witness(const Params &p, const View &s, unsigned Q, int b, int v, int c) {
        for (int a = 0; a < p.na; a++)
            if ((Q >> a & 1u) && !(s.maxBal[a] >= b)) return false;
        if (c != -1) {
            bool some = false;
            for (int a = 0; a < p.na; a++) some |= (Q >> a & 1u) && voted(p, s, a, c, v);
            if (!some) return false;
        }
        for (int d = c + 1; d < b; d++)
            for (int a = 0; a < p.na; a++)
                if ((Q >> a & 1u) && !did_not_vote_at(p, s, a, d)) return false;
        return true;
    }
    MC_HD static bool shows_safe_at_any(const Params &p, const View &s, int b, int v) {
        for (int q = 0; q < p.nq; q++)
            for (int c = -1; c < b; c++)
                if (shows_safe_witness(p, s, p.quorum[q], b, v, c)) return true;
        return false;
    }
    MC_HD static bool vote_for_guard(const Params &p, const View &s, int a, int b, int v) {  // :143-146
        if (!(s.maxBal[a] <= b) || !did_not_vote_at(p, s, a, b)) return false;
        for (int c = 0; c < p.na; c++)
            if (c != a && (s.votes[c] >> (b * p.nv) & ((1u << p.nv) - 1u) & ~(1u << v))) return false;
        return true;
    }
    MC_HD static bool voting_inv(const Params &p, const View &s) {  // :160: TypeOK (ranges hold by construction), VotesSafe :95, OneValuePerBallot :101
        for (int a = 0; a < p.na; a++)
            for (int b = 0; b < p.nb; b++)
                for (int v = 0; v < p.nv; v++)
                    if (voted(p, s, a, b, v) && !safe_at(p, s, b, v)) return false;
        for (int b = 0; b < p.nb; b++) {
            unsigned any = 0;
            for (int a = 0; a < p.na; a++) any |= s.votes[a] >> (b * p.nv) & ((1u << p.nv) - 1u);
            if (any & (any - 1u)) return false;
        }
        return true;
    }
    // [V!Next]_<<votes, maxBal>> between two views (Voting.tla:152-156)
    MC_HD static bool voting_step_ok(const Params &p, const View &s, const View &t) {
        int changed = -1;
        for (int a = 0; a < p.na; a++)
            if (s.votes[a] != t.votes[a] || s.maxBal[a] != t.maxBal[a]) {
                if (changed >= 0) return false;
                changed = a;
            }
        if (changed < 0) return true;
        const int a = changed, b = t.maxBal[a];
        if (b < 0) return false;
        if (t.votes[a] == s.votes[a]) return b > s.maxBal[a];  // IncreaseMaxBal(a, b)
        const unsigned added = t.votes[a] ^ s.votes[a];
        if ((t.votes[a] & s.votes[a]) != s.votes[a] || (added & (added - 1u))) return false;
        for (int v = 0; v < p.nv; v++)
            if (added == 1u << (b * p.nv + v)) return vote_for_guard(p, s, a, b, v) && shows_safe_at_any(p, s, b, v);  // VoteFor(a, b, v)
        return false;
    }

    // ------------------------------------------------------------------ SYMMETRY
    MC_HD static uint64_t permute_values_block(const Params &p, uint64_t blk, const int *pv) {
        uint64_t o = blk & ((1ull << p.o_2b) - 1ull);
        if (p.kind == 0) {
            const int mv = b_maxVal(p, blk);
            o = bits_set(o, 2 * p.bb, p.vb, (uint64_t)(mv < 0 ? 0 : pv[mv] + 1));
        }
        for (int b = 0; b < p.nb; b++)
            for (int v = 0; v < p.nv; v++)
                if (blk >> (p.o_2b + b * p.nv + v) & 1ull) o |= 1ull << (p.o_2b + b * p.nv + pv[v]);
        if (p.kind == 0)
            for (int b = 0; b < p.nb; b++)
                for (int mb = 0; mb <= p.nb; mb++) {
                    if (blk >> i_1b(p, b, mb, 0) & 1ull) o |= 1ull << i_1b(p, b, mb, 0);
                    for (int v = 0; v < p.nv; v++)
                        if (blk >> i_1b(p, b, mb, v + 1) & 1ull) o |= 1ull << i_1b(p, b, mb, pv[v] + 1);
                }
        return o;
    }
    MC_HD static uint64_t permute_values_global(const Params &p, uint64_t g, const int *pv) {
        uint64_t o = g & ((1ull << p.nb) - 1ull);
        for (int b = 0; b < p.nb; b++)
            for (int v = 0; v < p.nv; v++)
                if (g >> (p.nb + b * p.nv + v) & 1ull) o |= 1ull << (p.nb + b * p.nv + pv[v]);
        return o;
    }
    MC_HD static void sort_blocks(const Params &p, uint64_t *w) {
        for (int i = 1; i < p.na; i++)
            for (int j = i; j > 0 && w[1 + j] < w[j]; j--) {
                const uint64_t t = w[1 + j]; w[1 + j] = w[j]; w[j] = t;
            }
    }
    MC_HD static void canonicalise(const Params &p, uint64_t *w) {
        if (!p.sym) return;
        if (!(p.sym & 2) || p.nv == 1) {
            sort_blocks(p, w);
            return;
        }
        // value permutations in lexicographic order (nv <= 3: at most 6)
        const int PERMS[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        const int np = p.nv == 2 ? 2 : 6;
        uint64_t best[MAX_WORDS], cand[MAX_WORDS];
        for (int k = 0; k < np; k++) {
            const int *pv = p.nv == 2 ? PERMS[k == 0 ? 0 : 2] : PERMS[k];
            cand[0] = permute_values_global(p, w[0], pv);
            for (int a = 0; a < p.na; a++) cand[1 + a] = permute_values_block(p, w[1 + a], pv);
            if (p.sym & 1) sort_blocks(p, cand);
            bool less = k == 0;
            if (!less)
                for (int i = 0; i <= p.na; i++)
                    if (cand[i] != best[i]) { less = cand[i] < best[i]; break; }
            if (less)
                for (int i = 0; i <= p.na; i++) best[i] = cand[i];
        }
        for (int i = 0; i <= p.na; i++) w[i] = best[i];
    }

    // ------------------------------------------------------------------ engine interface
    MC_HD static uint64_t num_init(const Params &) { return 1; }
    MC_HD static void init(const Params &p, uint64_t, WordRef out) {  // Voting.tla:124-125, Paxos.tla:83-86: all -1 / None / {} = all-zero
        for (int i = 0; i <= p.na; i++) out.set(i, 0);
    }
    MC_HD static uint64_t fp_words(const Params &p, const uint64_t *w) {
        uint64_t h = 0x243f6a8885a308d3ull;
        for (int i = 0; i <= p.na; i++) h = fmix64(h ^ w[i]) + salt_of((unsigned)i);
        return fp_nonzero(h);
    }
    MC_HD static uint64_t fp_of(const Params &p, CWordRef s) {
        uint64_t w[MAX_WORDS];
        for (int i = 0; i <= p.na; i++) w[i] = s.get(i);
        return fp_words(p, w);
    }
    MC_HD static unsigned init_status(const Params &, CWordRef) { return ST_ENABLED; }  // Init satisfies every invariant and C!Init / V!Init
    MC_HD static void load(const Params &p, CWordRef s, Local &l) {
        for (int i = 0; i < MAX_WORDS; i++) l.w[i] = i <= p.na ? s.get(i) : 0;
    }
    MC_HD static int nslots(const Params &p, const Local &) { return max_slots(p); }
    MC_HD static unsigned parent_status(const Params &p, const Local &l, CWordRef) { return check_invariants(p, l.w); }

    // invariants of the cfg on a (not yet canonicalised) state; returns the status bits
    MC_HD static unsigned check_invariants(const Params &p, const uint64_t *w) {
        View s;
        view_of(p, w, s);
        if (p.kind == 1) return (p.inv_mask & 1) && !voting_inv(p, s) ? ST_INVARIANT | (0u << 8) : 0u;
        // Inv!1 = TypeOK (Paxos.tla:77-80) holds by construction of the packed fields
        if (p.inv_mask & 2)  // Inv!2 :193-195
            for (int a = 0; a < p.na; a++) {
                const int vb = b_maxVBal(p, w[1 + a]), mv = b_maxVal(p, w[1 + a]);
                if (vb == -1 ? mv != -1 : (mv < 0 || !voted(p, s, a, vb, mv))) return ST_INVARIANT | (1u << 8);
            }
        if (p.inv_mask & 4) {  // Inv!3 :196-206
            for (int a = 0; a < p.na; a++)
                for (int b = 0; b < p.nb; b++)
                    for (int mb = 0; mb <= p.nb; mb++)
                        for (int mv = 0; mv <= p.nv; mv++) {
                            if (!(w[1 + a] >> i_1b(p, b, mb, mv) & 1ull)) continue;
                            if (!(s.maxBal[a] >= b)) return ST_INVARIANT | (2u << 8);
                            if (mb >= 1 && !(mv >= 1 && voted(p, s, a, mb - 1, mv - 1))) return ST_INVARIANT | (2u << 8);
                        }
            for (int b = 0; b < p.nb; b++) {
                const unsigned sent = (unsigned)(w[0] >> (p.nb + b * p.nv)) & ((1u << p.nv) - 1u);
                if (sent & (sent - 1u)) return ST_INVARIANT | (2u << 8);
                for (int v = 0; v < p.nv; v++)
                    if ((sent >> v & 1u) && !shows_safe_at_any(p, s, b, v)) return ST_INVARIANT | (2u << 8);
            }
        }
        if ((p.inv_mask & 8) && !voting_inv(p, s)) return ST_INVARIANT | (3u << 8);  // Inv!4 = V!Inv :207
        return 0;
    }

    // one (state, slot) pair: the successor in t[] (not canonicalised), status bits (0 = the witness does not exist)
    MC_HD static unsigned step(const Params &p, const uint64_t *w, int slot, uint64_t *t) {
        for (int i = 0; i <= p.na; i++) t[i] = w[i];
        if (p.kind == 1) {  // ---------------------------------------------------------------- Voting
            View s;
            view_of(p, w, s);
            if (slot < p.na * p.nb) {  // IncreaseMaxBal(a, b) :133-136
                const int a = slot / p.nb, b = slot % p.nb;
                if (!(b > s.maxBal[a])) return 0;
                t[1 + a] = bits_set(w[1 + a], 0, p.bb, (uint64_t)(b + 1));
                return ST_ENABLED;
            }
            int k = slot - p.na * p.nb;  // VoteFor(a, b, v) :142-149, witness (Q, c)
            const int c = k % p.nb - 1; k /= p.nb;
            const int q = k % p.nq; k /= p.nq;
            const int v = k % p.nv; k /= p.nv;
            const int b = k % p.nb, a = k / p.nb;
            if (c >= b || !vote_for_guard(p, s, a, b, v) || !shows_safe_witness(p, s, p.quorum[q], b, v, c)) return 0;
            t[1 + a] = bits_set(w[1 + a], 0, p.bb, (uint64_t)(b + 1)) | 1ull << (p.o_2b + b * p.nv + v);
            unsigned st = ST_ENABLED;
            if (p.prop) {  // [C!Next]_chosen (Consensus.tla:26-27)
                View n;
                view_of(p, t, n);
                const unsigned c0 = chosen(p, s), c1 = chosen(p, n);
                if (!(c0 == c1 || (c0 == 0 && c1 != 0 && (c1 & (c1 - 1u)) == 0))) st |= ST_INVARIANT | (1u << 8);
            }
            return st;
        }
        // -------------------------------------------------------------------------------------- Paxos
        int k = slot;
        if (k < p.nb) {  // Phase1a(b) :93-94
            t[0] = w[0] | 1ull << k;
            return ST_ENABLED;
        }
        k -= p.nb;
        const int n2a = p.nb * p.nv * p.nq * n_p2a_wit(p);
        if (k < n2a) {  // Phase2a(b, v) :135-151, witness (Q, m)
            const int m = k % n_p2a_wit(p); k /= n_p2a_wit(p);
            const int q = k % p.nq; k /= p.nq;
            const int v = k % p.nv, b = k / p.nv;
            if ((unsigned)(w[0] >> (p.nb + b * p.nv)) & ((1u << p.nv) - 1u)) return 0;  // ~ \E m \in msgs : m.type = "2a" /\ m.bal = b
            if (p.mut) {
                if (q || m) return 0;
            } else {
                const unsigned Q = p.quorum[q];
                int best = -1, nq1bv = 0;
                for (int a = 0; a < p.na; a++) {
                    if (!(Q >> a & 1u)) continue;
                    const uint64_t row = w[1 + a] >> i_1b(p, b, 0, 0) & ((1ull << ((p.nb + 1) * (p.nv + 1))) - 1ull);
                    if (!row) return 0;  // \A a \in Q : \E m \in Q1b : m.acc = a
                    for (int mb = 1; mb <= p.nb; mb++)
                        for (int mv = 0; mv <= p.nv; mv++)
                            if (row >> (mb * (p.nv + 1) + mv) & 1ull) { nq1bv++; best = mb - 1 > best ? mb - 1 : best; }
                }
                if (m == 0) {
                    if (nq1bv) return 0;  // \/ Q1bv = {}
                } else {  // \/ \E m \in Q1bv : m.mval = v /\ \A mm \in Q1bv : m.mbal >= mm.mbal
                    const int a = (m - 1) / p.nb, mb = (m - 1) % p.nb;
                    if (!(Q >> a & 1u) || !(w[1 + a] >> i_1b(p, b, mb + 1, v + 1) & 1ull) || mb < best) return 0;
                }
            }
            t[0] = w[0] | 1ull << (p.nb + b * p.nv + v);
            return ST_ENABLED;
        }
        k -= n2a;
        unsigned st = ST_ENABLED;
        int a;
        if (k < p.na * p.nb) {  // Phase1b(a) :109-116, witness the 1a message of ballot b
            a = k / p.nb;
            const int b = k % p.nb;
            if (!(w[0] >> b & 1ull) || !(b > b_maxBal(p, w[1 + a]))) return 0;
            t[1 + a] = bits_set(w[1 + a], 0, p.bb, (uint64_t)(b + 1)) |
                       1ull << i_1b(p, b, b_maxVBal(p, w[1 + a]) + 1, b_maxVal(p, w[1 + a]) + 1);
        } else {  // Phase2b(a) :161-167, witness the 2a message (b, v)
            k -= p.na * p.nb;
            const int v = k % p.nv; k /= p.nv;
            const int b = k % p.nb;
            a = k / p.nb;
            if (!(w[0] >> (p.nb + b * p.nv + v) & 1ull) || !(b >= b_maxBal(p, w[1 + a]))) return 0;
            uint64_t blk = bits_set(w[1 + a], 0, p.bb, (uint64_t)(b + 1));
            blk = bits_set(blk, p.bb, p.bb, (uint64_t)(b + 1));
            blk = bits_set(blk, 2 * p.bb, p.vb, (uint64_t)(v + 1));
            t[1 + a] = blk | 1ull << (p.o_2b + b * p.nv + v);
        }
        if (p.prop) {  // [V!Next]_<<votes, maxBal>> under votes == ... (Paxos.tla:184-188)
            View s, n;
            view_of(p, w, s);
            view_of(p, t, n);
            if (!voting_step_ok(p, s, n)) st |= ST_INVARIANT | (4u << 8);
        }
        return st;
    }

    MC_HD static unsigned successor(const Params &p, const uint64_t *w, int slot, uint64_t *t) {
        unsigned st = step(p, w, slot, t);
        if (!st) return 0;
        bool same = true;
        for (int i = 0; i <= p.na; i++) same &= t[i] == w[i];
        if (same) return st | ST_SELFLOOP;  // e.g. Phase1a(b) of a ballot whose 1a message is already in msgs
        canonicalise(p, t);
        return st;
    }
    MC_HD static unsigned eval(const Params &p, const Local &l, CWordRef, int slot, uint64_t &fp) {
        uint64_t t[MAX_WORDS];
        const unsigned st = successor(p, l.w, slot, t);
        if (st & ST_ENABLED) fp = fp_words(p, t);
        return st;
    }
    MC_HD static unsigned apply(const Params &p, CWordRef s, int slot, WordRef out) {
        uint64_t w[MAX_WORDS], t[MAX_WORDS];
        for (int i = 0; i < MAX_WORDS; i++) w[i] = i <= p.na ? s.get(i) : 0;
        const unsigned st = successor(p, w, slot, t);
        for (int i = 0; i <= p.na; i++) out.set(i, st ? t[i] : w[i]);
        return st;
    }

    // ------------------------------------------------------------------ host side: actions and TLA+ text
    static int action_of(const Params &p, const uint64_t *, int slot) {
        if (p.kind == 1) return slot < p.na * p.nb ? 0 : 1;
        if (slot < p.nb) return 2;
        slot -= p.nb;
        if (slot < p.nb * p.nv * p.nq * n_p2a_wit(p)) return 3;
        slot -= p.nb * p.nv * p.nq * n_p2a_wit(p);
        return slot < p.na * p.nb ? 4 : 5;
    }
    static const char *action_name(int a) {
        static const char *nm[] = {"IncreaseMaxBal", "VoteFor", "Phase1a", "Phase2a", "Phase1b", "Phase2b"};
        return a >= 0 && a < 6 ? nm[a] : a < 0 ? "Initial predicate" : "?";
    }
    static int format(const Params &p, const uint64_t *w, char *buf, size_t cap) {
        // same canonical text as the oracle: functions as (k :> v @@ ...), set elements sorted by their text
        size_t k = 0;
        auto put = [&](const char *fmt, auto... a) {
            if (k >= cap) return;
            if constexpr (sizeof...(a) == 0) k += (size_t)snprintf(buf + k, cap - k, "%s", fmt);
            else k += (size_t)snprintf(buf + k, cap - k, fmt, a...);
        };
        static thread_local char items[320][96];
        char *ptr[320];
        auto emit_sorted = [&](int n) {
            for (int i = 0; i < n; i++) ptr[i] = items[i];
            qsort(ptr, (size_t)n, sizeof ptr[0], [](const void *x, const void *y) { return strcmp(*(char *const *)x, *(char *const *)y); });
            for (int i = 0; i < n; i++) put("%s%s", i ? ", " : "", ptr[i]);
        };
        if (p.kind == 1) {
            put("/\\ votes = (");
            for (int a = 0; a < p.na; a++) {
                put("%sa%d :> {", a ? " @@ " : "", a + 1);
                int n = 0;
                for (int b = 0; b < p.nb; b++)
                    for (int v = 0; v < p.nv; v++)
                        if (w[1 + a] >> (p.o_2b + b * p.nv + v) & 1ull) snprintf(items[n++], 96, "<<%d, v%d>>", b, v + 1);
                emit_sorted(n);
                put("}");
            }
            put(")\n/\\ maxBal = (");
            for (int a = 0; a < p.na; a++) put("%sa%d :> %d", a ? " @@ " : "", a + 1, b_maxBal(p, w[1 + a]));
            put(")");
            return (int)(k < cap ? k : cap);
        }
        put("/\\ maxBal = (");
        for (int a = 0; a < p.na; a++) put("%sa%d :> %d", a ? " @@ " : "", a + 1, b_maxBal(p, w[1 + a]));
        put(")\n/\\ maxVBal = (");
        for (int a = 0; a < p.na; a++) put("%sa%d :> %d", a ? " @@ " : "", a + 1, b_maxVBal(p, w[1 + a]));
        put(")\n/\\ maxVal = (");
        for (int a = 0; a < p.na; a++) {
            const int mv = b_maxVal(p, w[1 + a]);
            if (mv < 0) put("%sa%d :> None", a ? " @@ " : "", a + 1);
            else put("%sa%d :> v%d", a ? " @@ " : "", a + 1, mv + 1);
        }
        put(")\n/\\ msgs = {");
        int n = 0;
        for (int b = 0; b < p.nb; b++)
            if (w[0] >> b & 1ull) snprintf(items[n++], 96, "[bal |-> %d, type |-> \"1a\"]", b);
        for (int b = 0; b < p.nb; b++)
            for (int v = 0; v < p.nv; v++)
                if (w[0] >> (p.nb + b * p.nv + v) & 1ull) snprintf(items[n++], 96, "[bal |-> %d, type |-> \"2a\", val |-> v%d]", b, v + 1);
        for (int a = 0; a < p.na; a++)
            for (int b = 0; b < p.nb; b++) {
                for (int v = 0; v < p.nv; v++)
                    if (w[1 + a] >> (p.o_2b + b * p.nv + v) & 1ull)
                        snprintf(items[n++], 96, "[acc |-> a%d, bal |-> %d, type |-> \"2b\", val |-> v%d]", a + 1, b, v + 1);
                for (int mb = 0; mb <= p.nb; mb++)
                    for (int mv = 0; mv <= p.nv; mv++)
                        if (w[1 + a] >> i_1b(p, b, mb, mv) & 1ull) {
                            char val[16];
                            if (mv) snprintf(val, sizeof val, "v%d", mv); else snprintf(val, sizeof val, "None");
                            snprintf(items[n++], 96, "[acc |-> a%d, bal |-> %d, mbal |-> %d, mval |-> %s, type |-> \"1b\"]", a + 1, b, mb - 1, val);
                        }
            }
        emit_sorted(n);
        put("}");
        return (int)(k < cap ? k : cap);
    }
};

}  // namespace mc
