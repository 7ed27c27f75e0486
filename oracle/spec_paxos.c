/*
 * oracle/spec_paxos.c — CPU ORACLE (test infrastructure) for the reference's Paxos family:
 *   examples/Paxos/Voting.tla:133-160 (IncreaseMaxBal, VoteFor, Next; Inv :160; C == INSTANCE Consensus :185)
 *   examples/Paxos/Paxos.tla:93-179  (Phase1a, Phase1b, Phase2a, Phase2b, Next; votes :184; V == INSTANCE Voting :188;
 *                                     Inv :192-208)
 * under the model modules examples/Paxos/MCVoting.tla + .cfg and MCPaxos.tla + .cfg (constants as sets of model values,
 * Ballot <- 0..MaxBallot, SYMMETRY Permutations(Acceptor) \cup Permutations(Value), PROPERTY C!Spec / V!Spec).
 *
 * Unpacked restatement: a state is a struct of small arrays indexed the way the spec indexes them; every message of the
 * finite set Message (Paxos.tla:57-61) has one presence byte.  PINNED to the reference's own text: oracle/tlaplus.py evaluates
 * Voting.tla / Paxos.tla directly and tests/test_reference_text_paxos.py compares counters, per-level counts and per-level
 * state SETS (as canonical TLA+ text) with this file.
 *
 * Counting follows TLC's enumeration of an action: every witness of a bounded \E inside an action
 * is one generated successor, ALSO when the quantified formula has no primed variable — `\E Q \in Quorum : ShowsSafeAt(Q, b, v)`
 * (Voting.tla:147) generates VoteFor's successor once per quorum Q and once per witness c of ShowsSafeAt's own \E
 * (Voting.tla:113-115); Phase2a (Paxos.tla:138-149) once per quorum and per witness m of `\E m \in Q1bv`.
 *
 * PROPERTY: the safety part of C!Spec / V!Spec — the refinement mapping's Init on the initial state and [Next]_v on every
 * generated transition — is checked per successor; a violation is reported as invariant index PX_PROP (after the cfg's
 * invariants).  Liveness is not checked (neither cfg asks for it).
 *
 * params: {kind (0 Paxos, 1 Voting), nAcceptor (1..4), nValue (1..3), nBallot = MaxBallot + 1 (1..4), invariant mask,
 *          symmetry (bit 0 Permutations(Acceptor), bit 1 Permutations(Value)),
 *          property (bit 0: check the PROPERTY; bit 1: negative control, Phase2a without its quorum conjunct),
 *          nQuorum (0 = all majorities of minimal size), quorum bit masks ...}
 *   invariant mask, Paxos: bit k = Inv!(k+1) of Paxos.tla:192-208 (MCPaxos.tla:65-68 Inv1..Inv4);
 *                   Voting: bit 0 = Inv (Voting.tla:160 TypeOK /\ VotesSafe /\ OneValuePerBallot).
 */
#include "oracle_int.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define XA 4 /* max acceptors */
#define XV 3 /* max values    */
#define XB 4 /* max ballots   */
#define XQ 8

typedef struct {
    int kind, na, nv, nb, inv_mask, sym, prop, mut, nq;
    unsigned quorum[XQ]; /* acceptor bit masks */
} px_ctx;

/* ------------------------------------------------------------------------------------------ Voting */
typedef struct {
    int8_t maxBal[XA];           /* -1 .. nb-1 */
    uint8_t votes[XA][XB][XV];   /* <<b, v>> \in votes[a] */
} vt_state;

static int vt_VotedFor(const vt_state *s, int a, int b, int v) { return s->votes[a][b][v]; }           /* Voting.tla:51 */
static int vt_DidNotVoteAt(const px_ctx *x, const vt_state *s, int a, int b) {                         /* :66 */
    for (int v = 0; v < x->nv; v++) if (vt_VotedFor(s, a, b, v)) return 0;
    return 1;
}
static int vt_ChosenAt(const px_ctx *x, const vt_state *s, int b, int v) {                             /* :56-57 */
    for (int q = 0; q < x->nq; q++) {
        int all = 1;
        for (int a = 0; a < x->na; a++) if ((x->quorum[q] >> a & 1) && !vt_VotedFor(s, a, b, v)) all = 0;
        if (all) return 1;
    }
    return 0;
}
static unsigned vt_chosen(const px_ctx *x, const vt_state *s) {                                        /* :62 */
    unsigned m = 0;
    for (int v = 0; v < x->nv; v++)
        for (int b = 0; b < x->nb; b++) if (vt_ChosenAt(x, s, b, v)) m |= 1u << v;
    return m;
}
static int vt_CannotVoteAt(const px_ctx *x, const vt_state *s, int a, int b) {                         /* :68-69 */
    return s->maxBal[a] > b && vt_DidNotVoteAt(x, s, a, b);
}
static int vt_NoneOtherChoosableAt(const px_ctx *x, const vt_state *s, int b, int v) {                 /* :76-78 */
    for (int q = 0; q < x->nq; q++) {
        int all = 1;
        for (int a = 0; a < x->na; a++)
            if ((x->quorum[q] >> a & 1) && !(vt_VotedFor(s, a, b, v) || vt_CannotVoteAt(x, s, a, b))) all = 0;
        if (all) return 1;
    }
    return 0;
}
static int vt_SafeAt(const px_ctx *x, const vt_state *s, int b, int v) {                               /* :84 */
    for (int c = 0; c <= b - 1; c++) if (!vt_NoneOtherChoosableAt(x, s, c, v)) return 0;
    return 1;
}
/* number of witnesses c \in -1..(b-1) of ShowsSafeAt(Q, b, v) (:111-115); 0 = it does not hold */
static int vt_ShowsSafeAt_witnesses(const px_ctx *x, const vt_state *s, unsigned Q, int b, int v) {
    for (int a = 0; a < x->na; a++) if ((Q >> a & 1) && !(s->maxBal[a] >= b)) return 0;
    int n = 0;
    for (int c = -1; c <= b - 1; c++) {
        int ok = 1;
        if (c != -1) {
            int some = 0;
            for (int a = 0; a < x->na; a++) if ((Q >> a & 1) && vt_VotedFor(s, a, c, v)) some = 1;
            ok = some;
        }
        for (int d = c + 1; ok && d <= b - 1; d++)
            for (int a = 0; a < x->na; a++) if ((Q >> a & 1) && !vt_DidNotVoteAt(x, s, a, d)) ok = 0;
        n += ok;
    }
    return n;
}
static int vt_Inv(const px_ctx *x, const vt_state *s) {  /* :160 TypeOK /\ VotesSafe /\ OneValuePerBallot (:95-103) */
    for (int a = 0; a < x->na; a++) if (s->maxBal[a] < -1 || s->maxBal[a] >= x->nb) return 0;
    for (int a = 0; a < x->na; a++)
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++) if (vt_VotedFor(s, a, b, v) && !vt_SafeAt(x, s, b, v)) return 0;
    for (int a1 = 0; a1 < x->na; a1++)
        for (int a2 = 0; a2 < x->na; a2++)
            for (int b = 0; b < x->nb; b++)
                for (int v1 = 0; v1 < x->nv; v1++)
                    for (int v2 = 0; v2 < x->nv; v2++)
                        if (vt_VotedFor(s, a1, b, v1) && vt_VotedFor(s, a2, b, v2) && v1 != v2) return 0;
    return 1;
}
/* [C!Next]_chosen (Consensus.tla:26-27): chosen' = chosen, or chosen = {} and chosen' = {v} */
static int consensus_step_ok(unsigned before, unsigned after) {
    if (before == after) return 1;
    return before == 0 && after != 0 && (after & (after - 1)) == 0;
}
/* guard of VoteFor(a, b, v) without its last enabling conjunct (:143-146) */
static int vt_VoteFor_guard(const px_ctx *x, const vt_state *s, int a, int b, int v) {
    if (!(s->maxBal[a] <= b)) return 0;
    for (int w = 0; w < x->nv; w++) if (s->votes[a][b][w]) return 0;          /* \A vt \in votes[a] : vt[1] # b */
    for (int c = 0; c < x->na; c++) {
        if (c == a) continue;
        for (int w = 0; w < x->nv; w++) if (s->votes[c][b][w] && w != v) return 0;
    }
    return 1;
}

enum { VT_INCREASE = 0, VT_VOTEFOR = 1 };
static int vt_n_init(void *c) { (void)c; return 1; }
static size_t vt_init(void *c, int k, uint8_t *out) {  /* Init :124-125 */
    (void)c; (void)k;
    vt_state s;
    memset(&s, 0, sizeof s);
    for (int a = 0; a < XA; a++) s.maxBal[a] = -1;
    memcpy(out, &s, sizeof s);
    return sizeof s;
}
static void vt_succ(void *c, const uint8_t *sb, size_t len, or_emit *em) {
    const px_ctx *x = c;
    vt_state s, t;
    memcpy(&s, sb, sizeof s);
    (void)len;
    const unsigned ch = x->prop ? vt_chosen(x, &s) : 0;
    /* Next == \E a \in Acceptor, b \in Ballot : IncreaseMaxBal(a, b) \/ \E v \in Value : VoteFor(a, b, v)   (:152-154) */
    for (int a = 0; a < x->na; a++)
        for (int b = 0; b < x->nb; b++) {
            if (b > s.maxBal[a]) {  /* IncreaseMaxBal :133-136 */
                t = s;
                t.maxBal[a] = (int8_t)b;
                em->emit(em, (const uint8_t *)&t, sizeof t, VT_INCREASE, 0);  /* votes unchanged: chosen unchanged */
            }
            for (int v = 0; v < x->nv; v++) {
                if (!vt_VoteFor_guard(x, &s, a, b, v)) continue;
                for (int q = 0; q < x->nq; q++) {
                    const int w = vt_ShowsSafeAt_witnesses(x, &s, x->quorum[q], b, v);
                    if (!w) continue;
                    t = s;
                    t.votes[a][b][v] = 1;
                    t.maxBal[a] = (int8_t)b;
                    unsigned fl = 0;
                    if (x->prop && !consensus_step_ok(ch, vt_chosen(x, &t))) fl = OR_FLAG_PROPERTY | (1u << 8);
                    for (int k = 0; k < w; k++) em->emit(em, (const uint8_t *)&t, sizeof t, VT_VOTEFOR, fl);
                }
            }
        }
}
static int vt_invariant(void *c, const uint8_t *sb, size_t len) {
    const px_ctx *x = c;
    vt_state s;
    memcpy(&s, sb, sizeof s);
    (void)len;
    if ((x->inv_mask & 1) && !vt_Inv(x, &s)) return 0;
    return -1;
}

/* ---- canonical TLA+ text (the format of oracle/tlaplus.py fmt: functions as (k :> v @@ ...), sets sorted by text) */
static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
static size_t join_sorted(char items[][96], int n, const char *sep, char *out, size_t cap) {
    char *ptr[320];
    for (int i = 0; i < n; i++) ptr[i] = items[i];
    qsort(ptr, (size_t)n, sizeof ptr[0], cmp_str);
    size_t k = 0;
    for (int i = 0; i < n; i++) k += (size_t)snprintf(out + k, cap - k, "%s%s", i ? sep : "", ptr[i]);
    if (!n && cap) out[0] = 0;
    return k;
}
static size_t vt_print(void *c, const uint8_t *sb, size_t len, char *buf, size_t cap) {
    const px_ctx *x = c;
    vt_state s;
    memcpy(&s, sb, sizeof s);
    (void)len;
    size_t k = 0;
    char el[XA][96 * 4], items[320][96], tmp[1024];
    for (int a = 0; a < x->na; a++) {
        int n = 0;
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++)
                if (s.votes[a][b][v]) snprintf(items[n++], 96, "<<%d, v%d>>", b, v + 1);
        join_sorted(items, n, ", ", tmp, sizeof tmp);
        snprintf(el[a], sizeof el[a], "a%d :> {%s}", a + 1, tmp);
    }
    k += (size_t)snprintf(buf + k, cap - k, "/\\ votes = (");
    for (int a = 0; a < x->na; a++) k += (size_t)snprintf(buf + k, cap - k, "%s%s", a ? " @@ " : "", el[a]);
    k += (size_t)snprintf(buf + k, cap - k, ")\n/\\ maxBal = (");
    for (int a = 0; a < x->na; a++) k += (size_t)snprintf(buf + k, cap - k, "%sa%d :> %d", a ? " @@ " : "", a + 1, s.maxBal[a]);
    k += (size_t)snprintf(buf + k, cap - k, ")");
    return k;
}

/* ------------------------------------------------------------------------------------------ Paxos */
typedef struct {
    int8_t maxBal[XA], maxVBal[XA];  /* -1 .. nb-1 */
    int8_t maxVal[XA];               /* -1 = None, else value index */
    uint8_t m1a[XB];                                  /* [type |-> "1a", bal |-> b]                       */
    uint8_t m1b[XA][XB][XB + 1][XV + 1];              /* acc, bal, mbal + 1, mval + 1 (0 = None)          */
    uint8_t m2a[XB][XV];                              /* bal, val                                         */
    uint8_t m2b[XA][XB][XV];                          /* acc, bal, val                                    */
} px_state;

/* votes == [a \in Acceptor |-> {<<m.bal, m.val>> : m \in {mm \in msgs : mm.type = "2b" /\ mm.acc = a}}]   (Paxos.tla:184-186) */
static void px_votes(const px_ctx *x, const px_state *s, vt_state *o) {
    memset(o, 0, sizeof *o);
    for (int a = 0; a < x->na; a++) {
        o->maxBal[a] = s->maxBal[a];
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++) o->votes[a][b][v] = s->m2b[a][b][v];
    }
    for (int a = x->na; a < XA; a++) o->maxBal[a] = -1;
}

enum { PX_PHASE1A = 0, PX_PHASE2A = 1, PX_PHASE1B = 2, PX_PHASE2B = 3 };
static int px_n_init(void *c) { (void)c; return 1; }
static size_t px_init(void *c, int k, uint8_t *out) {  /* Init :83-86 */
    (void)c; (void)k;
    px_state s;
    memset(&s, 0, sizeof s);
    for (int a = 0; a < XA; a++) s.maxBal[a] = s.maxVBal[a] = s.maxVal[a] = -1;
    memcpy(out, &s, sizeof s);
    return sizeof s;
}
/* does the step s -> t satisfy [V!Next]_<<votes, maxBal>> (Voting.tla:152-156 under the refinement mapping)? */
static int px_voting_step_ok(const px_ctx *x, const px_state *s, const px_state *t) {
    vt_state vs, vt_;
    px_votes(x, s, &vs);
    px_votes(x, t, &vt_);
    if (!memcmp(&vs, &vt_, sizeof vs)) return 1;
    for (int a = 0; a < x->na; a++)
        for (int b = 0; b < x->nb; b++) {
            vt_state e = vs;
            if (b > vs.maxBal[a]) {
                e.maxBal[a] = (int8_t)b;
                if (!memcmp(&e, &vt_, sizeof e)) return 1;
            }
            for (int v = 0; v < x->nv; v++) {
                if (!vt_VoteFor_guard(x, &vs, a, b, v)) continue;
                int shows = 0;
                for (int q = 0; q < x->nq; q++) shows |= vt_ShowsSafeAt_witnesses(x, &vs, x->quorum[q], b, v) > 0;
                if (!shows) continue;
                e = vs;
                e.votes[a][b][v] = 1;
                e.maxBal[a] = (int8_t)b;
                if (!memcmp(&e, &vt_, sizeof e)) return 1;
            }
        }
    return 0;
}
static void px_emit(const px_ctx *x, const px_state *s, const px_state *t, int action, int times, or_emit *em) {
    unsigned fl = 0;
    if (x->prop && !px_voting_step_ok(x, s, t)) fl = OR_FLAG_PROPERTY | (4u << 8);
    for (int k = 0; k < times; k++) em->emit(em, (const uint8_t *)t, sizeof *t, action, fl);
}
static void px_succ(void *c, const uint8_t *sb, size_t len, or_emit *em) {
    const px_ctx *x = c;
    px_state s, t;
    memcpy(&s, sb, sizeof s);
    (void)len;
    /* Next == \/ \E b \in Ballot : Phase1a(b) \/ \E v \in Value : Phase2a(b, v)
     *         \/ \E a \in Acceptor : Phase1b(a) \/ Phase2b(a)                                  (Paxos.tla:173-176) */
    for (int b = 0; b < x->nb; b++) {
        t = s;                       /* Phase1a(b) :93-94: always enabled, Send is a set union */
        t.m1a[b] = 1;
        px_emit(x, &s, &t, PX_PHASE1A, 1, em);
        for (int v = 0; v < x->nv; v++) {  /* Phase2a(b, v) :135-151 */
            int sent = 0;
            for (int w = 0; w < x->nv; w++) sent |= s.m2a[b][w];
            if (sent) continue;
            int times = x->mut ? 1 : 0;
            for (int q = 0; q < x->nq && !x->mut; q++) {
                const unsigned Q = x->quorum[q];
                /* Q1b = 1b messages of ballot b from acceptors of Q; Q1bv = those with mbal >= 0 */
                int every = 1, nq1bv = 0, best = -1;
                for (int a = 0; a < x->na; a++) {
                    if (!(Q >> a & 1)) continue;
                    int any = 0;
                    for (int mb = 0; mb <= x->nb; mb++)
                        for (int mv = 0; mv <= x->nv; mv++)
                            if (s.m1b[a][b][mb][mv]) {
                                any = 1;
                                if (mb >= 1) { nq1bv++; if (mb - 1 > best) best = mb - 1; }
                            }
                    if (!any) every = 0;
                }
                if (!every) continue;
                if (nq1bv == 0) { times++; continue; }                 /* \/ Q1bv = {} */
                for (int a = 0; a < x->na; a++) {                      /* \/ \E m \in Q1bv : m.mval = v /\ \A mm : m.mbal >= mm.mbal */
                    if (!(Q >> a & 1)) continue;
                    for (int mb = 1; mb <= x->nb; mb++)
                        if (s.m1b[a][b][mb][v + 1] && mb - 1 >= best) times++;
                }
            }
            if (!times) continue;
            t = s;
            t.m2a[b][v] = 1;
            px_emit(x, &s, &t, PX_PHASE2A, times, em);
        }
    }
    for (int a = 0; a < x->na; a++) {
        for (int b = 0; b < x->nb; b++) {  /* Phase1b(a) :109-116: \E m \in msgs of type 1a with m.bal > maxBal[a] */
            if (!s.m1a[b] || !(b > s.maxBal[a])) continue;
            t = s;
            t.maxBal[a] = (int8_t)b;
            t.m1b[a][b][s.maxVBal[a] + 1][s.maxVal[a] + 1] = 1;
            px_emit(x, &s, &t, PX_PHASE1B, 1, em);
        }
        for (int b = 0; b < x->nb; b++)    /* Phase2b(a) :161-167: \E m \in msgs of type 2a with m.bal >= maxBal[a] */
            for (int v = 0; v < x->nv; v++) {
                if (!s.m2a[b][v] || !(b >= s.maxBal[a])) continue;
                t = s;
                t.maxBal[a] = t.maxVBal[a] = (int8_t)b;
                t.maxVal[a] = (int8_t)v;
                t.m2b[a][b][v] = 1;
                px_emit(x, &s, &t, PX_PHASE2B, 1, em);
            }
    }
}
static int px_invariant(void *c, const uint8_t *sb, size_t len) {
    const px_ctx *x = c;
    px_state s;
    vt_state vs;
    memcpy(&s, sb, sizeof s);
    (void)len;
    px_votes(x, &s, &vs);
    if (x->inv_mask & 1)  /* Inv!1 = TypeOK :77-80: the ranges of the three functions (msgs \subseteq Message holds by construction) */
        for (int a = 0; a < x->na; a++)
            if (s.maxBal[a] < -1 || s.maxBal[a] >= x->nb || s.maxVBal[a] < -1 || s.maxVBal[a] >= x->nb || s.maxVal[a] < -1 ||
                s.maxVal[a] >= x->nv) return 0;
    if (x->inv_mask & 2)  /* Inv!2 :193-195 */
        for (int a = 0; a < x->na; a++) {
            if (s.maxVBal[a] == -1) { if (s.maxVal[a] != -1) return 1; }
            else if (s.maxVal[a] < 0 || !vs.votes[a][s.maxVBal[a]][s.maxVal[a]]) return 1;
        }
    if (x->inv_mask & 4) {  /* Inv!3 :196-206 */
        for (int a = 0; a < x->na; a++)
            for (int b = 0; b < x->nb; b++)
                for (int mb = 0; mb <= x->nb; mb++)
                    for (int mv = 0; mv <= x->nv; mv++) {
                        if (!s.m1b[a][b][mb][mv]) continue;
                        if (!(s.maxBal[a] >= b)) return 2;
                        if (mb >= 1 && !(mv >= 1 && vs.votes[a][mb - 1][mv - 1])) return 2;
                    }
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++) {
                if (!s.m2a[b][v]) continue;
                int shows = 0;
                for (int q = 0; q < x->nq; q++) shows |= vt_ShowsSafeAt_witnesses(x, &vs, x->quorum[q], b, v) > 0;
                if (!shows) return 2;
                for (int w = 0; w < x->nv; w++) if (s.m2a[b][w] && w != v) return 2;
            }
    }
    if ((x->inv_mask & 8) && !vt_Inv(x, &vs)) return 3;  /* Inv!4 = V!Inv :207 */
    return -1;
}
static size_t px_print(void *c, const uint8_t *sb, size_t len, char *buf, size_t cap) {
    const px_ctx *x = c;
    px_state s;
    memcpy(&s, sb, sizeof s);
    (void)len;
    static __thread char items[320][96];
    int n = 0;
    char mv[16], val[16];
    for (int b = 0; b < x->nb; b++) if (s.m1a[b]) snprintf(items[n++], 96, "[bal |-> %d, type |-> \"1a\"]", b);
    for (int a = 0; a < x->na; a++)
        for (int b = 0; b < x->nb; b++)
            for (int mb = 0; mb <= x->nb; mb++)
                for (int w = 0; w <= x->nv; w++)
                    if (s.m1b[a][b][mb][w]) {
                        if (w) snprintf(mv, sizeof mv, "v%d", w); else snprintf(mv, sizeof mv, "None");
                        snprintf(items[n++], 96, "[acc |-> a%d, bal |-> %d, mbal |-> %d, mval |-> %s, type |-> \"1b\"]", a + 1, b, mb - 1, mv);
                    }
    for (int b = 0; b < x->nb; b++)
        for (int v = 0; v < x->nv; v++) if (s.m2a[b][v]) snprintf(items[n++], 96, "[bal |-> %d, type |-> \"2a\", val |-> v%d]", b, v + 1);
    for (int a = 0; a < x->na; a++)
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++)
                if (s.m2b[a][b][v]) snprintf(items[n++], 96, "[acc |-> a%d, bal |-> %d, type |-> \"2b\", val |-> v%d]", a + 1, b, v + 1);
    size_t k = 0;
    k += (size_t)snprintf(buf + k, cap - k, "/\\ maxBal = (");
    for (int a = 0; a < x->na; a++) k += (size_t)snprintf(buf + k, cap - k, "%sa%d :> %d", a ? " @@ " : "", a + 1, s.maxBal[a]);
    k += (size_t)snprintf(buf + k, cap - k, ")\n/\\ maxVBal = (");
    for (int a = 0; a < x->na; a++) k += (size_t)snprintf(buf + k, cap - k, "%sa%d :> %d", a ? " @@ " : "", a + 1, s.maxVBal[a]);
    k += (size_t)snprintf(buf + k, cap - k, ")\n/\\ maxVal = (");
    for (int a = 0; a < x->na; a++) {
        if (s.maxVal[a] >= 0) snprintf(val, sizeof val, "v%d", s.maxVal[a] + 1); else snprintf(val, sizeof val, "None");
        k += (size_t)snprintf(buf + k, cap - k, "%sa%d :> %s", a ? " @@ " : "", a + 1, val);
    }
    k += (size_t)snprintf(buf + k, cap - k, ")\n/\\ msgs = {");
    k += join_sorted(items, n, ", ", buf + k, cap - k);
    k += (size_t)snprintf(buf + k, cap - k, "}");
    return k;
}

/* ------------------------------------------------------------------------------------------ SYMMETRY
 * TLC keeps one state per orbit of the group generated by the cfg's permutations (MCVoting.tla:10, p-manual 4.7.3 p.41).
 * Brute force: the orbit's key is the byte-wise least image over every (acceptor permutation, value permutation). */
static int next_perm(int *p, int n) {
    int i = n - 2;
    while (i >= 0 && p[i] > p[i + 1]) i--;
    if (i < 0) return 0;
    int j = n - 1;
    while (p[j] < p[i]) j--;
    int t = p[i]; p[i] = p[j]; p[j] = t;
    for (int a = i + 1, b = n - 1; a < b; a++, b--) { t = p[a]; p[a] = p[b]; p[b] = t; }
    return 1;
}
static void vt_apply(const px_ctx *x, const vt_state *s, const int *pa, const int *pv, vt_state *o) {
    memset(o, 0, sizeof *o);
    for (int a = 0; a < XA; a++) o->maxBal[a] = -1;
    for (int a = 0; a < x->na; a++) {
        o->maxBal[pa[a]] = s->maxBal[a];
        for (int b = 0; b < x->nb; b++)
            for (int v = 0; v < x->nv; v++) o->votes[pa[a]][b][pv[v]] = s->votes[a][b][v];
    }
}
static void px_apply(const px_ctx *x, const px_state *s, const int *pa, const int *pv, px_state *o) {
    memset(o, 0, sizeof *o);
    for (int a = 0; a < XA; a++) o->maxBal[a] = o->maxVBal[a] = o->maxVal[a] = -1;
    memcpy(o->m1a, s->m1a, sizeof o->m1a);
    for (int a = 0; a < x->na; a++) {
        o->maxBal[pa[a]] = s->maxBal[a];
        o->maxVBal[pa[a]] = s->maxVBal[a];
        o->maxVal[pa[a]] = s->maxVal[a] < 0 ? -1 : (int8_t)pv[s->maxVal[a]];
        for (int b = 0; b < x->nb; b++) {
            for (int v = 0; v < x->nv; v++) o->m2b[pa[a]][b][pv[v]] = s->m2b[a][b][v];
            for (int mb = 0; mb <= x->nb; mb++)
                for (int mv = 0; mv <= x->nv; mv++) o->m1b[pa[a]][b][mb][mv ? pv[mv - 1] + 1 : 0] = s->m1b[a][b][mb][mv];
        }
    }
    for (int b = 0; b < x->nb; b++)
        for (int v = 0; v < x->nv; v++) o->m2a[b][pv[v]] = s->m2a[b][v];
}
static size_t px_canon(void *c, const uint8_t *sb, size_t len, uint8_t *out) {
    const px_ctx *x = c;
    int pa[XA], pv[XV], first = 1;
    uint8_t img[sizeof(px_state) > sizeof(vt_state) ? sizeof(px_state) : sizeof(vt_state)];
    for (int i = 0; i < XA; i++) pa[i] = i;
    do {
        for (int i = 0; i < XV; i++) pv[i] = i;
        do {
            if (x->kind == 1) { vt_state s, o; memcpy(&s, sb, sizeof s); vt_apply(x, &s, pa, pv, &o); memcpy(img, &o, sizeof o); }
            else { px_state s, o; memcpy(&s, sb, sizeof s); px_apply(x, &s, pa, pv, &o); memcpy(img, &o, sizeof o); }
            if (first || memcmp(img, out, len) < 0) { memcpy(out, img, len); first = 0; }
        } while ((x->sym & 2) && next_perm(pv, x->nv));
    } while ((x->sym & 1) && next_perm(pa, x->na));
    return len;
}


/* ------------------------------------------------------------------------------------------ census (Voting)
 * The two alternative configurations of examples/Paxos/MCVoting.cfg:7-8 (MCVoting.tla:36-55), restated: over EVERY type-correct
 * state (TypeOK, Voting.tla:46-47) — out[0] = how many there are, out[1] = how many satisfy Inv (:160), out[2] = successors
 * generated from those by Next (TLC's witness multiplicities), out[3] = how many of these successors violate Inv (0 = Inv is
 * inductive on this model).  Pinned to oracle/tlaplus.py's evaluation of MCSpecI (tests/test_reference_text_paxos.py). */
typedef struct { or_emit em; const px_ctx *x; uint64_t n, bad; } census_emit;
static void census_on_emit(or_emit *em, const uint8_t *sb, size_t len, int action, unsigned flags) {
    census_emit *c = (census_emit *)em;
    vt_state t;
    (void)len; (void)action; (void)flags;
    memcpy(&t, sb, sizeof t);
    c->n++;
    if (!vt_Inv(c->x, &t)) c->bad++;
}
int oracle_voting_census(const int64_t *p, int np, uint64_t out[4]) {
    or_spec sp;
    if (or_spec_paxos(p, np, &sp)) return -1;
    px_ctx *x = sp.ctx;
    if (x->kind != 1) { or_set_error("census: Voting models only"); free(x); return -1; }
    const int vbits = x->nb * x->nv;
    const uint64_t per = ((uint64_t)1 << vbits) * (uint64_t)(x->nb + 1);
    uint64_t total = 1;
    for (int a = 0; a < x->na; a++) total *= per;
    census_emit ce;
    memset(&ce, 0, sizeof ce);
    ce.em.emit = census_on_emit;
    ce.x = x;
    x->prop = 0;
    memset(out, 0, 4 * sizeof out[0]);
    for (uint64_t k = 0; k < total; k++) {
        vt_state s;
        memset(&s, 0, sizeof s);
        for (int a = 0; a < XA; a++) s.maxBal[a] = -1;
        uint64_t r = k;
        for (int a = 0; a < x->na; a++) {
            const uint64_t d = r % per;
            r /= per;
            s.maxBal[a] = (int8_t)((int)(d % (uint64_t)(x->nb + 1)) - 1);
            const uint64_t m = d / (uint64_t)(x->nb + 1);
            for (int b = 0; b < x->nb; b++)
                for (int v = 0; v < x->nv; v++) s.votes[a][b][v] = (uint8_t)(m >> (b * x->nv + v) & 1);
        }
        out[0]++;
        if (!vt_Inv(x, &s)) continue;
        out[1]++;
        vt_succ(x, (const uint8_t *)&s, sizeof s, &ce.em);
    }
    out[2] = ce.n;
    out[3] = ce.bad;
    free(x);
    return 0;
}

static const char *VT_ACT[] = {"IncreaseMaxBal", "VoteFor"};
static const char *PX_ACT[] = {"Phase1a", "Phase2a", "Phase1b", "Phase2b"};
static int g_px_kind;
const char *or_paxos_action(int a) {
    if (a < 0) return "Initial predicate";
    if (g_px_kind == 1) return a < 2 ? VT_ACT[a] : "?";
    return a < 4 ? PX_ACT[a] : "?";
}

int or_spec_paxos(const int64_t *p, int np, or_spec *o) {
    if (np < 4) { or_set_error("paxos: params {kind, nAcceptor, nValue, nBallot, ...}"); return -1; }
    px_ctx *x = calloc(1, sizeof *x);
    x->kind = (int)p[0]; x->na = (int)p[1]; x->nv = (int)p[2]; x->nb = (int)p[3];
    x->inv_mask = np > 4 ? (int)p[4] : (x->kind ? 1 : 15);
    x->sym = np > 5 ? (int)p[5] & 3 : 0;
    x->prop = np > 6 ? (int)p[6] & 1 : 1;
    x->mut = np > 6 ? (int)p[6] >> 1 & 1 : 0;  /* negative control: Phase2a without its quorum conjunct (specs/paxos/MCPaxosBad.tla) */
    x->nq = np > 7 ? (int)p[7] : 0;
    if (x->kind < 0 || x->kind > 1 || x->na < 1 || x->na > XA || x->nv < 1 || x->nv > XV || x->nb < 1 || x->nb > XB || x->nq < 0 ||
        x->nq > XQ || (x->nq > 0 && np < 8 + x->nq)) { or_set_error("paxos: parameter out of range"); free(x); return -1; }
    if (x->nq == 0) {  /* all majorities of minimal size (MCVoting.tla:8 for three acceptors, MCPaxos.tla:9 for one) */
        const int need = x->na / 2 + 1;
        for (unsigned m = 1; m < (1u << x->na); m++)
            if (__builtin_popcount(m) == need) {
                if (x->nq == XQ) { or_set_error("paxos: too many quorums"); free(x); return -1; }
                x->quorum[x->nq++] = m;
            }
    } else {
        for (int q = 0; q < x->nq; q++) x->quorum[q] = (unsigned)p[8 + q];
    }
    g_px_kind = x->kind;
    memset(o, 0, sizeof *o);
    o->ctx = x;
    o->action_name = or_paxos_action;
    if (x->kind == 1) {
        o->name = "voting"; o->max_state_bytes = sizeof(vt_state);
        o->n_init = vt_n_init; o->init = vt_init; o->succ = vt_succ; o->invariant = vt_invariant; o->print = vt_print;
    } else {
        o->name = "paxos"; o->max_state_bytes = sizeof(px_state);
        o->n_init = px_n_init; o->init = px_init; o->succ = px_succ; o->invariant = px_invariant; o->print = px_print;
    }
    if (x->sym) o->canon = px_canon;
    return 0;
}
