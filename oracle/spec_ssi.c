/*
 * oracle/spec_ssi.c — CPU ORACLE (test infrastructure) for examples/serializableSnapshotIsolation.tla
 * (Cahill's serializable snapshot isolation; reference lines cited per function).
 *
 * Model wrapper assumed: specs/MCssi.tla + specs/MCssi.cfg of this repo (the reference only
 * describes Toolbox clicks, serializableSnapshotIsolation.tla:26-96): TxnId = {T1..Tn},
 * Key = {K1..Km} as model values, NoLock = NoLock (a model value replacing the unbounded CHOOSE
 * of :24, as textbookSnapshotIsolation.tla:1276-1277 documents), INIT Init / NEXT Next (the
 * WF_allvars conjunct of Spec :1005 is irrelevant to safety), deadlock checking on (:57).
 *
 * Every CHOOSE over a set of transactions is resolved in ascending transaction order
 * (T1 < T2 < ...): Commit's AbortOpSeq (:465-474).  The other CHOOSEs of the spec are over
 * singletons (:574 readVerSet, :790 holder, :851 path).
 *
 * params: {nTxn, nKey, invariant mask, find, textbook, sym}
 *   sym (cfg SYMMETRY; :38-44 make Key and TxnId symmetry sets): bit 0 = Permutations(TxnId), bit 1 = Permutations(Key);
 *     bit 2 = TLC's representative (the orbit member that was generated first is stored and expanded) instead of the
 *     canonical one — to measure whether the orbit COUNT depends on the representative (Commit's CHOOSE, :465-474);
 *     brute-force canonicalisation, see s_canonical
 *   textbook = 1 selects examples/textbookSnapshotIsolation.tla: the same model WITHOUT Cahill's three variables
 *     (its allvars :115): Commit never aborts the pivot (:325-355), Read (:365-378) and HelperWriteCanAcquireXLock
 *     (:383-386) do no conflict bookkeeping.  Its serializability invariants are then EXPECTED to fail (write skew).
 *   invariant mask (serializableSnapshotIsolation.tla:59-79): 1 WellFormedTransactionsInHistory,
 *     2 CorrectnessOfHoldingXLocks, 4 CorrectnessOfWaitingForXLock, 8 CorrectReadView,
 *     16 FirstCommitterWins, 32 CahillSerializable, 64 BernsteinSerializable  (TypeInv holds by
 *     construction of the representation)
 *   find (":81-96 EXPECTED to be violated"): 0 none; 1..6 = ~AtLeastNTxnsAbortedDueToReason(1, r)
 *     with r the index into AbortReasons (:189-194); 7 = ~AtLeastNTxnsAreWaitingForLocks(2).
 *     Reported as invariant index 7.
 */
#include "oracle_int.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ST 4   /* max transactions */
#define SK 3   /* max keys         */
#define SH 40  /* max history      */
#define NOLOCK 255

enum { OP_BEGIN = 0, OP_READ = 1, OP_WRITE = 2, OP_COMMIT = 3, OP_ABORT = 4 };
enum { R_VOLUNTARY = 0, R_FCW = 1, R_DEADLOCK = 2, R_COMMIT = 3, R_READ = 4, R_WRITE = 5 };
static const char *reason_txt[] = {"voluntary", "forced by First Committer Wins", "forced by deadlock-prevention",
                                   "in attempted commit, to preserve serializability",
                                   "in attempted read, to preserve serializability",
                                   "in attempted write, to preserve serializability"};
enum { SA_BEGIN, SA_COMMIT, SA_ABORT, SA_READ, SA_WRITE, SA_FINISH, SA_TERMINATED };

typedef struct { uint8_t op, txn, key, ver, reason; } Event;
typedef struct {
    int n;                    /* Len(history) */
    Event h[SH];
    uint8_t xlocks[ST];       /* holdingXLocks[t]: bit k            */
    uint8_t waiting[ST];      /* waitingForXLock[t]: key or NOLOCK  */
    uint8_t inC[ST], outC[ST];
    uint8_t siread[ST];       /* holdingSIREADlocks[t]: bit k       */
} SState;

typedef struct { int nt, nk, inv_mask, find, textbook, sym; } ssi_ctx;  /* textbook = 1: examples/textbookSnapshotIsolation.tla */

static size_t s_ser(const ssi_ctx *c, const SState *s, uint8_t *out) {
    uint8_t *p = out;
    *p++ = (uint8_t)s->n;
    for (int i = 0; i < s->n; i++) { *p++ = s->h[i].op; *p++ = s->h[i].txn; *p++ = s->h[i].key; *p++ = s->h[i].ver; *p++ = s->h[i].reason; }
    for (int t = 0; t < c->nt; t++) { *p++ = s->xlocks[t]; *p++ = s->waiting[t]; *p++ = s->inC[t]; *p++ = s->outC[t]; *p++ = s->siread[t]; }
    return (size_t)(p - out);
}
static void s_deser(const ssi_ctx *c, const uint8_t *p, SState *s) {
    memset(s, 0, sizeof *s);
    s->n = *p++;
    for (int i = 0; i < s->n; i++) { s->h[i].op = *p++; s->h[i].txn = *p++; s->h[i].key = *p++; s->h[i].ver = *p++; s->h[i].reason = *p++; }
    for (int t = 0; t < c->nt; t++) { s->xlocks[t] = *p++; s->waiting[t] = *p++; s->inC[t] = *p++; s->outC[t] = *p++; s->siread[t] = *p++; }
}

/* ---------------------------------------------------------------- history helpers (:271-330)
 * indices are 1-based like TLA+ sequences; 0 = absent (the spec's IndexOfOpInHistory uses -1) */
static int idx_op(const SState *s, int len, int op, int txn) {   /* first index <= len of [op, txn] for begin/commit/abort */
    for (int i = 0; i < len; i++) if (s->h[i].op == op && s->h[i].txn == txn) return i + 1;
    return 0;
}
static int started(const SState *s, int t) { return idx_op(s, s->n, OP_BEGIN, t) != 0; }            /* ActiveOrFinalizedTxns :274 */
static int committed_in(const SState *s, int len, int t) { return idx_op(s, len, OP_COMMIT, t) != 0; } /* CommittedTxns(prefix) :276 */
static int committed(const SState *s, int t) { return committed_in(s, s->n, t); }
static int aborted(const SState *s, int t) { return idx_op(s, s->n, OP_ABORT, t) != 0; }            /* AbortedTxns :277 */
static int finalized(const SState *s, int t) { return committed(s, t) || aborted(s, t); }           /* :278 */
static int active(const SState *s, int t) { return started(s, t) && !finalized(s, t); }              /* ActiveTxns :279 */
static int start_time(const SState *s, int t) { return idx_op(s, s->n, OP_BEGIN, t); }               /* StartTime :287 */
/* KeysThatTxnHasDoneOperationOn(history, txn, op) :289-291, as a key bitmask */
static unsigned keys_done(const SState *s, int t, int op) {
    unsigned m = 0;
    for (int i = 0; i < s->n; i++) if (s->h[i].op == op && s->h[i].txn == t) m |= 1u << s->h[i].key;
    return m;
}
static int idx_rw(const SState *s, int op, int t, int k) {  /* index of [op (read|write), txn t, key k] */
    for (int i = 0; i < s->n; i++) if (s->h[i].op == op && s->h[i].txn == t && s->h[i].key == k) return i + 1;
    return 0;
}
/* StartedAndCanDoPublicOperation(txn) :328-336 */
static int can_do(const SState *s, int t) { return active(s, t) && s->waiting[t] == NOLOCK; }

/* LatestCommittedVersionOfKeyWhenTxnBegan(txn, key) :351-361: -1 = {} */
static int latest_committed_version(const SState *s, int txn, int key) {
    int st = start_time(s, txn), ver = -1;
    for (int i = 0; i < st; i++)
        if (s->h[i].op == OP_WRITE && s->h[i].key == key && committed_in(s, st, s->h[i].txn)) ver = s->h[i].txn;
    return ver;
}
/* VersionThatWouldBeReadBy(txn, key) :366-378: -1 = {} */
static int version_read_by(const SState *s, int txn, int key) {
    if (s->xlocks[txn] >> key & 1) return txn;
    return latest_committed_version(s, txn, key);
}
/* VersionIDsOfKeyNewerThanReadByTxn(txn, key) :384-399 as a txn bitmask (ver = what txn would read) */
static unsigned newer_versions(const SState *s, int key, int ver) {
    unsigned m = 0;
    int seen = 0;
    for (int i = 0; i < s->n; i++) {
        if (s->h[i].op != OP_WRITE || s->h[i].key != key) continue;
        if (seen) m |= 1u << s->h[i].txn;
        else if (s->h[i].txn == ver) seen = 1;
    }
    return m;
}
/* WritersCommittedToKeySinceTxnBegan(txn, key) :339-346 as a txn bitmask */
static unsigned writers_committed_since(const ssi_ctx *c, const SState *s, int txn, int key) {
    int st = start_time(s, txn);
    unsigned m = 0;
    for (int t = 0; t < c->nt; t++) {
        int ci = idx_op(s, s->n, OP_COMMIT, t);
        if (ci >= st && ci != 0 && (keys_done(s, t, OP_WRITE) >> key & 1)) m |= 1u << t;
    }
    return m;
}

static void append(SState *s, int op, int txn, int key, int ver, int reason) {
    if (s->n >= SH) { fprintf(stderr, "oracle/ssi: history capacity exceeded\n"); abort(); }
    Event e = {(uint8_t)op, (uint8_t)txn, (uint8_t)key, (uint8_t)ver, (uint8_t)reason};
    s->h[s->n++] = e;
}
/* internalAbort(txn, reason) :406-416 */
static void internal_abort(SState *s, int txn, int reason) {
    append(s, OP_ABORT, txn, 0, 0, reason);
    s->xlocks[txn] = 0; s->waiting[txn] = NOLOCK; s->inC[txn] = 0; s->outC[txn] = 0; s->siread[txn] = 0;
}

/* SYMMETRY.  The spec's run-book declares Key and TxnId "symmetry sets" (serializableSnapshotIsolation.tla:38-44,
 * p-manual section 4.7.3 p.41): states that differ by a permutation of the model values are one state.  The oracle does
 * it by BRUTE FORCE: apply every permutation of TxnId (sym & 1) x every permutation of Key (sym & 2) to the
 * successor, serialise each image, keep the lexicographically smallest byte string; that representative is what
 * is stored and later expanded (so the CHOOSE order of Commit's AbortOpSeq :465-474 is the ascending order of the
 * representative's labels). */
static void s_permute(const ssi_ctx *c, const SState *s, const int *pt, const int *pk, SState *o) {
    memset(o, 0, sizeof *o);
    o->n = s->n;
    for (int i = 0; i < s->n; i++) {
        Event e = s->h[i];
        e.txn = (uint8_t)pt[e.txn];
        if (e.op == OP_READ || e.op == OP_WRITE) e.key = (uint8_t)pk[e.key];
        if (e.op == OP_READ) e.ver = (uint8_t)pt[e.ver];
        o->h[i] = e;
    }
    for (int t = 0; t < c->nt; t++) {
        unsigned xl = 0, sr = 0;
        for (int k = 0; k < c->nk; k++) {
            if (s->xlocks[t] >> k & 1) xl |= 1u << pk[k];
            if (s->siread[t] >> k & 1) sr |= 1u << pk[k];
        }
        const int u = pt[t];
        o->xlocks[u] = (uint8_t)xl; o->siread[u] = (uint8_t)sr;
        o->waiting[u] = s->waiting[t] == NOLOCK ? NOLOCK : (uint8_t)pk[s->waiting[t]];
        o->inC[u] = s->inC[t]; o->outC[u] = s->outC[t];
    }
}
static int next_perm(int *a, int n) {   /* lexicographic successor; 0 when a was the last one */
    int i = n - 2;
    while (i >= 0 && a[i] > a[i + 1]) i--;
    if (i < 0) return 0;
    int j = n - 1;
    while (a[j] < a[i]) j--;
    int x = a[i]; a[i] = a[j]; a[j] = x;
    for (int l = i + 1, r = n - 1; l < r; l++, r--) { x = a[l]; a[l] = a[r]; a[r] = x; }
    return 1;
}
static size_t s_canonical(const ssi_ctx *c, const SState *t, uint8_t *best) {
    uint8_t cand[512];
    size_t blen = s_ser(c, t, best);   /* the identity image */
    int pt[ST] = {0, 1, 2, 3};
    do {
        int pk[SK] = {0, 1, 2};
        do {
            SState img;
            s_permute(c, t, pt, pk, &img);
            size_t len = s_ser(c, &img, cand);
            if (memcmp(cand, best, len < blen ? len : blen) < 0) { memcpy(best, cand, len); blen = len; }  /* equal lengths: same n, nt */
        } while ((c->sym & 2) && next_perm(pk, c->nk));
    } while ((c->sym & 1) && next_perm(pt, c->nt));
    return blen;
}

typedef struct { const ssi_ctx *c; or_emit *em; uint8_t buf[512]; } sgen;
static void s_emit(sgen *g, const SState *t, int action) {
    /* sym & 4 = TLC's own scheme: the state is stored and expanded AS GENERATED (first member of its orbit that was met);
     * only the seen-set looks at the canonical form (ssi_canon below) */
    size_t len = (g->c->sym & 3) && !(g->c->sym & 4) ? s_canonical(g->c, t, g->buf) : s_ser(g->c, t, g->buf);
    g->em->emit(g->em, g->buf, len, action, 0);
}

/* findConcurrentSIREADlockOwners(txn, key) :657-685 */
static unsigned concurrent_siread_owners(const ssi_ctx *c, const SState *s, int txn, int key) {
    unsigned m = 0;
    int bt = idx_op(s, s->n, OP_BEGIN, txn);
    for (int t = 0; t < c->nt; t++) {
        if (t == txn || !(s->siread[t] >> key & 1)) continue;
        int ci = idx_op(s, s->n, OP_COMMIT, t);
        if (ci == 0 || ci > bt) m |= 1u << t;   /* not committed, or committed after txn began */
    }
    return m;
}
/* snapshotIsolationWriteAction(txn, key) :688-691 */
static void si_write(SState *s, int txn, int key) {
    append(s, OP_WRITE, txn, key, 0, 0);
    s->xlocks[txn] |= (uint8_t)(1u << key);
    s->waiting[txn] = NOLOCK;
}
/* HelperWriteCanAcquireXLock(txn, key) :700-771 — exactly one successor */
static void write_can_acquire(sgen *g, const SState *s, int txn, int key, int action) {
    const ssi_ctx *c = g->c;
    SState t = *s;
    unsigned owners = c->textbook ? 0u : concurrent_siread_owners(c, s, txn, key);
    if (owners) {
        int danger = 0;
        for (int o = 0; o < c->nt; o++) if ((owners >> o & 1) && (committed(s, o) || s->inC[o])) danger = 1;   /* :726-728: \/ */
        if (danger) internal_abort(&t, txn, R_WRITE);
        else {
            si_write(&t, txn, key);
            for (int o = 0; o < c->nt; o++) if (owners >> o & 1) t.outC[o] = 1;
            t.inC[txn] = 1;
        }
    } else {
        si_write(&t, txn, key);
    }
    s_emit(g, &t, action);
}
/* HelperWriteConflictsWithXLock(txn, key) :774-880 — one successor, or one per member of the cycle */
static void write_conflicts(sgen *g, const SState *s, int txn, int key) {
    const ssi_ctx *c = g->c;
    int holder[SK];   /* xlockIsHeldBy :787-791: active holder or -1 (NoLock) */
    for (int k = 0; k < c->nk; k++) {
        holder[k] = -1;
        for (int t = 0; t < c->nt; t++) if (active(s, t) && (s->xlocks[t] >> k & 1)) { holder[k] = t; break; }
    }
    /* extendPath(<<txn>>) :823-836 over proposedWaitingForXLock = [waitingForXLock EXCEPT ![txn] = key] */
    int path[ST + 1], plen = 0, cycle = 0;
    path[plen++] = txn;
    for (;;) {
        int from = path[plen - 1];
        int want = from == txn ? key : s->waiting[from];
        if (!active(s, from) || want == NOLOCK || holder[want] < 0 || !active(s, holder[want])) break;   /* dead end: no cycle */
        int to = holder[want];
        if (to == txn) { cycle = 1; break; }
        if (plen > c->nt) { fprintf(stderr, "oracle/ssi: waits-for graph already cyclic\n"); abort(); }
        path[plen++] = to;
    }
    if (!cycle) {
        SState t = *s;
        t.waiting[txn] = (uint8_t)key;
        s_emit(g, &t, SA_WRITE);
        return;
    }
    /* \E to_abort \in Range(path) :851 — in ascending transaction order */
    for (int a = 0; a < c->nt; a++) {
        int inpath = 0;
        for (int q = 0; q < plen; q++) inpath |= path[q] == a;
        if (!inpath) continue;
        SState t = *s;
        append(&t, OP_ABORT, a, 0, 0, R_DEADLOCK);
        t.xlocks[a] = 0;
        if (a != txn) { t.waiting[txn] = (uint8_t)key; t.waiting[a] = NOLOCK; }
        t.inC[a] = 0; t.outC[a] = 0; t.siread[a] = 0;
        s_emit(g, &t, SA_WRITE);
    }
}

/* Next :971-996 */
static void ssi_succ(void *ctx, const uint8_t *sb, size_t len, or_emit *em) {
    const ssi_ctx *c = ctx; (void)len;
    SState s;
    s_deser(c, sb, &s);
    sgen g = {c, em, {0}};
    unsigned anylocked = 0;   /* KeysCurrentlyXLockedByAnyTxn :326 */
    for (int t = 0; t < c->nt; t++) anylocked |= s.xlocks[t];
    for (int txn = 0; txn < c->nt; txn++) {
        /* Begin(txn) :423-426 */
        if (!started(&s, txn)) { SState t = s; append(&t, OP_BEGIN, txn, 0, 0, 0); s_emit(&g, &t, SA_BEGIN); }
        /* Commit(txn) :429-491 */
        if (can_do(&s, txn)) {
            SState t = s;
            if (!c->textbook && s.inC[txn] && s.outC[txn]) internal_abort(&t, txn, R_COMMIT);
            else {
                append(&t, OP_COMMIT, txn, 0, 0, 0);
                unsigned losers = 0;   /* LoserTxns :461-462 */
                for (int b = 0; b < c->nt; b++) if (s.waiting[b] != NOLOCK && (s.xlocks[txn] >> s.waiting[b] & 1)) losers |= 1u << b;
                for (int b = 0; b < c->nt; b++) if (losers >> b & 1) append(&t, OP_ABORT, b, 0, 0, R_FCW);   /* AbortOpSeq :465-474 */
                for (int b = 0; b < c->nt; b++) {
                    if (b == txn || (losers >> b & 1)) t.xlocks[b] = 0;
                    if (losers >> b & 1) { t.waiting[b] = NOLOCK; t.inC[b] = 0; t.outC[b] = 0; t.siread[b] = 0; }
                }
            }
            s_emit(&g, &t, SA_COMMIT);
        }
        /* ChooseToAbort(txn) :494-496 */
        if (can_do(&s, txn)) { SState t = s; internal_abort(&t, txn, R_VOLUNTARY); s_emit(&g, &t, SA_ABORT); }
        for (int key = 0; key < c->nk; key++) {
            /* Read(txn, key) :525-626 */
            if (can_do(&s, txn) && !(keys_done(&s, txn, OP_READ) >> key & 1)) {
                int ver = version_read_by(&s, txn, key);
                if (ver >= 0) {
                    unsigned newer = newer_versions(&s, key, ver);
                    int danger = 0;
                    for (int x = 0; x < c->nt; x++) if ((newer >> x & 1) && committed(&s, x) && s.outC[x]) danger = 1;
                    SState t = s;
                    if (c->textbook) append(&t, OP_READ, txn, key, ver, 0);   /* textbookSnapshotIsolation.tla:365-378 */
                    else if (danger) internal_abort(&t, txn, R_READ);
                    else {
                        append(&t, OP_READ, txn, key, ver, 0);
                        t.siread[txn] |= (uint8_t)(1u << key);
                        unsigned lockers = 0;
                        for (int x = 0; x < c->nt; x++) if (x != txn && (s.xlocks[x] >> key & 1)) lockers |= 1u << x;
                        for (int x = 0; x < c->nt; x++) if ((newer | lockers) >> x & 1) t.inC[x] = 1;
                        if (newer || lockers) t.outC[txn] = 1;
                    }
                    s_emit(&g, &t, SA_READ);
                }
            }
            /* StartWriteMayBlock(txn, key) :883-911 */
            if (can_do(&s, txn) && !(s.xlocks[txn] >> key & 1)) {
                if (writers_committed_since(c, &s, txn, key)) {
                    SState t = s;   /* :899-905: like internalAbort but waitingForXLock is UNCHANGED (it is NoLock) */
                    append(&t, OP_ABORT, txn, 0, 0, R_FCW);
                    t.xlocks[txn] = 0; t.inC[txn] = 0; t.outC[txn] = 0; t.siread[txn] = 0;
                    s_emit(&g, &t, SA_WRITE);
                } else if (anylocked >> key & 1) {
                    write_conflicts(&g, &s, txn, key);
                } else {
                    write_can_acquire(&g, &s, txn, key, SA_WRITE);
                }
            }
        }
        /* FinishBlockedWrite(txn) :923-927 */
        if (s.waiting[txn] != NOLOCK && !(anylocked >> s.waiting[txn] & 1)) write_can_acquire(&g, &s, txn, s.waiting[txn], SA_FINISH);
    }
    /* LegitimateTermination /\ UNCHANGED allvars :963,996 */
    int all = 1;
    for (int t = 0; t < c->nt; t++) all &= finalized(&s, t);
    if (all) s_emit(&g, &s, SA_TERMINATED);
}

static int ssi_n_init(void *c) { (void)c; return 1; }
/* Init :938-943 */
static size_t ssi_init(void *ctx, int k, uint8_t *out) {
    const ssi_ctx *c = ctx; (void)k;
    SState s;
    memset(&s, 0, sizeof s);
    for (int t = 0; t < c->nt; t++) s.waiting[t] = NOLOCK;
    return s_ser(c, &s, out);
}

/* ---------------------------------------------------------------- invariants */
/* FindAllNodesInAnyCycle(edges) /= {}  (:1040-1060): adj[a] bit b = edge a -> b */
static int has_cycle(const unsigned *adj, int n) {
    unsigned reach[ST];
    for (int a = 0; a < n; a++) reach[a] = adj[a];
    for (int k = 0; k < n; k++) for (int a = 0; a < n; a++) if (reach[a] >> k & 1) reach[a] |= reach[k];
    for (int a = 0; a < n; a++) if (reach[a] >> a & 1) return 1;
    return 0;
}
/* AreConcurrent(h, t1, t2) :1118-1133 (with the spec's own duplicated iT1b test) */
static int concurrent(const SState *s, int t1, int t2) {
    int b1 = idx_op(s, s->n, OP_BEGIN, t1), c1 = idx_op(s, s->n, OP_COMMIT, t1);
    int b2 = idx_op(s, s->n, OP_BEGIN, t2), c2 = idx_op(s, s->n, OP_COMMIT, t2);
    if (!b1) return 0;
    /* with "absent" = -1 in the spec: iT1b < iT2b is evaluated on -1 for an unstarted t2 */
    int B1 = b1, B2 = b2 ? b2 : -1, C1 = c1 ? c1 : -1, C2 = c2 ? c2 : -1;
    if (B1 < B2) return C1 == -1 || C1 > B2;
    return C2 == -1 || C2 > B1;
}
static int inv_wellformed(const ssi_ctx *c, const SState *s) {   /* :1146-1179 */
    for (int t = 0; t < c->nt; t++) {
        int cnt = 0, nb = 0, fin = 0, rk[SK] = {0}, wk[SK] = {0};
        for (int i = 0; i < s->n; i++) {
            if (s->h[i].txn != t) continue;
            cnt++;
            if (fin) return 0;                                  /* something after commit/abort */
            if (s->h[i].op == OP_BEGIN) { nb++; if (cnt != 1) return 0; }
            else if (cnt == 1) return 0;                        /* first op is not begin */
            if (s->h[i].op == OP_COMMIT || s->h[i].op == OP_ABORT) fin = 1;
            if (s->h[i].op == OP_READ && ++rk[s->h[i].key] > 1) return 0;
            if (s->h[i].op == OP_WRITE && ++wk[s->h[i].key] > 1) return 0;
        }
        if (cnt && nb != 1) return 0;
    }
    return 1;
}
static int inv_holding_xlocks(const ssi_ctx *c, const SState *s) {   /* :1302-1321 */
    for (int k = 0; k < c->nk; k++) { int n = 0; for (int t = 0; t < c->nt; t++) n += s->xlocks[t] >> k & 1; if (n > 1) return 0; }
    for (int t = 0; t < c->nt; t++) {
        if (active(s, t)) { if (s->xlocks[t] != keys_done(s, t, OP_WRITE)) return 0; }
        else if (s->xlocks[t]) return 0;
    }
    return 1;
}
static int inv_waiting(const ssi_ctx *c, const SState *s) {   /* :1324-1330 */
    for (int t = 0; t < c->nt; t++) if (s->waiting[t] != NOLOCK && !active(s, t)) return 0;
    return 1;
}
static int inv_read_view(const ssi_ctx *c, const SState *s) {   /* CorrectReadView :1215-1268 */
    for (int txn = 0; txn < c->nt; txn++) {
        int itxnb = idx_op(s, s->n, OP_BEGIN, txn);
        for (int i = 0; i < s->n; i++) {
            if (s->h[i].op != OP_READ || s->h[i].txn != txn) continue;
            int key = s->h[i].key, ver = s->h[i].ver;
            if (ver != txn) {   /* only committed reads */
                int irfc = idx_op(s, s->n, OP_COMMIT, ver);
                if (!irfc || !(irfc < itxnb)) return 0;
            }
            /* only up-to-date reads: no committed write of key between the write read and txn's begin */
            int iwkv = idx_rw(s, OP_WRITE, ver, key);
            for (int j = iwkv; j < itxnb; j++)   /* SubSeq(history, iwkv+1, itxnb), 0-based j = index-1 */
                if (s->h[j].op == OP_WRITE && s->h[j].key == key && committed_in(s, itxnb, s->h[j].txn)) return 0;
            /* keys both read and written by txn */
            int iw = idx_rw(s, OP_WRITE, txn, key);
            if (iw) {
                int ir = i + 1;
                if (ir < iw) { if (ver != latest_committed_version(s, txn, key)) return 0; }
                else if (ver != txn) return 0;
            }
        }
    }
    return 1;
}
static int inv_fcw(const ssi_ctx *c, const SState *s) {   /* FirstCommitterWins :1271-1278 */
    for (int a = 0; a < c->nt; a++) for (int b = 0; b < c->nt; b++)
        if (a != b && committed(s, a) && committed(s, b) && concurrent(s, a, b) && (keys_done(s, a, OP_WRITE) & keys_done(s, b, OP_WRITE))) return 0;
    return 1;
}
/* index in ch (committed projection) — order-isomorphic to the index in h for committed transactions */
static int inv_cahill(const ssi_ctx *c, const SState *s) {   /* CahillSerializable :1379-1446 */
    unsigned adj[ST] = {0};
    for (int a = 0; a < c->nt; a++) for (int b = 0; b < c->nt; b++) {
        if (a == b || !committed(s, a) || !committed(s, b)) continue;
        for (int x = 0; x < c->nk; x++) {
            int aw = idx_rw(s, OP_WRITE, a, x), bw = idx_rw(s, OP_WRITE, b, x);
            int ww = aw && bw && aw < bw;
            int wr = aw && (keys_done(s, b, OP_READ) >> x & 1) && idx_op(s, s->n, OP_COMMIT, a) < idx_op(s, s->n, OP_BEGIN, b);
            int rw = (keys_done(s, a, OP_READ) >> x & 1) && bw && idx_op(s, s->n, OP_BEGIN, a) < idx_op(s, s->n, OP_COMMIT, b);
            if (ww || wr || rw) adj[a] |= 1u << b;
        }
    }
    return !has_cycle(adj, c->nt);
}
static int read_event(const SState *s, int t, int x, int ver) {   /* index of [read, t, x, ver] or 0 */
    for (int i = 0; i < s->n; i++) if (s->h[i].op == OP_READ && s->h[i].txn == t && s->h[i].key == x && s->h[i].ver == ver) return i + 1;
    return 0;
}
static int inv_bernstein(const ssi_ctx *c, const SState *s) {   /* BernsteinSerializable :1505-1556 */
    unsigned adj[ST] = {0};
    int n = c->nt;
    for (int w = 0; w < n; w++) for (int r = 0; r < n; r++) {   /* BernsteinSG: writer -> reader */
        if (w == r || !committed(s, w) || !committed(s, r)) continue;
        for (int i = 0; i < s->n; i++) if (s->h[i].op == OP_READ && s->h[i].txn == r && s->h[i].ver == w) adj[w] |= 1u << r;
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) for (int k = 0; k < n; k++) {   /* version order edges */
        if (i == j || i == k || j == k || !committed(s, i) || !committed(s, j) || !committed(s, k)) continue;
        for (int x = 0; x < c->nk; x++) {
            if (!read_event(s, k, x, j)) continue;            /* rk[xj] in C(H) */
            int xi = idx_rw(s, OP_WRITE, i, x), xj = idx_rw(s, OP_WRITE, j, x);
            if (!xi || !xj) continue;
            if (xi < xj) adj[i] |= 1u << j; else adj[k] |= 1u << i;
        }
    }
    return !has_cycle(adj, n);
}
static int ssi_invariant(void *ctx, const uint8_t *sb, size_t len) {
    const ssi_ctx *c = ctx; (void)len;
    SState s;
    s_deser(c, sb, &s);
    if ((c->inv_mask & 1) && !inv_wellformed(c, &s)) return 0;
    if ((c->inv_mask & 2) && !inv_holding_xlocks(c, &s)) return 1;
    if ((c->inv_mask & 4) && !inv_waiting(c, &s)) return 2;
    if ((c->inv_mask & 8) && !inv_read_view(c, &s)) return 3;
    if ((c->inv_mask & 16) && !inv_fcw(c, &s)) return 4;
    if ((c->inv_mask & 32) && !inv_cahill(c, &s)) return 5;
    if ((c->inv_mask & 64) && !inv_bernstein(c, &s)) return 6;
    if (c->find >= 1 && c->find <= 6) {   /* ~AtLeastNTxnsAbortedDueToReason(1, reason) :1579-1582 */
        for (int i = 0; i < s.n; i++) if (s.h[i].op == OP_ABORT && s.h[i].reason == c->find - 1) return 7;
    } else if (c->find == 7) {            /* ~AtLeastNTxnsAreWaitingForLocks(2) :1578 */
        int n = 0;
        for (int t = 0; t < c->nt; t++) n += s.waiting[t] != NOLOCK;
        if (n >= 2) return 7;
    }
    return -1;
}

/* ---------------------------------------------------------------- printing */
static size_t ssi_print(void *ctx, const uint8_t *sb, size_t len, char *buf, size_t cap) {
    const ssi_ctx *c = ctx; (void)len;
    SState s;
    s_deser(c, sb, &s);
    size_t k = 0;
#define P(...) do { if (k < cap) { int w_ = snprintf(buf + k, cap - k, __VA_ARGS__); if (w_ > 0) k += (size_t)w_; if (k > cap) k = cap; } } while (0)
    P("/\\ history = <<");
    for (int i = 0; i < s.n; i++) {
        const Event *e = &s.h[i];
        if (i) P(", ");
        switch (e->op) {
        case OP_BEGIN: P("[op |-> \"begin\", txnid |-> T%d]", e->txn + 1); break;
        case OP_COMMIT: P("[op |-> \"commit\", txnid |-> T%d]", e->txn + 1); break;
        case OP_ABORT: P("[op |-> \"abort\", reason |-> \"%s\", txnid |-> T%d]", reason_txt[e->reason], e->txn + 1); break;
        case OP_READ: P("[key |-> K%d, op |-> \"read\", txnid |-> T%d, ver |-> T%d]", e->key + 1, e->txn + 1, e->ver + 1); break;
        default: P("[key |-> K%d, op |-> \"write\", txnid |-> T%d]", e->key + 1, e->txn + 1); break;
        }
    }
    P(">>");
#define KEYSET(m) do { P("{"); int f_ = 1; for (int q = 0; q < c->nk; q++) if ((m) >> q & 1) { P("%sK%d", f_ ? "" : ", ", q + 1); f_ = 0; } P("}"); } while (0)
#define PER_TXN(title, expr) do { P("\n/\\ " title " = ("); for (int t = 0; t < c->nt; t++) { P("%sT%d :> ", t ? " @@ " : "", t + 1); expr; } P(")"); } while (0)
    PER_TXN("holdingXLocks", KEYSET(s.xlocks[t]));
    PER_TXN("waitingForXLock", if (s.waiting[t] == NOLOCK) P("NoLock"); else P("K%d", s.waiting[t] + 1));
    PER_TXN("inConflict", P("%s", s.inC[t] ? "TRUE" : "FALSE"));
    PER_TXN("outConflict", P("%s", s.outC[t] ? "TRUE" : "FALSE"));
    PER_TXN("holdingSIREADlocks", KEYSET(s.siread[t]));
#undef P
#undef KEYSET
#undef PER_TXN
    return k;
}
/* ---------------------------------------------------------------- the spec's own unit tests
 * UnitTests_FindAllNodesInAnyCycle (:1068-1077, 9 cases) and
 * UnitTest_WellFormedTransactionsInHistory (:1184-1205, 4 positive + 6 negative cases),
 * which TLC evaluates as constant expressions.  Returns the number of failing cases. */
static unsigned cycle_nodes(const unsigned *adj, int n) {
    unsigned reach[ST], out = 0;
    for (int a = 0; a < n; a++) reach[a] = adj[a];
    for (int k = 0; k < n; k++) for (int a = 0; a < n; a++) if (reach[a] >> k & 1) reach[a] |= reach[k];
    for (int a = 0; a < n; a++) if (reach[a] >> a & 1) out |= 1u << a;
    return out;
}
int oracle_ssi_unit_tests(void) {
    int fails = 0;
    /* nodes a,b,c,d = 0..3; each case: list of edges, expected node set */
    static const struct { int ne; int e[4][2]; unsigned expect; } cyc[9] = {
        {0, {{0}}, 0}, {1, {{0, 1}}, 0}, {3, {{0, 1}, {1, 2}, {2, 3}}, 0}, {1, {{0, 0}}, 1}, {2, {{0, 1}, {1, 0}}, 3},
        {4, {{0, 1}, {1, 2}, {2, 3}, {3, 0}}, 15}, {2, {{0, 0}, {1, 1}}, 3}, {4, {{0, 3}, {3, 1}, {2, 3}, {3, 2}}, 12},
        {4, {{0, 1}, {1, 0}, {2, 2}, {3, 2}}, 7}};
    for (int k = 0; k < 9; k++) {
        unsigned adj[ST] = {0};
        for (int q = 0; q < cyc[k].ne; q++) adj[cyc[k].e[q][0]] |= 1u << cyc[k].e[q][1];
        if (cycle_nodes(adj, 4) != cyc[k].expect) fails++;
        if (has_cycle(adj, 4) != (cyc[k].expect != 0)) fails++;
    }
    /* histories over T_1 = 0 (T_2 = 1 only as a version), K_X = 0, K_Y = 1 */
    static const struct { int n; Event h[4]; int expect; } wf[10] = {
        {1, {{OP_BEGIN, 0, 0, 0, 0}}, 1},
        {2, {{OP_BEGIN, 0, 0, 0, 0}, {OP_COMMIT, 0, 0, 0, 0}}, 1},
        {4, {{OP_BEGIN, 0, 0, 0, 0}, {OP_READ, 0, 0, 1, 0}, {OP_WRITE, 0, 1, 0, 0}, {OP_COMMIT, 0, 0, 0, 0}}, 1},
        {4, {{OP_BEGIN, 0, 0, 0, 0}, {OP_READ, 0, 0, 1, 0}, {OP_WRITE, 0, 0, 0, 0}, {OP_ABORT, 0, 0, 0, R_VOLUNTARY}}, 1},
        {2, {{OP_WRITE, 0, 0, 0, 0}, {OP_BEGIN, 0, 0, 0, 0}}, 0},
        {3, {{OP_BEGIN, 0, 0, 0, 0}, {OP_BEGIN, 0, 0, 0, 0}, {OP_WRITE, 0, 0, 0, 0}}, 0},
        {3, {{OP_BEGIN, 0, 0, 0, 0}, {OP_COMMIT, 0, 0, 0, 0}, {OP_WRITE, 0, 0, 0, 0}}, 0},
        {3, {{OP_BEGIN, 0, 0, 0, 0}, {OP_ABORT, 0, 0, 0, R_VOLUNTARY}, {OP_WRITE, 0, 0, 0, 0}}, 0},
        {3, {{OP_BEGIN, 0, 0, 0, 0}, {OP_WRITE, 0, 0, 0, 0}, {OP_WRITE, 0, 0, 0, 0}}, 0},
        {3, {{OP_BEGIN, 0, 0, 0, 0}, {OP_READ, 0, 0, 1, 0}, {OP_READ, 0, 0, 1, 0}}, 0}};
    ssi_ctx c = {2, 2, 127, 0, 0, 0};
    for (int k = 0; k < 10; k++) {
        SState s;
        memset(&s, 0, sizeof s);
        s.n = wf[k].n;
        for (int i = 0; i < s.n; i++) s.h[i] = wf[k].h[i];
        if (inv_wellformed(&c, &s) != wf[k].expect) fails++;
    }
    /* examples/textbookSnapshotIsolation.tla:1231-1263 — UnitTests_ReadOnlyAnomaly: the history of Fekete's paper
     *   R2(X0,0) R2(Y0,0) R1(Y0,0) W1(Y1,20) C1 R3(X0,0) R3(Y1,20) C3 W2(X2,-11) C2
     * as the spec encodes it (T_0 = 0 creates the keys; K_X = 0, K_Y = 1; R1(Y0,0) is written as a "write" there,
     * :1251).  ReadOnlyAnomaly(h) (:1216-1228): h is not serializable, and there is a transaction with reads and
     * no writes whose removal makes it serializable (that transaction is T_3). */
    {
        static const Event fek[19] = {
            {OP_BEGIN, 0, 0, 0, 0}, {OP_WRITE, 0, 0, 0, 0}, {OP_WRITE, 0, 1, 0, 0}, {OP_COMMIT, 0, 0, 0, 0},
            {OP_BEGIN, 2, 0, 0, 0}, {OP_READ, 2, 0, 0, 0}, {OP_READ, 2, 1, 0, 0},
            {OP_BEGIN, 1, 0, 0, 0}, {OP_WRITE, 1, 1, 0, 0}, {OP_WRITE, 1, 1, 0, 0}, {OP_COMMIT, 1, 0, 0, 0},
            {OP_BEGIN, 3, 0, 0, 0}, {OP_READ, 3, 0, 0, 0}, {OP_READ, 3, 1, 1, 0}, {OP_COMMIT, 3, 0, 0, 0},
            {OP_WRITE, 2, 0, 0, 0}, {OP_COMMIT, 2, 0, 0, 0}};
        ssi_ctx c4 = {4, 2, 127, 0, 1, 0};
        SState s;
        memset(&s, 0, sizeof s);
        s.n = 17;
        for (int i = 0; i < 17; i++) s.h[i] = fek[i];
        if (inv_cahill(&c4, &s) != 0) fails++;     /* ~CahillSerializable(h) */
        if (inv_bernstein(&c4, &s) != 0) fails++;  /* both formulations agree (:84-89) */
        int found = 0;
        for (int t = 0; t < 4; t++) {
            int reads = 0, writes = 0;
            for (int i = 0; i < 17; i++)
                if (fek[i].txn == t) { reads += fek[i].op == OP_READ; writes += fek[i].op == OP_WRITE; }
            if (!reads || writes) continue;
            SState r;
            memset(&r, 0, sizeof r);
            for (int i = 0; i < 17; i++) if (fek[i].txn != t) r.h[r.n++] = fek[i];
            if (inv_cahill(&c4, &r) && inv_bernstein(&c4, &r)) found = t + 1;   /* HistoryWithoutTxn(h, txn) is serializable */
        }
        if (found != 3 + 1) fails++;                /* the read-only transaction is T_3 */
    }
    return fails;
}

const char *or_ssi_action(int a) {
    static const char *nm[] = {"Begin", "Commit", "ChooseToAbort", "Read", "StartWriteMayBlock", "FinishBlockedWrite", "Terminated"};
    return a >= 0 && a < 7 ? nm[a] : "?";
}
static void ssi_stats(void *ctx, const uint8_t *sb, size_t len, uint64_t *mx) {
    (void)ctx; (void)len;
    if (sb[0] > mx[0]) mx[0] = sb[0];   /* longest history */
}
/* first-met mode (sym & 4): the seen-set's key of a state = its orbit's canonical form */
static size_t ssi_canon(void *ctx, const uint8_t *sb, size_t len, uint8_t *out) {
    const ssi_ctx *c = ctx; (void)len;
    SState s;
    s_deser(c, sb, &s);
    return s_canonical(c, &s, out);
}
int or_spec_ssi(const int64_t *p, int np, or_spec *o) {
    if (np < 2 || p[0] < 1 || p[0] > ST || p[1] < 1 || p[1] > SK) { or_set_error("ssi: need {nTxn <= 4, nKey <= 3[, invmask, find]}"); return -1; }
    ssi_ctx *c = calloc(1, sizeof *c);
    c->nt = (int)p[0]; c->nk = (int)p[1];
    c->inv_mask = np > 2 ? (int)p[2] : 127;
    c->find = np > 3 ? (int)p[3] : 0;
    c->textbook = np > 4 ? (int)p[4] : 0;
    c->sym = np > 5 ? (int)p[5] & 7 : 0;
    if (!(c->sym & 3)) c->sym = 0;
    o->name = "ssi"; o->ctx = c; o->max_state_bytes = 512;
    o->n_init = ssi_n_init; o->init = ssi_init; o->succ = ssi_succ; o->invariant = ssi_invariant; o->print = ssi_print;
    o->action_name = or_ssi_action; o->stats = ssi_stats;
    if (c->sym & 4) o->canon = ssi_canon;
    return 0;
}
