/*
 * oracle/spec_raft.c — CPU ORACLE (test infrastructure) for examples/raft.tla.
 *
 * A direct restatement of the reference spec over an unpacked C struct, one C function per
 * TLA+ action, each citing the raft.tla lines it follows.  The model wrapper it assumes is
 * specs/MCraft.tla + specs/MCraft.cfg of this repo (the reference has no cfg for raft,
 * SURVEY.md §0 item 4):
 *     StateConstraint == /\ \A i \in Server : currentTerm[i] <= MaxTerm
 *                        /\ \A i \in Server : Len(log[i]) <= MaxLogLen
 *                        /\ InFlight <= MaxMsgs          (sum of the bag's counts)
 *     NoTwoLeaders       == ~MoreThanOneLeader           (raft.tla:506-507)
 *     CommittedLogStable == ~committedLogDecrease        (raft.tla:74,302)
 *
 * Semantics that change counts and are deliberately kept:
 *   0. raft.tla:392-393 assigns commitIndex' and raft.tla:402 then says UNCHANGED logVars
 *      (which contains commitIndex, raft.tla:75): the "already done" branch is enabled only
 *      when m.mcommitIndex = commitIndex[i].  (params[6] = 1 selects the naive lowering as a
 *      negative control.)
 *   1. WithoutMessage keeps a zero-count key (raft.tla:125-129); WithMessage saturates at 2
 *      (raft.tla:117-121): the bag is a map msg -> {0,1,2} with a monotone key set.
 *   2. UpdateTerm (raft.tla:434-440), "return to follower" (raft.tla:374-378) and the
 *      conflict/append branches (raft.tla:410,416) do not consume the message.
 *   3. RequestVote(i,j) has no i /= j guard (raft.tla:209-217).
 *   4. voterLog[i] @@ (j :> m.mlog) keeps an existing entry (raft.tla:343-344, TLC.tla:11-12).
 *   5. committedLog' is <<>> unless newCommitIndex > 1 (raft.tla:296-300).
 *   6. committedLogDecrease' uses lazy left-to-right \/ (raft.tla:302-303).
 *   7. allLogs' uses the UNPRIMED logs (raft.tla:493); votesSent is constant FALSE.
 */
#include "oracle_int.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RN 5     /* max servers                         */
#define RL 7     /* max log length                      */
#define RM 160   /* max |DOMAIN messages|               */
#define RE 16    /* max |elections|                     */
#define RA 96    /* max |allLogs|                       */

enum { FOLLOWER = 0, CANDIDATE = 1, LEADER = 2 };
enum { RVREQ = 0, RVRESP = 1, AEREQ = 2, AERESP = 3 };
enum { A_RESTART, A_TIMEOUT, A_REQUESTVOTE, A_BECOMELEADER, A_CLIENTREQUEST, A_ADVANCECOMMIT,
       A_APPENDENTRIES, A_RECEIVE, A_DUPLICATE, A_DROP, A_NACT };

typedef struct { uint8_t term, value; } Entry;
typedef struct { uint8_t len; Entry e[RL]; } Log;

typedef struct {            /* unused fields are always zero => memcmp is equality */
    uint8_t mtype, mterm, msource, mdest;
    uint8_t mlastLogTerm, mlastLogIndex;                 /* RequestVoteRequest    */
    uint8_t mvoteGranted;                                /* RequestVoteResponse   */
    uint8_t mprevLogIndex, mprevLogTerm, nentries;       /* AppendEntriesRequest  */
    Entry mentry;
    uint8_t mcommitIndex;
    uint8_t msuccess, mmatchIndex;                       /* AppendEntriesResponse */
    Log mlog;                                            /* RVResp and AEReq      */
} Msg;

typedef struct {
    uint8_t eterm, eleader, evotes, evoterDom;
    Log elog;
    Log evoterLog[RN];
} Election;

typedef struct {
    /* raft.tla:36-103, in declaration order */
    int nm; Msg msg[RM]; uint8_t cnt[RM];     /* messages : [Message -> 0..2], sorted by key */
    int ne; Election el[RE];                  /* elections, sorted                            */
    int na; Log al[RA];                       /* allLogs, sorted                              */
    uint8_t currentTerm[RN], state[RN], votedFor[RN]; /* votedFor: 0 = Nil, j+1 = server j   */
    uint8_t clientRequests;
    Log log[RN];
    uint8_t commitIndex[RN];
    Log committedLog;
    uint8_t committedLogDecrease;
    /* votesSent[i] is FALSE in every reachable state (raft.tla:162,188,203) */
    uint8_t votesGranted[RN];                 /* bit j = server j                             */
    uint8_t voterDom[RN]; Log voterLog[RN][RN];
    uint8_t nextIndex[RN][RN], matchIndex[RN][RN];
} State;

typedef struct {
    int n, max_client_requests, max_term, max_log_len, max_msgs, inv_mask, naive_commit, max_keys;
} raft_ctx;

/* ---------------------------------------------------------------- (de)serialisation */
static uint8_t *put_log(uint8_t *p, const Log *l) {
    *p++ = l->len;
    for (int i = 0; i < l->len; i++) { *p++ = l->e[i].term; *p++ = l->e[i].value; }
    return p;
}
static const uint8_t *get_log(const uint8_t *p, Log *l) {
    memset(l, 0, sizeof *l);
    l->len = *p++;
    for (int i = 0; i < l->len; i++) { l->e[i].term = *p++; l->e[i].value = *p++; }
    return p;
}
static size_t ser(const raft_ctx *c, const State *s, uint8_t *out) {
    uint8_t *p = out;
    int n = c->n;
    *p++ = (uint8_t)s->nm;
    for (int k = 0; k < s->nm; k++) {
        const Msg *m = &s->msg[k];
        *p++ = m->mtype; *p++ = m->mterm; *p++ = m->msource; *p++ = m->mdest; *p++ = s->cnt[k];
        switch (m->mtype) {
        case RVREQ: *p++ = m->mlastLogTerm; *p++ = m->mlastLogIndex; break;
        case RVRESP: *p++ = m->mvoteGranted; p = put_log(p, &m->mlog); break;
        case AEREQ:
            *p++ = m->mprevLogIndex; *p++ = m->mprevLogTerm; *p++ = m->nentries;
            *p++ = m->mentry.term; *p++ = m->mentry.value; *p++ = m->mcommitIndex;
            p = put_log(p, &m->mlog);
            break;
        default: *p++ = m->msuccess; *p++ = m->mmatchIndex; break;
        }
    }
    *p++ = (uint8_t)s->ne;
    for (int k = 0; k < s->ne; k++) {
        const Election *e = &s->el[k];
        *p++ = e->eterm; *p++ = e->eleader; *p++ = e->evotes; *p++ = e->evoterDom;
        p = put_log(p, &e->elog);
        for (int j = 0; j < n; j++) if (e->evoterDom >> j & 1) p = put_log(p, &e->evoterLog[j]);
    }
    *p++ = (uint8_t)s->na;
    for (int k = 0; k < s->na; k++) p = put_log(p, &s->al[k]);
    for (int i = 0; i < n; i++) {
        *p++ = s->currentTerm[i]; *p++ = s->state[i]; *p++ = s->votedFor[i];
        p = put_log(p, &s->log[i]);
        *p++ = s->commitIndex[i]; *p++ = s->votesGranted[i]; *p++ = s->voterDom[i];
        for (int j = 0; j < n; j++) if (s->voterDom[i] >> j & 1) p = put_log(p, &s->voterLog[i][j]);
        for (int j = 0; j < n; j++) { *p++ = s->nextIndex[i][j]; *p++ = s->matchIndex[i][j]; }
    }
    *p++ = s->clientRequests;
    p = put_log(p, &s->committedLog);
    *p++ = s->committedLogDecrease;
    return (size_t)(p - out);
}
static void deser(const raft_ctx *c, const uint8_t *p, State *s) {
    int n = c->n;
    memset(s, 0, sizeof *s);
    s->nm = *p++;
    for (int k = 0; k < s->nm; k++) {
        Msg *m = &s->msg[k];
        m->mtype = *p++; m->mterm = *p++; m->msource = *p++; m->mdest = *p++; s->cnt[k] = *p++;
        switch (m->mtype) {
        case RVREQ: m->mlastLogTerm = *p++; m->mlastLogIndex = *p++; break;
        case RVRESP: m->mvoteGranted = *p++; p = get_log(p, &m->mlog); break;
        case AEREQ:
            m->mprevLogIndex = *p++; m->mprevLogTerm = *p++; m->nentries = *p++;
            m->mentry.term = *p++; m->mentry.value = *p++; m->mcommitIndex = *p++;
            p = get_log(p, &m->mlog);
            break;
        default: m->msuccess = *p++; m->mmatchIndex = *p++; break;
        }
    }
    s->ne = *p++;
    for (int k = 0; k < s->ne; k++) {
        Election *e = &s->el[k];
        e->eterm = *p++; e->eleader = *p++; e->evotes = *p++; e->evoterDom = *p++;
        p = get_log(p, &e->elog);
        for (int j = 0; j < n; j++) if (e->evoterDom >> j & 1) p = get_log(p, &e->evoterLog[j]);
    }
    s->na = *p++;
    for (int k = 0; k < s->na; k++) p = get_log(p, &s->al[k]);
    for (int i = 0; i < n; i++) {
        s->currentTerm[i] = *p++; s->state[i] = *p++; s->votedFor[i] = *p++;
        p = get_log(p, &s->log[i]);
        s->commitIndex[i] = *p++; s->votesGranted[i] = *p++; s->voterDom[i] = *p++;
        for (int j = 0; j < n; j++) if (s->voterDom[i] >> j & 1) p = get_log(p, &s->voterLog[i][j]);
        for (int j = 0; j < n; j++) { s->nextIndex[i][j] = *p++; s->matchIndex[i][j] = *p++; }
    }
    s->clientRequests = *p++;
    p = get_log(p, &s->committedLog);
    s->committedLogDecrease = *p++;
}

/* ---------------------------------------------------------------- helpers (raft.tla:106-151) */
static int popcount8(unsigned v) { int c = 0; while (v) { c += v & 1; v >>= 1; } return c; }
/* Quorum == {i \in SUBSET(Server) : Cardinality(i) * 2 > Cardinality(Server)}   raft.tla:110 */
static int in_quorum(const raft_ctx *c, unsigned set) { return popcount8(set) * 2 > c->n; }
/* LastTerm(xlog)   raft.tla:113 */
static int last_term(const Log *l) { return l->len == 0 ? 0 : l->e[l->len - 1].term; }

static void die(const char *what) {
    fprintf(stderr, "oracle/raft: capacity exceeded (%s) — raise the R* constants\n", what);
    abort();
}
/* position of m in the sorted key array, or -(insertion point)-1 */
static int msg_find(const State *s, const Msg *m) {
    int lo = 0, hi = s->nm;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        int c = memcmp(&s->msg[mid], m, sizeof *m);
        if (c == 0) return mid;
        if (c < 0) lo = mid + 1; else hi = mid;
    }
    return -lo - 1;
}
/* WithMessage(m, msgs)   raft.tla:117-121 */
static void with_message(State *s, const Msg *m) {
    int k = msg_find(s, m);
    if (k >= 0) { if (s->cnt[k] < 2) s->cnt[k]++; return; }
    k = -k - 1;
    if (s->nm >= RM) die("messages");
    memmove(&s->msg[k + 1], &s->msg[k], (size_t)(s->nm - k) * sizeof(Msg));
    memmove(&s->cnt[k + 1], &s->cnt[k], (size_t)(s->nm - k));
    s->msg[k] = *m;
    s->cnt[k] = 1;
    s->nm++;
}
/* WithoutMessage(m, msgs)   raft.tla:125-129 */
static void without_message(State *s, const Msg *m) {
    int k = msg_find(s, m);
    if (k >= 0 && s->cnt[k] > 0) s->cnt[k]--;
}
/* Reply(response, request)   raft.tla:145-146 */
static void reply(State *s, const Msg *resp, const Msg *req) {
    with_message(s, resp);
    without_message(s, req);
}
static int log_cmp(const Log *a, const Log *b) { return memcmp(a, b, sizeof *a); }
static void alllogs_add(State *s, const Log *l) {
    int k = 0;
    while (k < s->na && log_cmp(&s->al[k], l) < 0) k++;
    if (k < s->na && log_cmp(&s->al[k], l) == 0) return;
    if (s->na >= RA) die("allLogs");
    memmove(&s->al[k + 1], &s->al[k], (size_t)(s->na - k) * sizeof(Log));
    s->al[k] = *l;
    s->na++;
}
static void elections_add(State *s, const Election *e) {
    int k = 0;
    while (k < s->ne && memcmp(&s->el[k], e, sizeof *e) < 0) k++;
    if (k < s->ne && memcmp(&s->el[k], e, sizeof *e) == 0) return;
    if (s->ne >= RE) die("elections");
    memmove(&s->el[k + 1], &s->el[k], (size_t)(s->ne - k) * sizeof(Election));
    s->el[k] = *e;
    s->ne++;
}

typedef struct {
    const raft_ctx *c;
    or_emit *em;
    const State *parent;
    uint8_t buf[16384];
} gen_t;

/* Every disjunct of Next is conjoined with allLogs' = allLogs \cup {log[i] : i \in Server}
 * over the UNPRIMED logs (raft.tla:493). */
static void emit(gen_t *g, State *t, int action, unsigned flags) {
    for (int i = 0; i < g->c->n; i++) alllogs_add(t, &g->parent->log[i]);
    size_t len = ser(g->c, t, g->buf);
    g->em->emit(g->em, g->buf, len, action, flags);
}

/* ---------------------------------------------------------------- actions */
/* Restart(i)   raft.tla:186-194 */
static void restart(gen_t *g, int i) {
    State t = *g->parent;
    int n = g->c->n;
    t.state[i] = FOLLOWER;
    t.votesGranted[i] = 0;
    t.voterDom[i] = 0;
    memset(t.voterLog[i], 0, sizeof t.voterLog[i]);
    for (int j = 0; j < n; j++) { t.nextIndex[i][j] = 1; t.matchIndex[i][j] = 0; }
    t.commitIndex[i] = 0;
    emit(g, &t, A_RESTART, 0);
}
/* Timeout(i)   raft.tla:197-206 */
static void timeout_(gen_t *g, int i) {
    const State *s = g->parent;
    if (!(s->state[i] == FOLLOWER || s->state[i] == CANDIDATE)) return;
    State t = *s;
    t.state[i] = CANDIDATE;
    t.currentTerm[i] = s->currentTerm[i] + 1;
    t.votedFor[i] = 0;
    t.votesGranted[i] = 0;
    t.voterDom[i] = 0;
    memset(t.voterLog[i], 0, sizeof t.voterLog[i]);
    emit(g, &t, A_TIMEOUT, 0);
}
/* RequestVote(i, j)   raft.tla:209-217 (no i /= j guard) */
static void request_vote(gen_t *g, int i, int j) {
    const State *s = g->parent;
    if (s->state[i] != CANDIDATE) return;
    State t = *s;
    Msg m;
    memset(&m, 0, sizeof m);
    m.mtype = RVREQ; m.mterm = s->currentTerm[i];
    m.mlastLogTerm = (uint8_t)last_term(&s->log[i]);
    m.mlastLogIndex = s->log[i].len;
    m.msource = (uint8_t)i; m.mdest = (uint8_t)j;
    with_message(&t, &m);
    emit(g, &t, A_REQUESTVOTE, 0);
}
/* AppendEntries(i, j)   raft.tla:222-244 */
static void append_entries(gen_t *g, int i, int j) {
    const State *s = g->parent;
    if (i == j || s->state[i] != LEADER) return;
    State t = *s;
    unsigned flags = 0;
    int prevLogIndex = s->nextIndex[i][j] - 1;
    int prevLogTerm = 0;
    if (prevLogIndex > 0) {
        if (prevLogIndex > s->log[i].len) flags |= OR_FLAG_SPECERR; /* log[i][prevLogIndex] undefined */
        else prevLogTerm = s->log[i].e[prevLogIndex - 1].term;
    }
    /* lastEntry == Min({Len(log[i]), nextIndex[i][j]}) */
    int lastEntry = s->log[i].len < s->nextIndex[i][j] ? s->log[i].len : s->nextIndex[i][j];
    Msg m;
    memset(&m, 0, sizeof m);
    m.mtype = AEREQ; m.mterm = s->currentTerm[i];
    m.mprevLogIndex = (uint8_t)prevLogIndex; m.mprevLogTerm = (uint8_t)prevLogTerm;
    /* entries == SubSeq(log[i], nextIndex[i][j], lastEntry): at most one entry */
    if (s->nextIndex[i][j] <= lastEntry) { m.nentries = 1; m.mentry = s->log[i].e[s->nextIndex[i][j] - 1]; }
    m.mlog = s->log[i];
    m.mcommitIndex = (uint8_t)(s->commitIndex[i] < lastEntry ? s->commitIndex[i] : lastEntry);
    m.msource = (uint8_t)i; m.mdest = (uint8_t)j;
    with_message(&t, &m);
    emit(g, &t, A_APPENDENTRIES, flags);
}
/* BecomeLeader(i)   raft.tla:247-261 */
static void become_leader(gen_t *g, int i) {
    const State *s = g->parent;
    int n = g->c->n;
    if (s->state[i] != CANDIDATE || !in_quorum(g->c, s->votesGranted[i])) return;
    State t = *s;
    t.state[i] = LEADER;
    for (int j = 0; j < n; j++) { t.nextIndex[i][j] = (uint8_t)(s->log[i].len + 1); t.matchIndex[i][j] = 0; }
    Election e;
    memset(&e, 0, sizeof e);
    e.eterm = s->currentTerm[i]; e.eleader = (uint8_t)i; e.elog = s->log[i];
    e.evotes = s->votesGranted[i]; e.evoterDom = s->voterDom[i];
    for (int j = 0; j < n; j++) e.evoterLog[j] = s->voterLog[i][j];
    elections_add(&t, &e);
    emit(g, &t, A_BECOMELEADER, 0);
}
/* ClientRequest(i)   raft.tla:264-274 */
static void client_request(gen_t *g, int i) {
    const State *s = g->parent;
    if (s->state[i] != LEADER || !(s->clientRequests < g->c->max_client_requests)) return;
    State t = *s;
    if (t.log[i].len >= RL) die("log length");
    t.log[i].e[t.log[i].len].term = s->currentTerm[i];
    t.log[i].e[t.log[i].len].value = s->clientRequests;
    t.log[i].len++;
    t.clientRequests = s->clientRequests + 1;
    emit(g, &t, A_CLIENTREQUEST, 0);
}
/* AdvanceCommitIndex(i)   raft.tla:280-305 */
static void advance_commit_index(gen_t *g, int i) {
    const State *s = g->parent;
    int n = g->c->n;
    if (s->state[i] != LEADER) return;
    State t = *s;
    unsigned flags = 0;
    int maxAgree = 0; /* Max(agreeIndexes), 0 if the set is empty */
    for (int index = 1; index <= s->log[i].len; index++) {
        unsigned agree = 1u << i;
        for (int k = 0; k < n; k++) if (s->matchIndex[i][k] >= index) agree |= 1u << k;
        if (in_quorum(g->c, agree)) maxAgree = index;
    }
    int newCommitIndex = (maxAgree > 0 && s->log[i].e[maxAgree - 1].term == s->currentTerm[i])
                             ? maxAgree : s->commitIndex[i];
    Log newCommittedLog;
    memset(&newCommittedLog, 0, sizeof newCommittedLog);
    if (newCommitIndex > 1) {
        if (newCommitIndex > s->log[i].len) flags |= OR_FLAG_SPECERR; /* log[i][j] undefined */
        else { newCommittedLog.len = (uint8_t)newCommitIndex;
               for (int j = 0; j < newCommitIndex; j++) newCommittedLog.e[j] = s->log[i].e[j]; }
    }
    int decrease = newCommitIndex < s->committedLog.len;
    if (!decrease && !(flags & OR_FLAG_SPECERR))
        for (int j = 0; j < s->committedLog.len; j++)
            if (memcmp(&s->committedLog.e[j], &newCommittedLog.e[j], sizeof(Entry)) != 0) decrease = 1;
    t.commitIndex[i] = (uint8_t)newCommitIndex;
    t.committedLogDecrease = (uint8_t)decrease;
    t.committedLog = newCommittedLog;
    emit(g, &t, A_ADVANCECOMMIT, flags);
}

/* Receive(m)   raft.tla:449-464; the disjuncts are mutually exclusive for a given m */
static void receive(gen_t *g, int k) {
    const State *s = g->parent;
    const raft_ctx *c = g->c;
    const Msg *m = &s->msg[k];
    int i = m->mdest, j = m->msource;
    State t = *s;
    /* UpdateTerm(i, j, m)   raft.tla:434-440 — does not consume m */
    if (m->mterm > s->currentTerm[i]) {
        t.currentTerm[i] = m->mterm; t.state[i] = FOLLOWER; t.votedFor[i] = 0;
        emit(g, &t, A_RECEIVE, 0);
        return;
    }
    Msg r;
    memset(&r, 0, sizeof r);
    switch (m->mtype) {
    case RVREQ: { /* HandleRequestVoteRequest   raft.tla:313-332 (m.mterm <= currentTerm[i] holds here) */
        int lt = last_term(&s->log[i]);
        int logOk = m->mlastLogTerm > lt || (m->mlastLogTerm == lt && m->mlastLogIndex >= s->log[i].len);
        int grant = m->mterm == s->currentTerm[i] && logOk && (s->votedFor[i] == 0 || s->votedFor[i] == j + 1);
        if (grant) t.votedFor[i] = (uint8_t)(j + 1);
        r.mtype = RVRESP; r.mterm = s->currentTerm[i]; r.mvoteGranted = (uint8_t)grant;
        r.mlog = s->log[i]; r.msource = (uint8_t)i; r.mdest = (uint8_t)j;
        reply(&t, &r, m);
        emit(g, &t, A_RECEIVE, 0);
        return;
    }
    case RVRESP:
        if (m->mterm < s->currentTerm[i]) { /* DropStaleResponse   raft.tla:443-446 */
            without_message(&t, m);
            emit(g, &t, A_RECEIVE, 0);
            return;
        }
        /* HandleRequestVoteResponse   raft.tla:336-349 (m.mterm = currentTerm[i]) */
        if (m->mvoteGranted) {
            t.votesGranted[i] |= (uint8_t)(1u << j);
            if (!(s->voterDom[i] >> j & 1)) { /* f @@ g keeps f's value (TLC.tla:11-12) */
                t.voterDom[i] |= (uint8_t)(1u << j);
                t.voterLog[i][j] = m->mlog;
            }
        }
        without_message(&t, m);
        emit(g, &t, A_RECEIVE, 0);
        return;
    case AEREQ: { /* HandleAppendEntriesRequest   raft.tla:355-417 */
        int logOk = m->mprevLogIndex == 0 ||
                    (m->mprevLogIndex > 0 && m->mprevLogIndex <= s->log[i].len &&
                     m->mprevLogTerm == s->log[i].e[m->mprevLogIndex - 1].term);
        if (m->mterm < s->currentTerm[i] ||
            (m->mterm == s->currentTerm[i] && s->state[i] == FOLLOWER && !logOk)) { /* reject   :361-373 */
            r.mtype = AERESP; r.mterm = s->currentTerm[i]; r.msuccess = 0; r.mmatchIndex = 0;
            r.msource = (uint8_t)i; r.mdest = (uint8_t)j;
            reply(&t, &r, m);
            emit(g, &t, A_RECEIVE, 0);
            return;
        }
        if (m->mterm == s->currentTerm[i] && s->state[i] == CANDIDATE) { /* return to follower   :374-378 */
            t.state[i] = FOLLOWER;
            emit(g, &t, A_RECEIVE, 0);
            return;
        }
        if (m->mterm == s->currentTerm[i] && s->state[i] == FOLLOWER && logOk) { /* accept   :379-416 */
            int index = m->mprevLogIndex + 1;
            if (m->nentries == 0 ||
                (s->log[i].len >= index && s->log[i].e[index - 1].term == m->mentry.term)) {
                /* already done with request   :384-402.  commitIndex' = [.. ![i] = m.mcommitIndex]
                 * followed by UNCHANGED logVars: enabled only if the value does not change. */
                if (!c->naive_commit && m->mcommitIndex != s->commitIndex[i]) return;
                t.commitIndex[i] = m->mcommitIndex;
                r.mtype = AERESP; r.mterm = s->currentTerm[i]; r.msuccess = 1;
                r.mmatchIndex = (uint8_t)(m->mprevLogIndex + m->nentries);
                r.msource = (uint8_t)i; r.mdest = (uint8_t)j;
                reply(&t, &r, m);
                emit(g, &t, A_RECEIVE, 0);
                return;
            }
            if (s->log[i].len >= index && s->log[i].e[index - 1].term != m->mentry.term) {
                /* conflict: remove 1 entry   :403-410 (message not consumed) */
                t.log[i].len--;
                memset(&t.log[i].e[t.log[i].len], 0, sizeof(Entry));
                emit(g, &t, A_RECEIVE, 0);
                return;
            }
            if (s->log[i].len == m->mprevLogIndex) { /* no conflict: append entry   :411-416 */
                if (t.log[i].len >= RL) die("log length");
                t.log[i].e[t.log[i].len++] = m->mentry;
                emit(g, &t, A_RECEIVE, 0);
                return;
            }
        }
        return; /* e.g. a Leader receiving an AppendEntries of its own term: no disjunct enabled */
    }
    default: /* AERESP */
        if (m->mterm < s->currentTerm[i]) { /* DropStaleResponse */
            without_message(&t, m);
            emit(g, &t, A_RECEIVE, 0);
            return;
        }
        /* HandleAppendEntriesResponse   raft.tla:421-431 (m.mterm = currentTerm[i]) */
        if (m->msuccess) {
            t.nextIndex[i][j] = (uint8_t)(m->mmatchIndex + 1);
            t.matchIndex[i][j] = m->mmatchIndex;
        } else { /* Max({nextIndex[i][j] - 1, 1}) */
            t.nextIndex[i][j] = (uint8_t)(s->nextIndex[i][j] - 1 > 1 ? s->nextIndex[i][j] - 1 : 1);
        }
        without_message(&t, m);
        emit(g, &t, A_RECEIVE, 0);
        return;
    }
}

/* Next   raft.tla:482-493 */
static void raft_succ(void *ctx, const uint8_t *sb, size_t len, or_emit *em) {
    const raft_ctx *c = ctx;
    (void)len;
    static _Thread_local State s; /* single-threaded oracle */
    deser(c, sb, &s);
    gen_t g;
    g.c = c; g.em = em; g.parent = &s;
    int n = c->n;
    for (int i = 0; i < n; i++) restart(&g, i);
    for (int i = 0; i < n; i++) timeout_(&g, i);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) request_vote(&g, i, j);
    for (int i = 0; i < n; i++) become_leader(&g, i);
    for (int i = 0; i < n; i++) client_request(&g, i);
    for (int i = 0; i < n; i++) advance_commit_index(&g, i);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) append_entries(&g, i, j);
    /* \E m \in ValidMessage(messages) : Receive(m)   raft.tla:131-132,489 */
    for (int k = 0; k < s.nm; k++) if (s.cnt[k] > 0) receive(&g, k);
    /* \E m \in SingleMessage(messages) : DuplicateMessage(m)   raft.tla:134-135,471-473,490 */
    for (int k = 0; k < s.nm; k++) if (s.cnt[k] == 1) {
        State t = s; with_message(&t, &s.msg[k]); emit(&g, &t, A_DUPLICATE, 0);
    }
    /* \E m \in ValidMessage(messages) : DropMessage(m)   raft.tla:476-478,491 */
    for (int k = 0; k < s.nm; k++) if (s.cnt[k] > 0) {
        State t = s; without_message(&t, &s.msg[k]); emit(&g, &t, A_DROP, 0);
    }
}

/* Init   raft.tla:156-179 */
static int raft_n_init(void *ctx) { (void)ctx; return 1; }
static size_t raft_init(void *ctx, int k, uint8_t *out) {
    const raft_ctx *c = ctx; (void)k;
    static _Thread_local State s;
    memset(&s, 0, sizeof s);
    for (int i = 0; i < c->n; i++) {
        s.currentTerm[i] = 1; s.state[i] = FOLLOWER; s.votedFor[i] = 0;
        for (int j = 0; j < c->n; j++) { s.nextIndex[i][j] = 1; s.matchIndex[i][j] = 0; }
    }
    s.clientRequests = 1;
    return ser(c, &s, out);
}
/* StateConstraint of specs/MCraft.tla */
static int raft_constraint(void *ctx, const uint8_t *sb, size_t len) {
    const raft_ctx *c = ctx; (void)len;
    static _Thread_local State s;
    deser(c, sb, &s);
    for (int i = 0; i < c->n; i++) {
        if (s.currentTerm[i] > c->max_term) return 0;
        if (s.log[i].len > c->max_log_len) return 0;
    }
    int inflight = 0;
    for (int k = 0; k < s.nm; k++) inflight += s.cnt[k];
    if (c->max_keys && s.nm > c->max_keys) return 0;  /* Cardinality(DOMAIN messages) <= MaxMsgKeys */
    return inflight <= c->max_msgs;
}
/* NoTwoLeaders == ~MoreThanOneLeader (raft.tla:500-507); CommittedLogStable == ~committedLogDecrease */
static int raft_invariant(void *ctx, const uint8_t *sb, size_t len) {
    const raft_ctx *c = ctx; (void)len;
    static _Thread_local State s;
    deser(c, sb, &s);
    if (c->inv_mask & 1)
        for (int i = 0; i < c->n; i++) for (int j = 0; j < c->n; j++)
            if (i != j && s.currentTerm[i] == s.currentTerm[j] && s.state[i] == LEADER && s.state[j] == LEADER) return 0;
    if ((c->inv_mask & 2) && s.committedLogDecrease) return 1;
    return -1;
}
static void raft_stats(void *ctx, const uint8_t *sb, size_t len, uint64_t *mx) {
    const raft_ctx *c = ctx; (void)len;
    static _Thread_local State s;
    deser(c, sb, &s);
    uint64_t inflight = 0;
    for (int k = 0; k < s.nm; k++) inflight += s.cnt[k];
    if ((uint64_t)s.nm > mx[0]) mx[0] = (uint64_t)s.nm;
    if ((uint64_t)s.ne > mx[1]) mx[1] = (uint64_t)s.ne;
    if ((uint64_t)s.na > mx[2]) mx[2] = (uint64_t)s.na;
    if (inflight > mx[3]) mx[3] = inflight;
    if (len > mx[4]) mx[4] = len;
}

/* ---------------------------------------------------------------- printing (canonical TLA+ text)
 * Format shared with the engine's mc_state_format (documented in DESIGN.md §"state text"):
 * records with fields in alphabetical order, functions as (k :> v @@ ...), sets as {..} with
 * the elements sorted by strcmp of their printed text, empty function/sequence as <<>>. */
typedef struct { char *b; size_t cap, k; } sb_t;
static void sb_put(sb_t *o, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void sb_put(sb_t *o, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    if (o->k < o->cap) {
        int w = vsnprintf(o->b + o->k, o->cap - o->k, fmt, ap);
        if (w > 0) o->k += (size_t)w;
        if (o->k > o->cap) o->k = o->cap;
    }
    va_end(ap);
}
static void p_log(sb_t *o, const Log *l) {
    if (l->len == 0) { sb_put(o, "<<>>"); return; }
    sb_put(o, "<<");
    for (int i = 0; i < l->len; i++) sb_put(o, "%s[term |-> %d, value |-> %d]", i ? ", " : "", l->e[i].term, l->e[i].value);
    sb_put(o, ">>");
}
static void p_set_of_servers(sb_t *o, unsigned mask, int n) {
    sb_put(o, "{");
    int first = 1;
    for (int j = 0; j < n; j++) if (mask >> j & 1) { sb_put(o, "%ss%d", first ? "" : ", ", j + 1); first = 0; }
    sb_put(o, "}");
}
static void p_voterlog(sb_t *o, unsigned dom, const Log *vl, int n) {
    if (!dom) { sb_put(o, "<<>>"); return; }
    sb_put(o, "(");
    int first = 1;
    for (int j = 0; j < n; j++) if (dom >> j & 1) {
        sb_put(o, "%ss%d :> ", first ? "" : " @@ ", j + 1); p_log(o, &vl[j]); first = 0;
    }
    sb_put(o, ")");
}
static void p_msg(sb_t *o, const Msg *m) {
    switch (m->mtype) {
    case RVREQ:
        sb_put(o, "[mdest |-> s%d, mlastLogIndex |-> %d, mlastLogTerm |-> %d, msource |-> s%d, mterm |-> %d, mtype |-> RequestVoteRequest]",
               m->mdest + 1, m->mlastLogIndex, m->mlastLogTerm, m->msource + 1, m->mterm);
        break;
    case RVRESP:
        sb_put(o, "[mdest |-> s%d, mlog |-> ", m->mdest + 1); p_log(o, &m->mlog);
        sb_put(o, ", msource |-> s%d, mterm |-> %d, mtype |-> RequestVoteResponse, mvoteGranted |-> %s]",
               m->msource + 1, m->mterm, m->mvoteGranted ? "TRUE" : "FALSE");
        break;
    case AEREQ:
        sb_put(o, "[mcommitIndex |-> %d, mdest |-> s%d, mentries |-> ", m->mcommitIndex, m->mdest + 1);
        if (m->nentries) sb_put(o, "<<[term |-> %d, value |-> %d]>>", m->mentry.term, m->mentry.value);
        else sb_put(o, "<<>>");
        sb_put(o, ", mlog |-> "); p_log(o, &m->mlog);
        sb_put(o, ", mprevLogIndex |-> %d, mprevLogTerm |-> %d, msource |-> s%d, mterm |-> %d, mtype |-> AppendEntriesRequest]",
               m->mprevLogIndex, m->mprevLogTerm, m->msource + 1, m->mterm);
        break;
    default:
        sb_put(o, "[mdest |-> s%d, mmatchIndex |-> %d, msource |-> s%d, msuccess |-> %s, mterm |-> %d, mtype |-> AppendEntriesResponse]",
               m->mdest + 1, m->mmatchIndex, m->msource + 1, m->msuccess ? "TRUE" : "FALSE", m->mterm);
        break;
    }
}
static int cmp_str(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }
static void p_sorted(sb_t *o, char **items, int n, const char *open, const char *sep, const char *close, const char *empty) {
    if (n == 0) { sb_put(o, "%s", empty); return; }
    qsort(items, (size_t)n, sizeof *items, cmp_str);
    sb_put(o, "%s", open);
    for (int i = 0; i < n; i++) { sb_put(o, "%s%s", i ? sep : "", items[i]); free(items[i]); }
    sb_put(o, "%s", close);
}
static const char *st_name[] = {"Follower", "Candidate", "Leader"};
static size_t raft_print(void *ctx, const uint8_t *sbytes, size_t len, char *buf, size_t cap) {
    const raft_ctx *c = ctx; (void)len;
    static _Thread_local State s;
    deser(c, sbytes, &s);
    int n = c->n;
    sb_t o = {buf, cap, 0};
    char tmp[2048];
    char *items[RM > RA ? RM : RA];
    /* messages */
    sb_put(&o, "/\\ messages = ");
    for (int k = 0; k < s.nm; k++) {
        sb_t e = {tmp, sizeof tmp, 0};
        p_msg(&e, &s.msg[k]); sb_put(&e, " :> %d", s.cnt[k]);
        items[k] = strndup(tmp, e.k);
    }
    p_sorted(&o, items, s.nm, "(", " @@ ", ")", "<<>>");
    sb_put(&o, "\n/\\ elections = ");
    for (int k = 0; k < s.ne; k++) {
        sb_t e = {tmp, sizeof tmp, 0};
        const Election *el = &s.el[k];
        sb_put(&e, "[eleader |-> s%d, elog |-> ", el->eleader + 1); p_log(&e, &el->elog);
        sb_put(&e, ", eterm |-> %d, evoterLog |-> ", el->eterm); p_voterlog(&e, el->evoterDom, el->evoterLog, n);
        sb_put(&e, ", evotes |-> "); p_set_of_servers(&e, el->evotes, n); sb_put(&e, "]");
        items[k] = strndup(tmp, e.k);
    }
    p_sorted(&o, items, s.ne, "{", ", ", "}", "{}");
    sb_put(&o, "\n/\\ allLogs = ");
    for (int k = 0; k < s.na; k++) {
        sb_t e = {tmp, sizeof tmp, 0};
        p_log(&e, &s.al[k]);
        items[k] = strndup(tmp, e.k);
    }
    p_sorted(&o, items, s.na, "{", ", ", "}", "{}");
#define PER_SERVER(title, expr)                                                   \
    sb_put(&o, "\n/\\ " title " = (");                                            \
    for (int i = 0; i < n; i++) { sb_put(&o, "%ss%d :> ", i ? " @@ " : "", i + 1); expr; } \
    sb_put(&o, ")");
    PER_SERVER("currentTerm", sb_put(&o, "%d", s.currentTerm[i]));
    PER_SERVER("state", sb_put(&o, "%s", st_name[s.state[i]]));
    PER_SERVER("votedFor", if (s.votedFor[i]) sb_put(&o, "s%d", s.votedFor[i]); else sb_put(&o, "Nil"));
    sb_put(&o, "\n/\\ clientRequests = %d", s.clientRequests);
    PER_SERVER("log", p_log(&o, &s.log[i]));
    PER_SERVER("commitIndex", sb_put(&o, "%d", s.commitIndex[i]));
    sb_put(&o, "\n/\\ committedLog = "); p_log(&o, &s.committedLog);
    sb_put(&o, "\n/\\ committedLogDecrease = %s", s.committedLogDecrease ? "TRUE" : "FALSE");
    PER_SERVER("votesSent", sb_put(&o, "FALSE"));
    PER_SERVER("votesGranted", p_set_of_servers(&o, s.votesGranted[i], n));
    PER_SERVER("voterLog", p_voterlog(&o, s.voterDom[i], s.voterLog[i], n));
    PER_SERVER("nextIndex", { sb_put(&o, "("); for (int j = 0; j < n; j++) sb_put(&o, "%ss%d :> %d", j ? " @@ " : "", j + 1, s.nextIndex[i][j]); sb_put(&o, ")"); });
    PER_SERVER("matchIndex", { sb_put(&o, "("); for (int j = 0; j < n; j++) sb_put(&o, "%ss%d :> %d", j ? " @@ " : "", j + 1, s.matchIndex[i][j]); sb_put(&o, ")"); });
#undef PER_SERVER
    return o.k;
}

const char *or_raft_action(int a) {
    static const char *nm[] = {"Restart", "Timeout", "RequestVote", "BecomeLeader", "ClientRequest",
                               "AdvanceCommitIndex", "AppendEntries", "Receive", "DuplicateMessage", "DropMessage"};
    return a >= 0 && a < A_NACT ? nm[a] : "?";
}
int or_spec_raft(const int64_t *p, int np, or_spec *o) {
    if (np < 5) { or_set_error("raft: need {nServer, MaxClientRequests, MaxTerm, MaxLogLen, MaxMsgs[, invmask, naive]}"); return -1; }
    raft_ctx *c = calloc(1, sizeof *c);
    c->n = (int)p[0]; c->max_client_requests = (int)p[1]; c->max_term = (int)p[2];
    c->max_log_len = (int)p[3]; c->max_msgs = (int)p[4];
    c->inv_mask = np > 5 ? (int)p[5] : 1;
    c->naive_commit = np > 6 ? (int)p[6] : 0;
    c->max_keys = np > 7 ? (int)p[7] : 0;  /* MaxMsgKeys of specs/MCraft.tla, 0 = unbounded */
    if (c->n < 1 || c->n > RN || c->max_client_requests < 1 || c->max_client_requests - 1 > RL || c->max_term < 1 || c->max_term > 250) {
        or_set_error("raft: parameters out of range"); free(c); return -1;
    }
    o->name = "raft"; o->ctx = c; o->max_state_bytes = 16384;
    o->n_init = raft_n_init; o->init = raft_init; o->succ = raft_succ;
    o->constraint = raft_constraint; o->invariant = raft_invariant; o->print = raft_print;
    o->action_name = or_raft_action; o->stats = raft_stats;
    return 0;
}
