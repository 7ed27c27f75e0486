/*
 * oracle/bfs.c — exact breadth-first search core of the CPU oracle (TEST INFRASTRUCTURE).
 *
 * Restates the observable contract of TLC's BFS (it is an external Java tool; the
 * reference only shows its behaviour: README.md:267-321, testout2:1-266, p-manual §4):
 * FIFO frontier, "seen" set, CONSTRAINT filter, invariant / Assert / deadlock checks,
 * the three counters and the depth.  Dedup is EXACT: the seen-set stores whole canonical
 * state byte strings, never fingerprints, so a count produced here cannot be off by a
 * hash collision.
 */
#include "oracle_int.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static char g_err[512];
void or_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *oracle_last_error(void) { return g_err; }

typedef struct {
    const or_spec *spec;
    const or_options *opt;
    or_result *res;
    /* state store */
    uint8_t *arena;
    uint64_t arena_len, arena_cap;
    uint64_t *off;     /* n+1 offsets */
    uint32_t *parent;  /* parent index, UINT32_MAX for initial states */
    int16_t *act;
    uint64_t n, ncap;
    /* optional key arena (spec->canon): what the seen-set compares; entry i belongs to state i */
    uint8_t *karena, *kbuf;
    uint64_t karena_len, karena_cap;
    uint64_t *koff;
    /* exact seen-set: open addressing over state indices */
    uint32_t *tab;
    uint64_t tab_cap; /* power of two */
    /* expansion context */
    uint64_t cur;        /* index of the state being expanded (UINT64_MAX during init) */
    uint32_t cur_level;  /* level (1-based) of the states being created                */
    uint64_t nsucc;
    FILE *dump;
    char *pbuf;
    size_t pcap;
    /* first violation */
    int have_violation;
    uint64_t viol_parent;
    uint8_t *viol_state;
    size_t viol_len;
    int viol_action;
    int viol_in_arena; /* the violating state is arena[viol_parent] itself (deadlock / new state) */
} bfs_t;

static uint64_t hash_bytes(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9e3779b97f4a7c15ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 32;
    return h;
}

static void tab_grow(bfs_t *b) {
    uint64_t ncap = b->tab_cap ? b->tab_cap * 2 : (1u << 16);
    uint32_t *nt = calloc(ncap, sizeof *nt);
    if (!nt) { fprintf(stderr, "oracle: out of memory (table)\n"); abort(); }
    for (uint64_t i = 0; i < b->n; i++) {
        uint64_t h = (b->koff ? hash_bytes(b->karena + b->koff[i], b->koff[i + 1] - b->koff[i])
                              : hash_bytes(b->arena + b->off[i], b->off[i + 1] - b->off[i])) & (ncap - 1);
        while (nt[h]) h = (h + 1) & (ncap - 1);
        nt[h] = (uint32_t)(i + 1);
    }
    free(b->tab);
    b->tab = nt;
    b->tab_cap = ncap;
}

/* returns 1 if the state was new (and appends it), 0 if already present */
static int store_insert(bfs_t *b, const uint8_t *s, size_t len, uint32_t parent, int action) {
    if ((b->n + 1) * 2 > b->tab_cap) tab_grow(b);
    uint64_t mask = b->tab_cap - 1;
    const uint8_t *key = s;
    size_t klen = len;
    if (b->spec->canon) {  /* dedup on the orbit's canonical form, store the state as it was generated */
        if (!b->kbuf) b->kbuf = malloc(b->spec->max_state_bytes);
        klen = b->spec->canon(b->spec->ctx, s, len, b->kbuf);
        key = b->kbuf;
    }
    uint64_t h = hash_bytes(key, klen) & mask;
    while (b->tab[h]) {
        uint64_t i = b->tab[h] - 1;
        if (b->spec->canon) {
            if (b->koff[i + 1] - b->koff[i] == klen && memcmp(b->karena + b->koff[i], key, klen) == 0) return 0;
        } else if (b->off[i + 1] - b->off[i] == len && memcmp(b->arena + b->off[i], s, len) == 0) return 0;
        h = (h + 1) & mask;
    }
    if (b->n + 2 > b->ncap) {
        b->ncap = b->ncap ? b->ncap * 2 : 1024;
        b->off = realloc(b->off, (b->ncap + 1) * sizeof *b->off);
        if (b->spec->canon) {
            b->koff = realloc(b->koff, (b->ncap + 1) * sizeof *b->koff);
            if (!b->koff) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
            if (b->n == 0) b->koff[0] = 0;
        }
        b->parent = realloc(b->parent, b->ncap * sizeof *b->parent);
        b->act = realloc(b->act, b->ncap * sizeof *b->act);
        if (!b->off || !b->parent || !b->act) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
        if (b->n == 0) b->off[0] = 0;
    }
    if (b->arena_len + len > b->arena_cap) {
        while (b->arena_len + len > b->arena_cap) b->arena_cap = b->arena_cap ? b->arena_cap * 2 : (1u << 20);
        b->arena = realloc(b->arena, b->arena_cap);
        if (!b->arena) { fprintf(stderr, "oracle: out of memory (arena)\n"); abort(); }
    }
    memcpy(b->arena + b->arena_len, s, len);
    b->arena_len += len;
    b->off[b->n + 1] = b->arena_len;
    if (b->spec->canon) {
        if (b->karena_len + klen > b->karena_cap) {
            while (b->karena_len + klen > b->karena_cap) b->karena_cap = b->karena_cap ? b->karena_cap * 2 : (1u << 20);
            b->karena = realloc(b->karena, b->karena_cap);
            if (!b->karena) { fprintf(stderr, "oracle: out of memory (key arena)\n"); abort(); }
        }
        memcpy(b->karena + b->karena_len, key, klen);
        b->karena_len += klen;
        b->koff[b->n + 1] = b->karena_len;
    }
    b->parent[b->n] = parent;
    b->act[b->n] = (int16_t)action;
    b->tab[h] = (uint32_t)(b->n + 1);
    b->n++;
    if (b->n >= 0xfffffff0ull) { fprintf(stderr, "oracle: state index overflow\n"); abort(); }
    return 1;
}

static void note_violation(bfs_t *b, int verdict, int inv, uint64_t parent, const uint8_t *s, size_t len,
                           int action, int in_arena) {
    if (b->have_violation) return;
    b->have_violation = 1;
    b->res->verdict = verdict;
    b->res->violated_invariant = inv;
    b->viol_parent = parent;
    b->viol_action = action;
    b->viol_in_arena = in_arena;
    if (!in_arena) {
        b->viol_state = malloc(len ? len : 1);
        memcpy(b->viol_state, s, len);
        b->viol_len = len;
    }
}

static void on_emit(or_emit *em, const uint8_t *s, size_t len, int action, unsigned flags) {
    bfs_t *b = em->bfs;
    const or_spec *sp = b->spec;
    or_result *r = b->res;
    if (b->opt->stop_on_violation == 2) {
        /* TLC-like: evaluation stops at the first error; the failing successor is not counted */
        if (b->have_violation) return;
        if (flags & OR_FLAG_ASSERT) { b->nsucc++; note_violation(b, OR_ASSERT, -1, b->cur, NULL, 0, action, 1); return; }
    }
    b->nsucc++;
    r->generated++;
    if (b->cur_level - 1 < OR_MAX_LEVELS && b->cur != UINT64_MAX) r->level_generated[b->cur_level - 2]++;
    if (flags & OR_FLAG_SPECERR) {
        note_violation(b, OR_SPEC_ERROR, -1, b->cur, s, len, action, 0);
        return;
    }
    if (flags & OR_FLAG_ASSERT) {
        /* TLC reports the Assert failure while evaluating the action: the trace ends at the
         * state being expanded (README.md:268-311 prints 6 states, the last one being the
         * state in which `C` was evaluated). */
        note_violation(b, OR_ASSERT, -1, b->cur, NULL, 0, action, 1);
        return;
    }
    if (flags & OR_FLAG_PROPERTY) note_violation(b, OR_INVARIANT, (int)(flags >> 8), b->cur, s, len, action, 0);
    int inmodel = sp->constraint ? sp->constraint(sp->ctx, s, len) : 1;
    int is_new = 0;
    if (inmodel) {
        uint32_t par = b->cur == UINT64_MAX ? UINT32_MAX : (uint32_t)b->cur;
        is_new = store_insert(b, s, len, par, action);
        if (is_new) {
            r->distinct++;
            if (b->cur_level - 1 < OR_MAX_LEVELS) r->level_distinct[b->cur_level - 1]++;
            if (sp->stats) sp->stats(sp->ctx, s, len, r->max_stat);
            if (b->dump) {
                size_t k = sp->print(sp->ctx, s, len, b->pbuf, b->pcap);
                for (size_t i = 0; i < k; i++)
                    if (b->pbuf[i] == '\n') b->pbuf[i] = ' ';
                fprintf(b->dump, "L%u %.*s\n", b->cur_level, (int)k, b->pbuf);
            }
        }
    }
    if ((is_new || !inmodel) && sp->invariant) {
        int inv = sp->invariant(sp->ctx, s, len);
        if (inv >= 0) {
            if (is_new) note_violation(b, OR_INVARIANT, inv, b->n - 1, NULL, 0, action, 1);
            else note_violation(b, OR_INVARIANT, inv, b->cur, s, len, action, 0);
        }
    }
}

static char **g_trace;
static uint32_t g_trace_n;
static void trace_clear(void) {
    for (uint32_t i = 0; i < g_trace_n; i++) free(g_trace[i]);
    free(g_trace);
    g_trace = NULL;
    g_trace_n = 0;
}
const char *oracle_trace_state(uint32_t k) { return k < g_trace_n ? g_trace[k] : NULL; }

static void build_trace(bfs_t *b) {
    or_result *r = b->res;
    uint64_t chain[OR_MAX_TRACE];
    uint32_t n = 0;
    uint64_t i = b->viol_parent;
    while (i != UINT32_MAX && n < OR_MAX_TRACE - 1) {
        chain[n++] = i;
        i = b->parent[i];
    }
    uint32_t total = n + (b->viol_in_arena ? 0 : 1);
    g_trace = calloc(total, sizeof *g_trace);
    g_trace_n = total;
    for (uint32_t k = 0; k < n; k++) {
        uint64_t idx = chain[n - 1 - k];
        size_t len = b->off[idx + 1] - b->off[idx];
        size_t m = b->spec->print(b->spec->ctx, b->arena + b->off[idx], len, b->pbuf, b->pcap);
        g_trace[k] = strndup(b->pbuf, m);
        r->trace_action[k] = b->parent[idx] == UINT32_MAX ? -1 : b->act[idx];
    }
    if (!b->viol_in_arena) {
        size_t m = b->spec->print(b->spec->ctx, b->viol_state, b->viol_len, b->pbuf, b->pcap);
        g_trace[n] = strndup(b->pbuf, m);
        r->trace_action[n] = b->viol_action;
    }
    r->trace_len = total;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int run_bfs(const or_spec *sp, const or_options *opt, or_result *r) {
    bfs_t b;
    memset(&b, 0, sizeof b);
    memset(r, 0, sizeof *r);
    r->violated_invariant = -1;
    b.spec = sp;
    b.opt = opt;
    b.res = r;
    b.pcap = 1 << 16;
    b.pbuf = malloc(b.pcap);
    trace_clear();
    if (opt->dump_path) {
        b.dump = fopen(opt->dump_path, "w");
        if (!b.dump) { or_set_error("cannot open dump file %s", opt->dump_path); return -1; }
    }
    uint8_t *tmp = malloc(sp->max_state_bytes);
    or_emit em = {&b, on_emit};
    double t0 = now_s();

    /* level 1: initial states */
    b.cur = UINT64_MAX;
    b.cur_level = 1;
    int ni = sp->n_init(sp->ctx);
    for (int k = 0; k < ni; k++) {
        size_t len = sp->init(sp->ctx, k, tmp);
        on_emit(&em, tmp, len, -1, 0);
    }
    uint64_t lo = 0, hi = b.n;
    uint32_t level = 1;
    int stopped_budget = 0;
    while (hi > lo) {
        if (b.have_violation && opt->stop_on_violation) break;
        if (opt->max_levels && level >= opt->max_levels) { stopped_budget = 1; break; }
        if (opt->max_distinct && r->distinct >= opt->max_distinct) { stopped_budget = 1; break; }
        b.cur_level = level + 1;
        for (uint64_t i = lo; i < hi; i++) {
            b.cur = i;
            b.nsucc = 0;
            /* copy: the arena may be reallocated while successors are inserted */
            size_t len = b.off[i + 1] - b.off[i];
            memcpy(tmp, b.arena + b.off[i], len);
            sp->succ(sp->ctx, tmp, len, &em);
            if (b.nsucc == 0 && opt->check_deadlock)
                note_violation(&b, OR_DEADLOCK, -1, i, NULL, 0, -1, 1);
            if (b.have_violation && opt->stop_on_violation == 2) { /* TLC-like: stop at once */
                r->seconds = now_s() - t0;
                r->depth = level + (b.n > hi ? 1 : 0);
                r->queue_left = b.n - (i + 1);
                goto done;
            }
        }
        lo = hi;
        hi = b.n;
        if (hi > lo) level++;
        if (level >= OR_MAX_LEVELS) { or_set_error("too many levels"); break; }
    }
    r->seconds = now_s() - t0;
    r->depth = level;
    r->queue_left = hi - lo;
done:
    if (!b.have_violation) r->verdict = stopped_budget ? OR_BUDGET : OR_OK;
    if (b.have_violation) build_trace(&b);
    r->arena_bytes = b.arena_len;
    if (b.dump) fclose(b.dump);
    free(tmp);
    free(b.pbuf);
    free(b.arena);
    free(b.off);
    free(b.parent);
    free(b.act);
    free(b.tab);
    free(b.viol_state);
    free(b.karena);
    free(b.koff);
    free(b.kbuf);
    return 0;
}

static int run_any(const char *spec, const int64_t *params, int nparams, const or_options *opt, int threads, double max_seconds, or_result *res) {
    or_spec sp;
    memset(&sp, 0, sizeof sp);
    int rc;
    g_err[0] = 0;
    if (!strcmp(spec, "atomic_add")) rc = or_spec_atomic_add(params, nparams, &sp);
    else if (!strcmp(spec, "pcal_intro")) rc = or_spec_pcal_intro(params, nparams, &sp);
    else if (!strcmp(spec, "raft")) rc = or_spec_raft(params, nparams, &sp);
    else if (!strcmp(spec, "ssi")) rc = or_spec_ssi(params, nparams, &sp);
    else if (!strcmp(spec, "paxos")) rc = or_spec_paxos(params, nparams, &sp);
    else { or_set_error("unknown spec '%s'", spec); return -1; }
    if (rc) return rc;
    or_options o = {0, 0, 1, 1, NULL};
    if (opt) o = *opt;
    rc = threads > 0 ? or_run_bfs_mt(&sp, &o, threads, max_seconds, res) : run_bfs(&sp, &o, res);
    free(sp.ctx);
    return rc;
}
int oracle_run(const char *spec, const int64_t *params, int nparams, const or_options *opt, or_result *res) {
    return run_any(spec, params, nparams, opt, 0, 0.0, res);
}
int oracle_run_mt(const char *spec, const int64_t *params, int nparams, const or_options *opt, int threads, double max_seconds, or_result *res) {
    return run_any(spec, params, nparams, opt, threads < 1 ? 1 : threads, max_seconds, res);
}

const char *oracle_action_name(const char *spec, int action) {
    if (action < 0) return "Initial predicate";
    if (!strcmp(spec, "atomic_add")) return or_atomic_add_action(action);
    if (!strcmp(spec, "pcal_intro")) return or_pcal_intro_action(action);
    if (!strcmp(spec, "raft")) return or_raft_action(action);
    if (!strcmp(spec, "ssi")) return or_ssi_action(action);
    if (!strcmp(spec, "paxos")) return or_paxos_action(action);
    return "?";
}
