/*
 * oracle/bfs_mt.c — multi-threaded variant of the CPU oracle's BFS (TEST INFRASTRUCTURE; also the
 * "in-house CPU BFS" that bench.py times as cpu_baseline on all host cores — NOT TLC).
 *
 * Same observable contract as oracle/bfs.c (which restates TLC's BFS: README.md:267-321,
 * testout2:1-266, p-manual §4): level-synchronous breadth-first search, CONSTRAINT filter, invariant /
 * Assert / deadlock checks, the three counters, the depth, per-level distinct counts.  Dedup is EXACT:
 * the seen-set stores whole canonical state byte strings, sharded by hash over NSHARD independently
 * locked stores; a count produced here cannot be off by a hash collision.  The reference prescribes
 * TLC with "Number of worker threads: Use the number of cpu cores"
 * (examples/serializableSnapshotIsolation.tla:52-53); this is the stand-in where no JVM exists.
 *
 * Differences from bfs.c: no parent pointers, hence no counterexample trace (a violation is reported
 * by verdict + invariant index only; the single-threaded oracle rebuilds traces), no state dump, and
 * stop_on_violation == 2 (TLC's stop-at-once order) is not available.
 */
#include "oracle_int.h"
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MT_SHARD_BITS 12
#define MT_NSHARD (1u << MT_SHARD_BITS)

typedef struct {
    pthread_mutex_t mu;
    uint8_t *arena;
    uint64_t arena_len, arena_cap;
    uint64_t *off;        /* n+1 offsets */
    uint32_t n, ncap;
    uint32_t *tab;        /* open addressing over state indices + 1 */
    uint32_t tab_cap;     /* power of two */
    uint32_t lvl_lo, lvl_hi; /* frontier of the level being expanded */
} shard_t;

typedef struct mt_bfs mt_bfs;
typedef struct {
    mt_bfs *b;
    uint64_t generated, distinct, nsucc;
    uint64_t max_stat[8];
    int verdict, inv;     /* first violation seen by this thread on the current level (verdict 0 = none) */
    uint8_t *copy;        /* private copy of one shard's frontier slice */
    uint64_t copy_cap;
    uint64_t *copy_off;
    uint32_t copy_off_cap;
} worker_t;

struct mt_bfs {
    const or_spec *spec;
    const or_options *opt;
    shard_t *sh;
    atomic_uint next_shard;
    int nthreads;
    worker_t *w;
};

static uint64_t mt_hash(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull ^ (n * 0x9e3779b97f4a7c15ull);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t x;
        memcpy(&x, p + i, 8);
        h = (h ^ x) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    h ^= h >> 32;
    h *= 0xd6e8feb86659fd93ull;
    h ^= h >> 32;
    return h;
}

static void oom(const char *what) { fprintf(stderr, "oracle(mt): out of memory (%s)\n", what); abort(); }

/* caller holds the shard's lock */
static void shard_grow_table(shard_t *s) {
    uint32_t ncap = s->tab_cap ? s->tab_cap * 2 : 64;
    uint32_t *nt = calloc(ncap, sizeof *nt);
    if (!nt) oom("table");
    for (uint32_t i = 0; i < s->n; i++) {
        /* the low bits of the hash index the table, the high bits chose the shard */
        uint64_t h = mt_hash(s->arena + s->off[i], s->off[i + 1] - s->off[i]) & (ncap - 1);
        while (nt[h]) h = (h + 1) & (ncap - 1);
        nt[h] = i + 1;
    }
    free(s->tab);
    s->tab = nt;
    s->tab_cap = ncap;
}

/* 1 = new (appended), 0 = already present */
static int shard_insert(shard_t *s, uint64_t hash, const uint8_t *st, size_t len) {
    int is_new = 0;
    pthread_mutex_lock(&s->mu);
    if ((uint64_t)(s->n + 1) * 2 > s->tab_cap) shard_grow_table(s);
    uint64_t mask = s->tab_cap - 1, h = hash & mask;
    for (;;) {
        uint32_t e = s->tab[h];
        if (!e) break;
        uint64_t o = s->off[e - 1];
        if (s->off[e] - o == len && memcmp(s->arena + o, st, len) == 0) goto out;
        h = (h + 1) & mask;
    }
    if (s->n + 2 > s->ncap) {
        s->ncap = s->ncap ? s->ncap * 2 : 32;
        s->off = realloc(s->off, ((size_t)s->ncap + 1) * sizeof *s->off);
        if (!s->off) oom("offsets");
        if (s->n == 0) s->off[0] = 0;
    }
    if (s->arena_len + len > s->arena_cap) {
        while (s->arena_len + len > s->arena_cap) s->arena_cap = s->arena_cap ? s->arena_cap + s->arena_cap / 2 : 4096;
        s->arena = realloc(s->arena, s->arena_cap);
        if (!s->arena) oom("arena");
    }
    memcpy(s->arena + s->arena_len, st, len);
    s->arena_len += len;
    s->off[s->n + 1] = s->arena_len;
    s->tab[h] = s->n + 1;
    s->n++;
    is_new = 1;
out:
    pthread_mutex_unlock(&s->mu);
    return is_new;
}

static void mt_note(worker_t *w, int verdict, int inv) {
    /* keep the smallest (verdict, invariant) pair so that the report does not depend on thread timing */
    if (!w->verdict || verdict < w->verdict || (verdict == w->verdict && inv < w->inv)) { w->verdict = verdict; w->inv = inv; }
}

static void mt_emit(or_emit *em, const uint8_t *s, size_t len, int action, unsigned flags) {
    worker_t *w = em->bfs;
    const or_spec *sp = w->b->spec;
    (void)action;
    w->nsucc++;
    w->generated++;
    if (flags & OR_FLAG_SPECERR) { mt_note(w, OR_SPEC_ERROR, -1); return; }
    if (flags & OR_FLAG_ASSERT) { mt_note(w, OR_ASSERT, -1); return; }
    if (flags & OR_FLAG_PROPERTY) mt_note(w, OR_INVARIANT, (int)(flags >> 8));
    int inmodel = sp->constraint ? sp->constraint(sp->ctx, s, len) : 1;
    int is_new = 0;
    if (inmodel) {
        uint64_t h = mt_hash(s, len);
        is_new = shard_insert(&w->b->sh[h >> (64 - MT_SHARD_BITS)], h, s, len);
        if (is_new) {
            w->distinct++;
            if (sp->stats) sp->stats(sp->ctx, s, len, w->max_stat);
        }
    }
    if ((is_new || !inmodel) && sp->invariant) {
        int inv = sp->invariant(sp->ctx, s, len);
        if (inv >= 0) mt_note(w, OR_INVARIANT, inv);
    }
}

static void *mt_level_worker(void *arg) {
    worker_t *w = arg;
    mt_bfs *b = w->b;
    const or_spec *sp = b->spec;
    or_emit em = {w, mt_emit};
    for (;;) {
        unsigned si = atomic_fetch_add(&b->next_shard, 1u);
        if (si >= MT_NSHARD) break;
        shard_t *s = &b->sh[si];
        if (s->lvl_hi == s->lvl_lo) continue;
        /* private copy of the shard's frontier slice: other threads append to (and may reallocate) the shard meanwhile */
        pthread_mutex_lock(&s->mu);
        uint32_t cnt = s->lvl_hi - s->lvl_lo;
        uint64_t o0 = s->off[s->lvl_lo], bytes = s->off[s->lvl_hi] - o0;
        if (bytes > w->copy_cap) { w->copy_cap = bytes * 2; free(w->copy); w->copy = malloc(w->copy_cap); if (!w->copy) oom("copy"); }
        if (cnt + 1 > w->copy_off_cap) { w->copy_off_cap = (cnt + 1) * 2; free(w->copy_off); w->copy_off = malloc(w->copy_off_cap * sizeof *w->copy_off); if (!w->copy_off) oom("copy"); }
        memcpy(w->copy, s->arena + o0, bytes);
        for (uint32_t k = 0; k <= cnt; k++) w->copy_off[k] = s->off[s->lvl_lo + k] - o0;
        pthread_mutex_unlock(&s->mu);
        for (uint32_t k = 0; k < cnt; k++) {
            w->nsucc = 0;
            sp->succ(sp->ctx, w->copy + w->copy_off[k], w->copy_off[k + 1] - w->copy_off[k], &em);
            if (w->nsucc == 0 && b->opt->check_deadlock) mt_note(w, OR_DEADLOCK, -1);
        }
    }
    return NULL;
}

static double mt_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int or_run_bfs_mt(const or_spec *sp, const or_options *opt, int nthreads, double max_seconds, or_result *r) {
    if (nthreads < 1) nthreads = 1;
    if (sp->canon) { or_set_error("the multi-threaded oracle does not implement first-met SYMMETRY representatives"); return -1; }
    if (opt->stop_on_violation == 2 || opt->dump_path) { or_set_error("the multi-threaded oracle has no stop-at-once mode and no dump"); return -1; }
    mt_bfs b;
    memset(&b, 0, sizeof b);
    memset(r, 0, sizeof *r);
    r->violated_invariant = -1;
    b.spec = sp;
    b.opt = opt;
    b.nthreads = nthreads;
    b.sh = calloc(MT_NSHARD, sizeof *b.sh);
    b.w = calloc((size_t)nthreads, sizeof *b.w);
    if (!b.sh || !b.w) oom("setup");
    for (unsigned i = 0; i < MT_NSHARD; i++) pthread_mutex_init(&b.sh[i].mu, NULL);
    for (int t = 0; t < nthreads; t++) b.w[t].b = &b;
    pthread_t *th = calloc((size_t)nthreads, sizeof *th);
    double t0 = mt_now();

    /* level 1: initial states (one thread) */
    uint8_t *tmp = malloc(sp->max_state_bytes);
    or_emit em = {&b.w[0], mt_emit};
    int ni = sp->n_init(sp->ctx);
    for (int k = 0; k < ni; k++) {
        size_t len = sp->init(sp->ctx, k, tmp);
        mt_emit(&em, tmp, len, -1, 0);
    }
    free(tmp);
    uint32_t level = 1;
    int budget = 0, have_violation = 0;
    uint64_t frontier = 0;
    for (;;) {
        /* close the level: gather the workers' counters, advance every shard's frontier */
        uint64_t newd = 0;
        for (int t = 0; t < nthreads; t++) {
            worker_t *w = &b.w[t];
            r->generated += w->generated;
            if (level >= 2) r->level_generated[level - 2] += w->generated;
            newd += w->distinct;
            w->generated = w->distinct = 0;
            for (int q = 0; q < 8; q++) if (w->max_stat[q] > r->max_stat[q]) r->max_stat[q] = w->max_stat[q];
            if (w->verdict && (!have_violation || w->verdict < r->verdict || (w->verdict == r->verdict && w->inv < r->violated_invariant))) {
                have_violation = 1;
                r->verdict = w->verdict;
                r->violated_invariant = w->inv;
            }
        }
        frontier = 0;
        for (unsigned i = 0; i < MT_NSHARD; i++) {
            b.sh[i].lvl_lo = b.sh[i].lvl_hi;
            b.sh[i].lvl_hi = b.sh[i].n;
            frontier += b.sh[i].lvl_hi - b.sh[i].lvl_lo;
        }
        r->distinct += newd;
        if (newd) {
            if (level - 1 < OR_MAX_LEVELS) r->level_distinct[level - 1] = newd;
        } else {
            level--; /* the level just expanded produced nothing new: it was the last */
        }
        if (frontier == 0) break;
        if (have_violation && opt->stop_on_violation) break;
        if (opt->max_levels && level >= opt->max_levels) { budget = 1; break; }
        if (opt->max_distinct && r->distinct >= opt->max_distinct) { budget = 1; break; }
        if (max_seconds > 0 && mt_now() - t0 >= max_seconds) { budget = 1; break; }
        if (level + 1 >= OR_MAX_LEVELS) { or_set_error("too many levels"); break; }
        level++;
        atomic_store(&b.next_shard, 0u);
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, mt_level_worker, &b.w[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    r->seconds = mt_now() - t0;
    r->depth = level;
    r->queue_left = frontier;
    if (!have_violation) r->verdict = budget ? OR_BUDGET : OR_OK;
    for (unsigned i = 0; i < MT_NSHARD; i++) {
        r->arena_bytes += b.sh[i].arena_len;
        free(b.sh[i].arena);
        free(b.sh[i].off);
        free(b.sh[i].tab);
        pthread_mutex_destroy(&b.sh[i].mu);
    }
    for (int t = 0; t < nthreads; t++) { free(b.w[t].copy); free(b.w[t].copy_off); }
    free(b.sh);
    free(b.w);
    free(th);
    return 0;
}
