/*
 * oracle/bfs_mt.c — multi-threaded variant of the CPU oracle's BFS (TEST INFRASTRUCTURE; also the
 * "in-house CPU BFS" that bench.py times as cpu_baseline on all host cores — NOT TLC).
 *
 * Same observable contract as oracle/bfs.c (which restates TLC's BFS: README.md:267-321,
 * testout2:1-266, p-manual §4): level-synchronous breadth-first search, CONSTRAINT filter, invariant /
 * Assert / deadlock checks, the three counters, the depth, per-level distinct counts.  Dedup is EXACT:
 * the seen-set refers to whole canonical state byte strings and compares them (the 64-bit table entries carry a 19-bit
 * tag only to skip most comparisons); a count produced here cannot be off by a hash collision.  The reference prescribes
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

/*
 * Data structures (round 3: the round-2 form — 4096 hash shards, one mutex per shard taken on EVERY insert, arenas grown by
 * realloc, i.e. mremap under the process-wide mmap lock — scaled 4.3x on 256 threads; it measured its locks, not the search):
 *
 *   - one append-only ARENA per worker thread, a fixed virtual reservation (MAP_NORESERVE) that never moves: a state is
 *     written once by its discoverer (2-byte length + bytes), first touched by that thread (NUMA-local), and read by anyone
 *     afterwards without a lock — the frontier of a level IS the tail ranges of the arenas;
 *   - ONE lock-free open-addressing table of 64-bit entries  tag[19] | thread[9] | offset[36]  claimed with a CAS; a probe
 *     compares the tag and only then the bytes (exact dedup: a count produced here cannot be off by a hash collision).  An
 *     inserter appends its candidate to its own arena first and publishes it with the CAS; if it loses the slot to an equal
 *     state it takes the append back (nobody has seen it);
 *   - the table is resized only BETWEEN levels (every thread rehashes a slice), so that no reader ever meets a moving table;
 *   - work of a level: the arenas' frontier ranges cut into blocks of MT_BLOCK states, handed out by one atomic counter.
 */
#include <sys/mman.h>

#define MT_MAX_THREADS 512
#define MT_BLOCK 64u
#define MT_OFF_BITS 36
#define MT_TID_BITS 9
#define MT_TAG_SHIFT (MT_OFF_BITS + MT_TID_BITS)
#define MT_ARENA_BYTES (1ull << MT_OFF_BITS)  /* virtual reservation per thread (64 GiB); touched pages only cost memory */

typedef struct mt_bfs mt_bfs;
typedef struct {
    mt_bfs *b;
    int tid;
    uint8_t *arena;           /* MT_ARENA_BYTES reserved */
    uint64_t len;             /* bytes used (starts at 8: offset 0 is "no entry") */
    uint64_t lvl_lo, lvl_hi;  /* byte range of the states this thread found on the level being expanded */
    uint64_t *blk;            /* offsets of every MT_BLOCK-th state found on the level being filled (frontier blocks of the next) */
    uint64_t nblk, blk_cap, nnew;
    uint64_t *fblk;           /* ... of the level being expanded */
    uint64_t nfblk, fblk_cap;
    uint64_t generated, distinct, nsucc;
    uint64_t max_stat[8];
    int verdict, inv;         /* first violation seen by this thread on the current level (verdict 0 = none) */
} worker_t;

struct mt_bfs {
    const or_spec *spec;
    const or_options *opt;
    _Atomic uint64_t *tab;
    uint64_t tab_cap;         /* power of two */
    _Atomic uint64_t used;    /* entries (approximate while a level runs: per-thread counts are folded in at level ends) */
    atomic_uint_fast64_t next_block;
    uint64_t nblocks;
    uint64_t *blk_base;       /* prefix sums of the threads' frontier block counts */
    int nthreads;
    worker_t *w;
    /* rehash */
    _Atomic uint64_t *old_tab;
    uint64_t old_cap;
    atomic_uint_fast64_t rehash_next;
    atomic_int table_full;
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

static void *mt_reserve(uint64_t bytes, int populate_hint) {
    void *p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return NULL;
    if (populate_hint) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

static inline const uint8_t *mt_state_at(const mt_bfs *b, uint64_t entry, size_t *len) {
    const worker_t *o = &b->w[(entry >> MT_OFF_BITS) & ((1u << MT_TID_BITS) - 1)];
    const uint8_t *p = o->arena + (entry & ((1ull << MT_OFF_BITS) - 1));
    uint16_t l;
    memcpy(&l, p, 2);
    *len = l;
    return p + 2;
}

/* 1 = new (this thread's arena now holds it), 0 = already present */
static int mt_insert(worker_t *w, uint64_t hash, const uint8_t *st, size_t len) {
    mt_bfs *b = w->b;
    const uint64_t mask = b->tab_cap - 1, tag = (hash >> (64 - 19)) << MT_TAG_SHIFT;
    uint64_t h = hash & mask, mine = 0;
    for (uint64_t probes = 0;; probes++) {
        uint64_t e = atomic_load_explicit(&b->tab[h], memory_order_acquire);
        if (e == 0) {
            if (!mine) {  /* append the candidate to this thread's arena (not yet visible to anyone) */
                if (w->len + 2 + len > MT_ARENA_BYTES) oom("a thread's arena reservation");
                uint16_t l16 = (uint16_t)len;
                memcpy(w->arena + w->len, &l16, 2);
                memcpy(w->arena + w->len + 2, st, len);
                mine = tag | ((uint64_t)w->tid << MT_OFF_BITS) | w->len;
            }
            uint64_t expect = 0;
            if (atomic_compare_exchange_strong_explicit(&b->tab[h], &expect, mine, memory_order_acq_rel, memory_order_acquire)) {
                if ((w->nnew % MT_BLOCK) == 0) {
                    if (w->nblk == w->blk_cap) {
                        w->blk_cap = w->blk_cap ? w->blk_cap * 2 : 1024;
                        w->blk = realloc(w->blk, w->blk_cap * sizeof *w->blk);
                        if (!w->blk) oom("block index");
                    }
                    w->blk[w->nblk++] = w->len;
                }
                w->nnew++;
                w->len += 2 + len;
                return 1;
            }
            e = expect;  /* somebody took the slot: is it the same state? */
        }
        if ((e >> MT_TAG_SHIFT) == (tag >> MT_TAG_SHIFT)) {
            size_t ol;
            const uint8_t *os = mt_state_at(b, e, &ol);
            if (ol == len && memcmp(os, st, len) == 0) return 0;  /* (a tentative append is simply overwritten by the next one) */
        }
        h = (h + 1) & mask;
        if (probes > mask) { atomic_store(&b->table_full, 1); return 0; }
    }
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
        is_new = mt_insert(w, mt_hash(s, len), s, len);
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

/* one level: blocks of MT_BLOCK frontier states, whoever found them */
static void *mt_level_worker(void *arg) {
    worker_t *w = arg;
    mt_bfs *b = w->b;
    const or_spec *sp = b->spec;
    or_emit em = {w, mt_emit};
    for (;;) {
        const uint64_t g = atomic_fetch_add(&b->next_block, 1);
        if (g >= b->nblocks || atomic_load(&b->table_full)) break;
        int t = 0;
        while (g >= b->blk_base[t + 1]) t++;  /* owner of block g */
        const worker_t *o = &b->w[t];
        const uint64_t k = g - b->blk_base[t];
        uint64_t off = o->fblk[k];
        const uint64_t end = k + 1 < o->nfblk ? o->fblk[k + 1] : o->lvl_hi;
        while (off < end) {
            uint16_t l;
            memcpy(&l, o->arena + off, 2);
            w->nsucc = 0;
            sp->succ(sp->ctx, o->arena + off + 2, l, &em);
            if (w->nsucc == 0 && b->opt->check_deadlock) mt_note(w, OR_DEADLOCK, -1);
            off += 2 + (uint64_t)l;
        }
    }
    return NULL;
}

/* between levels: every thread moves a slice of the old table into the new one */
static void *mt_rehash_worker(void *arg) {
    worker_t *w = arg;
    mt_bfs *b = w->b;
    const uint64_t slice = 1u << 16, mask = b->tab_cap - 1;
    for (;;) {
        const uint64_t lo = atomic_fetch_add(&b->rehash_next, slice);
        if (lo >= b->old_cap) break;
        const uint64_t hi = lo + slice < b->old_cap ? lo + slice : b->old_cap;
        for (uint64_t i = lo; i < hi; i++) {
            const uint64_t e = atomic_load_explicit(&b->old_tab[i], memory_order_relaxed);
            if (!e) continue;
            size_t len;
            const uint8_t *st = mt_state_at(b, e, &len);
            uint64_t h = mt_hash(st, len) & mask;
            for (;;) {
                uint64_t expect = 0;
                if (atomic_compare_exchange_strong_explicit(&b->tab[h], &expect, e, memory_order_acq_rel, memory_order_relaxed)) break;
                h = (h + 1) & mask;
            }
        }
    }
    return NULL;
}

static double mt_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void mt_run_threads(mt_bfs *b, pthread_t *th, void *(*fn)(void *)) {
    for (int t = 0; t < b->nthreads; t++) pthread_create(&th[t], NULL, fn, &b->w[t]);
    for (int t = 0; t < b->nthreads; t++) pthread_join(th[t], NULL);
}

int or_run_bfs_mt(const or_spec *sp, const or_options *opt, int nthreads, double max_seconds, or_result *r) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > MT_MAX_THREADS) nthreads = MT_MAX_THREADS;
    if (sp->canon) { or_set_error("the multi-threaded oracle does not implement first-met SYMMETRY representatives"); return -1; }
    if (opt->stop_on_violation == 2 || opt->dump_path) { or_set_error("the multi-threaded oracle has no stop-at-once mode and no dump"); return -1; }
    if (sp->max_state_bytes > 65535) { or_set_error("the multi-threaded oracle stores state lengths in 16 bits"); return -1; }
    mt_bfs b;
    memset(&b, 0, sizeof b);
    memset(r, 0, sizeof *r);
    r->violated_invariant = -1;
    b.spec = sp;
    b.opt = opt;
    b.nthreads = nthreads;
    b.w = calloc((size_t)nthreads, sizeof *b.w);
    b.blk_base = calloc((size_t)nthreads + 1, sizeof *b.blk_base);
    if (!b.w || !b.blk_base) oom("setup");
    for (int t = 0; t < nthreads; t++) {
        b.w[t].b = &b;
        b.w[t].tid = t;
        b.w[t].arena = mt_reserve(MT_ARENA_BYTES, 1);
        if (!b.w[t].arena) { or_set_error("cannot reserve %d thread arenas of %llu GiB of address space", nthreads, (unsigned long long)(MT_ARENA_BYTES >> 30)); return -1; }
        b.w[t].len = b.w[t].lvl_lo = b.w[t].lvl_hi = 8;
    }
    b.tab_cap = 1ull << 22;
    b.tab = mt_reserve(b.tab_cap * sizeof(uint64_t), 1);
    if (!b.tab) oom("table");
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
    int budget = 0, have_violation = 0, rc = 0;
    uint64_t frontier = 0, grow = 4;
    for (;;) {
        /* close the level: gather the workers' counters, turn what every thread found into its frontier */
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
            w->lvl_lo = w->lvl_hi;
            w->lvl_hi = w->len;
            uint64_t *sw = w->fblk; w->fblk = w->blk; w->blk = sw;
            uint64_t sc = w->fblk_cap; w->fblk_cap = w->blk_cap; w->blk_cap = sc;
            w->nfblk = w->nblk;
            w->nblk = 0;
            w->nnew = 0;
            b.blk_base[t + 1] = b.blk_base[t] + w->nfblk;
        }
        if (atomic_load(&b.table_full)) { or_set_error("the multi-threaded oracle's table filled up inside a level"); rc = -1; break; }
        b.nblocks = b.blk_base[nthreads];
        frontier = newd;
        r->distinct += newd;
        if (frontier > 0 && r->distinct > frontier) { const uint64_t g = r->distinct / (r->distinct - frontier) + 2; if (g > grow) grow = g < 16 ? g : 16; }
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
        /* room for the next level at a load below 1/2: what is there + the frontier times the largest growth seen so far */
        uint64_t need = 2 * (r->distinct + frontier * grow);
        if (need > b.tab_cap) {
            b.old_tab = b.tab;
            b.old_cap = b.tab_cap;
            while (b.tab_cap < need) b.tab_cap *= 2;
            b.tab = mt_reserve(b.tab_cap * sizeof(uint64_t), 1);
            if (!b.tab) oom("table");
            atomic_store(&b.rehash_next, 0);
            mt_run_threads(&b, th, mt_rehash_worker);
            munmap((void *)b.old_tab, b.old_cap * sizeof(uint64_t));
        }
        level++;
        atomic_store(&b.next_block, 0);
        mt_run_threads(&b, th, mt_level_worker);
    }
    r->seconds = mt_now() - t0;
    r->depth = level;
    r->queue_left = frontier;
    if (!have_violation) r->verdict = budget ? OR_BUDGET : OR_OK;
    for (int t = 0; t < nthreads; t++) {
        r->arena_bytes += b.w[t].len - 8;
        munmap(b.w[t].arena, MT_ARENA_BYTES);
        free(b.w[t].blk);
        free(b.w[t].fblk);
    }
    munmap((void *)b.tab, b.tab_cap * sizeof(uint64_t));
    free(b.blk_base);
    free(b.w);
    free(th);
    return rc;
}
