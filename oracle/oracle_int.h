/* oracle/oracle_int.h — internal spec interface of the CPU oracle (test infrastructure). */
#ifndef TLA_ORACLE_INT_H
#define TLA_ORACLE_INT_H
#include "oracle.h"

#define OR_FLAG_ASSERT 1u      /* Assert(...) failed while generating this successor */
#define OR_FLAG_SPECERR 2u     /* TLC would have raised an evaluation error           */
#define OR_FLAG_PROPERTY 4u    /* this TRANSITION violates the safety part of a PROPERTY; bits 8.. = the index reported */

typedef struct or_emit {
    void *bfs;                 /* opaque */
    void (*emit)(struct or_emit *, const uint8_t *s, size_t len, int action, unsigned flags);
} or_emit;

typedef struct or_spec {
    const char *name;
    void *ctx;
    size_t max_state_bytes;
    int (*n_init)(void *ctx);
    size_t (*init)(void *ctx, int k, uint8_t *out);
    void (*succ)(void *ctx, const uint8_t *s, size_t len, or_emit *em);
    int (*constraint)(void *ctx, const uint8_t *s, size_t len);       /* 1 = inside the model */
    int (*invariant)(void *ctx, const uint8_t *s, size_t len);        /* -1 ok, else index    */
    size_t (*print)(void *ctx, const uint8_t *s, size_t len, char *buf, size_t cap);
    const char *(*action_name)(int action);
    void (*stats)(void *ctx, const uint8_t *s, size_t len, uint64_t *max_stat); /* optional */
    /* optional: the seen-set compares canon(s) instead of s (SYMMETRY with TLC's first-met representative: the state that is
     * stored and later expanded is the one that was generated, the ORBIT decides whether it is new) */
    size_t (*canon)(void *ctx, const uint8_t *s, size_t len, uint8_t *out);
} or_spec;

int or_spec_atomic_add(const int64_t *p, int np, or_spec *out);
int or_spec_pcal_intro(const int64_t *p, int np, or_spec *out);
int or_spec_raft(const int64_t *p, int np, or_spec *out);
int or_spec_ssi(const int64_t *p, int np, or_spec *out);
int or_spec_paxos(const int64_t *p, int np, or_spec *out);
const char *or_atomic_add_action(int a);
const char *or_pcal_intro_action(int a);
const char *or_raft_action(int a);
const char *or_ssi_action(int a);
const char *or_paxos_action(int a);

void or_set_error(const char *fmt, ...);
/* bfs_mt.c: the multi-threaded BFS (counts and verdict only; max_seconds > 0 stops after the level that exceeds it) */
int or_run_bfs_mt(const or_spec *sp, const or_options *opt, int nthreads, double max_seconds, or_result *r);

#endif
