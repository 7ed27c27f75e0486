/*
 * oracle/oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT THE PRODUCT PATH).
 *
 * A plain-C, single-threaded, exact-dedup (full state bytes, no fingerprints)
 * breadth-first model checker for the hand-lowered specs of spacejam/tla-rust.
 * It restates, on the CPU and with an unpacked "obviously correct" state
 * representation, what the external TLC tool does for `make test`
 * (reference Makefile:6-7, README.md:262): enumerate Init, expand Next level
 * by level, apply CONSTRAINT, check Assert / INVARIANT / deadlock, count
 * "states generated / distinct states found / depth".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or execute this code.  The product (tla_rust_amd/, libtlamc.so) never
 * does.
 *
 * PARITY STATUS: "parity unpinned" against TLC for every complete count
 * (no JVM / tla2tools.jar in this image).  Pinned against the
 * only golden material the reference holds for this path:
 *   - README.md:267-321 (assert text, 6-state shortest counterexample ending
 *     alice_account = -1, 9097/6164/999 partial counters as bounds, depth 7);
 *   - README.md:349-352 (committed pcal_intro must pass);
 * and against the survey-derived regression anchors of BASELINE.md §2.
 *
 * Counting conventions (SURVEY.md §7 hard part 2):
 *   generated = initial states + every successor produced by an enabled
 *               (disjunct, witness) pair of Next, duplicates and self loops
 *               included;
 *   distinct  = states that satisfy the CONSTRAINT and were not seen before;
 *   a successor that violates the CONSTRAINT is generated, invariant-checked,
 *   but neither stored nor expanded (FIFO/MCInnerFIFO.cfg:23-26, p-manual §4.3);
 *   depth     = number of BFS levels, the initial states being level 1
 *               (README.md:320).
 */
#ifndef TLA_ORACLE_H
#define TLA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    OR_OK = 0,            /* Model checking completed. No error has been found. */
    OR_INVARIANT = 1,     /* an INVARIANT is violated                          */
    OR_ASSERT = 2,        /* first argument of Assert evaluated to FALSE       */
    OR_DEADLOCK = 3,      /* a reachable state has no successor at all         */
    OR_SPEC_ERROR = 4,    /* TLC would raise an evaluation error               */
    OR_BUDGET = 5         /* stopped by max_levels / max_distinct              */
};

#define OR_MAX_LEVELS 4096
#define OR_MAX_TRACE 4096

typedef struct {
    uint64_t max_levels;    /* 0 = unlimited; stop after expanding this many levels         */
    uint64_t max_distinct;  /* 0 = unlimited; stop after the level whose cumulative D >= it */
    int check_deadlock;     /* TLC default: on                                              */
    int stop_on_violation;  /* 1: stop at the end of the level that found the violation; 2: at once (TLC-like) */
    const char *dump_path;  /* if set: one line per distinct state, "L<level> <text>"       */
} or_options;

typedef struct {
    uint64_t distinct, generated, queue_left;
    uint32_t depth;             /* number of levels reached (TLC2 "depth")              */
    int verdict;                /* OR_*                                                 */
    int violated_invariant;     /* index into the spec's invariant list, or -1          */
    uint64_t level_distinct[OR_MAX_LEVELS];   /* new distinct states per level (1-based at [0]) */
    uint64_t level_generated[OR_MAX_LEVELS];  /* successors generated while expanding level     */
    uint32_t trace_len;         /* number of states in the counterexample (0 if none)   */
    int trace_action[OR_MAX_TRACE]; /* action id that produced state k (-1 for initial) */
    double seconds;             /* BFS wall time (init -> last level)                   */
    uint64_t max_stat[8];       /* spec-specific maxima (raft: msg domain, elections, allLogs, inflight) */
    uint64_t arena_bytes;
} or_result;

/* spec = "atomic_add" | "pcal_intro" | "raft" | "ssi"; params = spec-specific int list:
 *   atomic_add : {N adders}
 *   pcal_intro : {variant (0 = committed file, 1 = README variant with labels A/B),
 *                 check MoneyInvariant (0/1), MaxMoney (20), nproc (2)}
 *   ssi        : {nTxn, nKey, invariant mask (127 = all), find}  — see spec_ssi.c
 *   raft       : {nServer, MaxClientRequests, MaxTerm, MaxLogLen, MaxMsgs, invariant mask
 *                 (bit0 NoTwoLeaders, bit1 CommittedLogStable), naive_commit (0; 1 = the
 *                 "obvious but wrong" lowering of raft.tla:392-402, a negative control)}
 */
int oracle_run(const char *spec, const int64_t *params, int nparams,
               const or_options *opt, or_result *res);

/* Multi-threaded variant (oracle/bfs_mt.c): same specs, same counting conventions, exact dedup over a
 * hash-sharded seen-set, `threads` worker threads per BFS level.  Counts, depth, per-level counts and the
 * verdict only (no counterexample trace).  max_seconds > 0: stop (verdict budget) after the first level
 * that ends later than that.  This is also the "in-house CPU BFS, all host cores - not TLC" baseline that
 * bench.py reports beside the GPU number. */
int oracle_run_mt(const char *spec, const int64_t *params, int nparams,
                  const or_options *opt, int threads, double max_seconds, or_result *res);

/* text of the k-th state of the last counterexample trace (valid until next oracle_run) */
const char *oracle_trace_state(uint32_t k);
const char *oracle_action_name(const char *spec, int action);
const char *oracle_last_error(void);
/* Voting census (examples/Paxos/MCVoting.cfg:7-8's alternative configurations): out = {type-correct states, states satisfying Inv,
 * successors generated from those, successors violating Inv}; params as for spec "paxos" with kind = 1 */
int oracle_voting_census(const int64_t *params, int nparams, uint64_t out[4]);
/* the in-spec unit tests of serializableSnapshotIsolation.tla:1068-1077,1184-1205; returns #failures */
int oracle_ssi_unit_tests(void);

#ifdef __cplusplus
}
#endif
#endif
