/*
 * include/tlamc.h — C ABI of libtlamc.so, the MI355X-native explicit-state model checker
 * for the spacejam/tla-rust specs.
 *
 * The reference has no FFI of its own: its boundary is the command line `tlc X.tla` with
 * `X.cfg` beside it (reference Makefile:6-7, README.md:262,356) and TLC's stdout report
 * (README.md:267-321).  BASELINE.json:north_star asks for a host (Rust in intent) that calls
 * HIP "through a thin C-ABI FFI"; this header IS that FFI.  Every entry point below names the
 * part of the reference's workflow it replaces.  Plain pointers and sizes only; the caller
 * owns every *out buffer, the engine owns all device memory; no function throws or aborts
 * across the boundary: each returns 0 or a negative MC_E* code (mc_strerror()).
 *
 * There is no CPU fallback behind this ABI: every function that computes needs a HIP device
 * (gfx950) and fails with MC_EHIP without one.
 */
#ifndef TLAMC_H
#define TLAMC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ error codes */
#define MC_OK 0
#define MC_EBADCFG (-1)    /* bad spec id / constants / configuration                        */
#define MC_EHIP (-2)       /* HIP runtime error or no device                                 */
#define MC_EOVERFLOW (-3)  /* a fixed-capacity slot array of the packed state overflowed     */
#define MC_ETABLEFULL (-4) /* seen-set load factor exceeded                                  */
#define MC_EARENA (-5)     /* state arena (frontier storage) exhausted                       */
#define MC_ERCCL (-6)      /* exchange failure in sharded mode                               */
#define MC_ESTATE (-7)     /* call sequence error                                            */
#define MC_EPARSE (-8)     /* .cfg / .tla front-end error                                    */
#define MC_ENOSPEC (-9)    /* module is not one of the lowered specs (or its text changed)   */
#define MC_EROUTE (-10)    /* sharded mode: an exchange bucket of a round was too small for the candidates it had to hold
                            * (mc_shard_opts.packed_fanout / move_fanout); nothing was truncated.  mc_shard_run* does not hand
                            * this to its caller until it has restarted the search with twice the allowance (mc_shard_stats.restarts) */

/* ------------------------------------------------------------------ verdicts (TLC's outcomes) */
#define MC_V_OK 0          /* "Model checking completed. No error has been found." (testout2:260) */
#define MC_V_INVARIANT 1   /* an INVARIANT of the cfg is violated (pcal_intro.cfg:3)          */
#define MC_V_ASSERT 2      /* "The first argument of Assert evaluated to FALSE" (README.md:268) */
#define MC_V_DEADLOCK 3    /* p-manual §4.7.1 p.41                                            */
#define MC_V_SPECERR 4     /* TLC would raise an evaluation error (index out of domain)       */
#define MC_V_BUDGET 5      /* stopped by max_levels / max_distinct                            */
#define MC_V_ASSUME 6      /* an ASSUME of the module is false (TLC's "No Behavior Spec" mode: a cfg that names neither SPECIFICATION
                            * nor INIT / NEXT — SpecifyingSystems/SimpleMath/SimpleMath.cfg — makes mc_check_files evaluate the ASSUMEs) */

/* ------------------------------------------------------------------ lowered specs */
#define MC_SPEC_ATOMIC_ADD 1 /* reference atomic_add.tla:4-23, N adders + checker; params {N}          */
#define MC_SPEC_PCAL_INTRO 2 /* reference pcal_intro.tla:4-23; params {variant, checkInv, MaxMoney, P}  */
#define MC_SPEC_RAFT 3       /* reference examples/raft.tla:110-507 under specs/MCraft.tla;
                                params {nServer, MaxClientRequests, MaxTerm, MaxLogLen, MaxMsgs,
                                        invariantMask (1 NoTwoLeaders | 2 CommittedLogStable),
                                        [6..8] capacities of the messages / elections / allLogs slot arrays of the
                                        packed state (0 = default 40 / 4 / 16; exceeding one is MC_EOVERFLOW),
                                        [9] MaxMsgKeys (0 = unbounded)}                             */

#define MC_SPEC_SSI 4        /* reference examples/serializableSnapshotIsolation.tla:219-996 under specs/MCssi.tla;
                                params {nTxn <= 4, nKey <= 3, invariantMask (1 WellFormed | 2 HoldingXLocks |
                                4 WaitingForXLock | 8 CorrectReadView | 16 FirstCommitterWins | 32 Cahill |
                                64 Bernstein), find (0; 1..6 abort reason, 7 two waiters: ":81-96 expected"),
                                textbook (1 = examples/textbookSnapshotIsolation.tla: no Cahill variables)}  */

#define MC_SPEC_PCAL 5       /* a PlusCal algorithm compiled by mc_program_compile (any module of the supported subset,
                                the reference's untranslated pcal_intro.tla:4-19 and atomic_add.tla:4-23 included);
                                params {(int64) mc_program handle} — build the descriptor with mc_program_spec      */

#define MC_SPEC_PAXOS 6      /* reference examples/Paxos/Voting.tla:133-160 and Paxos.tla:93-208 under MCVoting.tla / MCPaxos.tla
                                (+ .cfg: INVARIANT, PROPERTY C!Spec / V!Spec as a per-transition refinement check, SYMMETRY);
                                params {kind (0 Paxos, 1 Voting), nAcceptor <= 4, nValue <= 3, nBallot = MaxBallot + 1 <= 4,
                                invariantMask (Paxos: bit k = Inv!(k+1), MCPaxos.tla:65-68; Voting: 1 = Inv),
                                symmetry (1 Permutations(Acceptor) | 2 Permutations(Value)),
                                property (1 = check the cfg's PROPERTY; a violation is reported as invariant index 4 (Paxos) /
                                1 (Voting); 2 = negative control: Phase2a without its quorum conjunct),
                                nQuorum (0 = all majorities of minimal size), quorum acceptor bit masks ...}            */

typedef struct {
    uint32_t spec_id;
    uint32_t nparams;
    int64_t params[16];
} mc_spec_desc;

#define MC_F_DEADLOCK 1u /* check deadlock (TLC default on; serializableSnapshotIsolation.tla:57) */
#define MC_F_TRACE 2u    /* keep (parent, action) per state so a counterexample can be rebuilt     */
#define MC_F_TIMING 4u   /* time every kernel launch with HIP events on the engine stream          */
#define MC_F_MATRIX 8u   /* A/B only: unfused expand -> candidate matrix -> insert kernels          */
#define MC_F_NOPROBE 16u /* profiling only (mc_engine_debug_reexpand): skip the seen-set probes   */
#define MC_F_NOFAMILY 32u /* A/B only: expand raft slot by slot instead of by action family        */
#define MC_F_OCC3 2048u      /* A/B only: the by-family expand kernel compiled for 3 wavefronts per SIMD (146 VGPRs, no spills) instead of 4 */
#define MC_F_NOINWAVE 65536u /* A/B only: every new state through the new-list and k_materialise (rounds 1-3) instead of being written by
                                the expand wavefront that found it (round 4; by-family kernels of a fused run) */
#define MC_F_WAVETAIL 131072u /* A/B only: in-wave writes by wavefront (no workgroup barrier) instead of by workgroup */
#define MC_F_PARK 32768u /* measured, not the default (round 6): a by-family wavefront whose survivor list fills up PARKS the overflow in the
                           * new-list's memory and the workgroup's own tail writes it in later rounds (k_expand_family<.., PARK>), instead of
                           * sending it through the new-list to k_materialise: every state is written in-wave, the step takes as long */
#define MC_F_NOFILTER 8192u  /* A/B only: by-family expand kernel without the per-wavefront duplicate filter in front of the seen-set */
#define MC_F_SYNCPROBE 4096u /* A/B only: the by-family expand kernel waits for every seen-set probe where it issues it (rounds 1-4) instead of
                              * resolving a batch of probes one batch later (round 5: split-phase probes, engine.hip MC_ASYNC_PROBE) */
#define MC_F_PROGRESS 16384u /* mc_check_files: print TLC's "Progress(d): ..." lines (testout2:4-259) to stdout while the search runs,
                              * at most one per second */
#define MC_F_NOBATCH 256u /* A/B only: one host round trip per BFS level even while the frontier is small         */
#define MC_F_UNVERIFIED 512u /* mc_check_files / mc_resolve_files: when the module an MC wrapper EXTENDS (raft.tla, the snapshot-isolation
                             specs) is found neither beside it nor under $TLA_PATH, use the built-in lowering anyway (the report
                             carries a warning) instead of failing with MC_ENOSPEC                                          */
#define MC_F_JIT 262144u /* MC_SPEC_PCAL engines: translate the compiled program into straight-line C++ (mc_program_codegen) and build it for
                            the device when the engine is created (hipcc, cached by the hash of the text: seconds to a minute the first
                            time) instead of interpreting its bytecode on the device; $TLAMC_JIT=1 does the same for every such engine.
                            Same states and counts; the engine STORES its rows packed to the cells' inferred ranges (an interval analysis
                            of the program; $TLAMC_JIT_PACK=0: 32 bits per cell like the interpreter) and hands them out — mc_engine_trace,
                            mc_engine_read_states — as the interpreter's rows, mc_state_bytes(spec) each; a checkpoint of packed rows is
                            continued by an engine of the same generated code only (mc_check_files' recover path tries that by itself);
                            the mc_shard_* entry points refuse a packed engine (a config with shard_count > 1 is built unpacked).
                            A program the translator does not cover (sets of records) or a box without hipcc falls back to the
                            interpreter and says so on stderr */
#define MC_F_GENERIC 128u /* mc_check_files: run a PlusCal module through the compiled program (MC_SPEC_PCAL) even
                             when a hand lowering of its algorithm exists (A/B of the two paths)  */

typedef struct {
    int32_t device;          /* HIP device ordinal                                              */
    uint32_t flags;          /* MC_F_*                                                          */
    uint64_t table_capacity; /* seen-set slots, ANY number (rounded up to whole 64-slot groups; 0 = default 2^24):
                                size it to the HBM that is left, not to a power of two             */
    uint64_t arena_capacity; /* states kept resident in HBM (0 = default)                       */
    uint64_t chunk_states;   /* frontier states expanded per launch (0 = default)               */
    uint64_t max_levels;     /* 0 = unlimited                                                   */
    uint64_t max_distinct;   /* 0 = unlimited; stop after the level whose cumulative D >= it    */
    uint32_t shard_rank;     /* fingerprint-sharded mode: this engine's rank ...                */
    uint32_t shard_count;    /* ... of shard_count (0 or 1 = single GPU)                        */
} mc_config;

#define MC_MAX_LEVELS 4096

typedef struct {
    uint64_t distinct;        /* "distinct states found"   (README.md:319)                      */
    uint64_t generated;       /* "states generated"        (README.md:319)                      */
    uint64_t queue_left;      /* "states left on queue"    (README.md:319)                      */
    uint32_t depth;           /* "The depth of the complete state graph search is D." (README.md:320) */
    int32_t verdict;          /* MC_V_*                                                         */
    int32_t violated_invariant; /* index into the spec's invariant list, -1 if none             */
    uint32_t trace_len;       /* states in the counterexample, 0 if none                        */
    uint32_t levels;          /* entries valid in level_distinct[]                              */
    uint32_t host_evaluated;  /* 1: mc_check_files ran a module WITHOUT a GPU lowering on the host's general TLA+ evaluator */
    uint32_t unchecked_properties; /* cfg PROPERTIES with a liveness part (<>, ~>, WF_ / SF_) that were NOT checked: only their safety
                                 parts ([]P, [][A]_v, initial predicates) are; the report names them in a "Warning:" line.  A caller
                                 that needs TLC's verdict on them must not read "No error has been found" as that verdict. */
    double seconds;           /* init -> last level complete, device work included              */
    uint64_t level_distinct[MC_MAX_LEVELS]; /* new distinct states per BFS level                */
} mc_result;

/* per-kernel timing, measured with HIP events on the engine's own stream */
typedef struct {
    uint64_t launches;
    double ms_total;
    uint64_t units;           /* states (expand/materialise) or candidates (insert) processed   */
} mc_kernel_stat;
typedef struct {
    mc_kernel_stat expand, insert, materialise;
    uint64_t state_bytes;     /* W: packed bytes per state in HBM                               */
    uint64_t cand_cells;      /* candidate-matrix cells probed                                  */
    uint64_t inwave_states;   /* states the expand wavefronts wrote themselves (the others went through k_materialise) */
} mc_kernel_stats;

typedef struct mc_engine mc_engine;

/* ------------------------------------------------------------------ whole-run API
 * Together these replace the BFS inside `tlc X.tla` (reference Makefile:6-7): initial-state
 * enumeration, successor generation, fingerprinting, seen-set, invariant / Assert / deadlock
 * checks, counters, depth and counterexample (README.md:267-321, testout2:1-266). */
int mc_engine_create(const mc_spec_desc *spec, const mc_config *cfg, mc_engine **out);
int mc_engine_run(mc_engine *e, mc_result *out);
/* Incremental search: `levels` more BFS levels.  The first call starts from Init (like mc_engine_run with max_levels = levels);
 * a call after a step that stopped at its budget (MC_V_BUDGET) continues where it stopped — arena, seen-set and parent pointers
 * stay resident in HBM, nothing is recomputed; `out` holds the counters of the whole search so far.  After a step that ended
 * the search (MC_V_OK) or found an error, the next call starts over.  cfg.max_distinct still applies. */
int mc_engine_step(mc_engine *e, uint32_t levels, mc_result *out);
/* Progress reports while mc_engine_run searches (TLC's "Progress(5): 6117 states generated, 195 distinct states found, 1 states
 * left on queue.", testout2:4-259): `fn` is called from the calling thread between two BFS levels, at most once per
 * `min_interval_seconds`, with the number of levels found so far and the three counters.  fn = NULL switches it off. */
typedef void (*mc_progress_fn)(void *user, uint32_t levels, uint64_t generated, uint64_t distinct, uint64_t queue);
int mc_engine_set_progress(mc_engine *e, mc_progress_fn fn, void *user, double min_interval_seconds);
/* From inside a progress callback: the running mc_engine_run / mc_engine_step stops before its next BFS level as if its budget
 * were spent (MC_V_BUDGET; everything searched so far stays resident, mc_engine_step continues it).  TLC has no counterpart (one
 * interrupts it with ^C and -recover's from the last checkpoint); `mc` uses it to move a compiled PlusCal program from the device
 * interpreter to generated code once that is built. */
int mc_engine_request_stop(mc_engine *e);
/* counterexample of the last run: states_out receives trace_len records of mc_state_bytes()
 * bytes each (plain word order), actions_out the action id that produced each state (-1 for
 * the initial state).  *n_inout: capacity in, count out. */
int mc_engine_trace(mc_engine *e, uint8_t *states_out, int32_t *actions_out, size_t *n_inout);
int mc_engine_kernel_stats(mc_engine *e, mc_kernel_stats *out);
/* copy `count` resident states starting at arena index `first` (discovery order: level by level)
 * to the host, mc_state_bytes() bytes each — TLC's "states/" dump, for tests and tooling */
int mc_engine_read_states(mc_engine *e, uint64_t first, uint64_t count, uint8_t *out);
/* Checkpoint / recover — TLC checkpoints a run into its states/ directory ("-- Checkpointing of run states/01-08-03-18-14-01
 * completed.", reference examples/SpecifyingSystems/AdvancedExamples/testout1:10; .gitignore:2) and continues it with -recover.
 * mc_engine_checkpoint: after an mc_engine_run that ended without a violation (normally MC_V_BUDGET), write every distinct
 *   state found (the arena's blocks as they lie in HBM), the level boundaries, the counters and — with MC_F_TRACE — the
 *   parent pointers to `path`.  The seen-set is not written.
 * mc_engine_restore: load such a file into an engine created for the SAME spec descriptor (MC_EBADCFG otherwise); the next
 *   mc_engine_run rebuilds the seen-set from the states' fingerprints and continues with the checkpoint's last level as
 *   the frontier: counters, depth, per-level counts and counterexamples are those of an uninterrupted run.  max_levels /
 *   max_distinct of the new engine are absolute (they count the checkpointed part).  Single-GPU engines only. */
int mc_engine_checkpoint(mc_engine *e, const char *path);
int mc_engine_restore(mc_engine *e, const char *path);
/* profiling aid: re-expand every resident state of the last run (all probes hit); extra_flags 16 = no probes */
int mc_engine_debug_reexpand(mc_engine *e, unsigned extra_flags, double *ms);
/* profiling builds only (libtlamc.so compiled with -DMC_PHASE_PROF, profiles/phase_prof.sh): shader-clock cycles per phase of the
 * by-family expand kernel summed over all wavefronts since the last reset (48 words: layout in engine.hip); MC_ESTATE otherwise */
int mc_engine_debug_phases(mc_engine *e, uint64_t *out48, int reset);
/* profiling only: set / clear bits of the engine's flags between two calls (the A/B and ablation bits a kernel reads at run time) — e.g. run
 * to a level normally, then time ONE more level (mc_engine_step) with an ablation bit on.  Never changes what a run without the bits computes. */
int mc_engine_debug_flags(mc_engine *e, uint32_t set, uint32_t clear);
void mc_engine_destroy(mc_engine *e);

/* ------------------------------------------------------------------ sharded (multi-GPU) step API
 * One engine per GPU / process; the seen-set is partitioned by fingerprint high bits
 * (owner = mc_fp_owner).  The caller (tla_rust_amd/sharded.py) moves the buckets between
 * ranks with an all-to-all over RCCL/xGMI; all pointers are device pointers it allocated.
 * Per level and per round:
 *   expand     : frontier chunk -> candidate fingerprints bucketed by owner
 *   (all-to-all of fingerprints)
 *   probe      : owner inserts received fingerprints, answers one byte (1 = new) per fingerprint
 *   (all-to-all of answers, reversed)
 *   materialise: sender builds the full successor state of every "new" answer, bucketed by owner
 *   (all-to-all of states)
 *   ingest     : owner appends received states to its next-level frontier.
 * Exchange format of full states: each owner's bucket is a whole number of 64-state BLOCKS
 * (mc_state_bytes() * 64 bytes each), word-major inside a block like the HBM arena, so both ends
 * move them with coalesced accesses; send_counts[] are STATES, a bucket occupies
 * ceil(count / 64) blocks.  mc_shard_ingest takes ONE source's bucket per call. */
/* Optional: hand the engine the HIP stream (hipStream_t) the caller issues its collectives on.  The engine then
 * enqueues bucket compaction, probe, keep, materialise and ingest on THAT stream and returns without waiting
 * (stream order ties them to the caller's all-to-alls); only calls that return counts to the host
 * (expand_finish, materialise, end_level, counters) block.  enable = 0 restores the blocking default.
 * With it mc_shard_keep* reports *n_new = 0: the count stays on the device until mc_shard_end_level. */
int mc_shard_set_stream(mc_engine *e, void *hip_stream, int enable);
int mc_shard_begin(mc_engine *e);                                  /* Init: keep the initial states this rank owns */
/* Alternative start: every rank runs the same single-GPU BFS until a level has at least min_frontier states (the
 * small first levels are not worth a collective each), keeps the states of that level whose fingerprint it owns as its
 * local frontier and holds all prefix fingerprints in its seen-set.  levels_out[0 .. *nlevels) = states per level of the
 * prefix, the last one being the level the sharded rounds continue with (*nlevels in: capacity, out: count).  Only
 * rank 0 reports the prefix's `generated`; distinct_local excludes what another rank already counts.  max_distinct /
 * max_levels (0 = none) are the budgets of the WHOLE job: the prefix stops where a single-GPU run would. */
int mc_shard_begin_replicated(mc_engine *e, uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels,
                              uint64_t *levels_out, uint32_t *nlevels);
int mc_shard_level_size(mc_engine *e, uint64_t *frontier_states);  /* local frontier of the current level          */
int mc_shard_expand(mc_engine *e, uint64_t first, uint64_t count,  /* chunk of the local frontier                  */
                    uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts /* [shard_count] host */);
/* The same step split in two so that rounds can be software-pipelined: _launch enqueues the expand of a chunk
 * into `slot` (0 or 1) on the engine's main stream and returns at once; _finish waits for it and compacts the
 * slot's buckets into send_fp.  Between them the caller may run probe / keep / materialise / ingest for the
 * OTHER slot (they execute on the engine's side stream) and its collectives.  mc_shard_expand = both, slot 0. */
int mc_shard_expand_launch(mc_engine *e, uint32_t slot, uint64_t first, uint64_t count, uint64_t send_cap);
int mc_shard_expand_finish(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts);
int mc_shard_probe(mc_engine *e, const uint64_t *recv_fp, uint64_t n, uint8_t *answers);
int mc_shard_materialise(mc_engine *e, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap,
                         uint64_t *send_counts /* [shard_count] host, in states */);
int mc_shard_materialise_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint8_t *send_states,
                              uint64_t send_cap, uint64_t *send_counts);
int mc_shard_ingest(mc_engine *e, const uint8_t *recv_states, uint64_t n);
/* FIXED-CAPACITY rounds of the "stay" form — nothing in a round waits for the host: every rank sends every rank a bucket of
 * exactly `cap` words, word 0 = the number of fingerprints that follow (in band: no size message, no device-to-host copy
 * before the payload all-to-all can be sized), so the two collectives of a round are equal-split all-to-alls of cap words and
 * cap bytes per peer.
 *   _expand_pack(slot, send_fp[shard_count * cap], cap): after mc_shard_expand_launch(slot, ...) — orders itself behind the
 *       expand kernel with an event and packs owner t's candidates into send_fp[t * cap + 1 ...], count in send_fp[t * cap];
 *   _probe_pack(recv_fp[shard_count * cap], cap, answers[shard_count * cap]): bucket s came from rank s; answers keep the
 *       positions and are 0 outside a bucket's count;
 *   _keep_pack(slot, answers_back[shard_count * cap], cap): the sender materialises its positively answered candidates.
 * cap must be the same on every rank (derive it from the level's frontier sizes, which every rank knows).  A bucket that
 * does not fit is not truncated: the level fails with MC_EROUTE at mc_shard_end_level ("an exchange bucket is full"). */
int mc_shard_expand_pack(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t cap);
int mc_shard_probe_pack(mc_engine *e, const uint64_t *recv_fp, uint64_t cap, uint8_t *answers);
int mc_shard_keep_pack(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t cap);
/* device-side wait of the caller's stream for the slot's last keep: call it before a collective overwrites the answers
 * buffer that keep was given (a caller with one buffer per slot; callers that allocate per round need not) */
int mc_shard_wait_keep(mc_engine *e, uint32_t slot);
/* "stay" alternative to materialise + all-to-all + ingest: the positively answered candidates of the last
 * mc_shard_expand are materialised into THIS rank's frontier (only fingerprints crossed xGMI).  The caller
 * uses it once the frontier is large enough to stay balanced, and falls back to the moving form to rebalance. */
int mc_shard_keep(mc_engine *e, const uint8_t *answers_back, uint64_t *n_new);              /* slot 0 */
int mc_shard_keep_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t *n_new);
int mc_shard_end_level(mc_engine *e, uint64_t *new_local_states);  /* swap frontiers                                */
int mc_shard_counters(mc_engine *e, uint64_t *generated, uint64_t *distinct_local, int32_t *verdict);
/* Before a sharded run that stops on a budget reports: evaluate the invariants of this rank's unexpanded frontier for the
 * specs that check invariants when a state is expanded (the snapshot-isolation models); TLC checks a state when it is
 * generated, so no counted state may stay unchecked.  A violation shows up in mc_shard_counters' verdict.  No-op for the
 * other specs. */
int mc_shard_check_frontier(mc_engine *e);
/* Counterexamples of a sharded run (engines created with MC_F_TRACE): every state records the rank and arena index of its
 * parent.  A state that stays on the rank that generated it (stay mode, local-owner shortcut, replicated prefix) has a local
 * parent; a state that MOVES to its owner takes its parent with it: after mc_shard_materialise_slot the sender fills
 * `send_parents` (one word per moved state, same owner order, (index << 16) | slot), the caller exchanges it like the states,
 * and the owner calls mc_shard_ingest_parents right after the mc_shard_ingest of the same bucket.
 * mc_shard_violation: this rank's first violation (arena index, slot code as in the single-GPU engine: 0xffff deadlock,
 * 0xfffd the state itself violates an invariant, otherwise the slot whose successor does).
 * mc_shard_fetch: one step of the walk — the packed state at `idx` of this rank, the rank / index of its parent
 * (0xffffffff: an initial state) and the slot that produced it (0xfffc: this entry is a COPY of state parent_idx made by the
 * replicated prefix, not a step).  With bit 63 of `idx` set the low bits are an ORDINAL of Init's enumeration: the state is
 * rebuilt from it (an invariant violated by an initial state reports that ordinal, slot 0xfffe, not an arena index). */
int mc_shard_materialise_parents(mc_engine *e, uint32_t slot, uint64_t *send_parents);
int mc_shard_ingest_parents(mc_engine *e, const uint64_t *recv_parents, uint64_t n, uint32_t src_rank);
int mc_shard_violation(mc_engine *e, int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant);
int mc_shard_fetch(mc_engine *e, uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot);
/* One checkpoint file per rank (TLC checkpoints a run and continues it with -recover: testout1:10 "-- Checkpointing of run
 * states/... completed."; mc_engine_checkpoint is the single-GPU form).  After an mc_shard_run* that ended without an error —
 * normally MC_V_BUDGET — every rank writes ITS share with mc_shard_checkpoint (a different path per rank): arena, parent
 * pointers (index, slot, rank), its seen-set slice as it lies in HBM (the states a rank holds are the ones it generated or
 * was sent, the fingerprints it stores are the ones it OWNS: the slice cannot be rebuilt from the arena), its counters and the
 * job's level table.  mc_shard_restore loads such a file into an engine created for the same spec, rank, world size,
 * table_capacity and MC_F_TRACE (MC_EBADCFG otherwise; arena_capacity may differ); the next mc_shard_run* of those engines
 * continues with the unexpanded frontier instead of starting at Init — every rank must have restored a file of the SAME run
 * (checked: MC_EBADCFG on all ranks otherwise).
 * mc_shard_note_levels / mc_shard_resume are the level loop's side of it (shard_loop.h): the loop leaves the job's level table
 * with the engine when a run ends, and asks a restored engine for it when the next one begins (*nlevels = 0: nothing restored). */
int mc_shard_checkpoint(mc_engine *e, const char *path);
int mc_shard_restore(mc_engine *e, const char *path);
int mc_shard_note_levels(mc_engine *e, const uint64_t *levels, uint32_t n, int32_t verdict);
int mc_shard_resume(mc_engine *e, uint64_t *levels_out, uint32_t *nlevels);

/* ------------------------------------------------------------------ hip-rccl back-end (tla_rust_amd/csrc/shard_rccl.cpp)
 * The level loop of the sharded search behind the C ABI, collectives over RCCL (xGMI): one process per GPU, no Python.
 *   rank 0:      mc_comm_unique_id(id)            -- ncclGetUniqueId; ship the MC_COMM_ID_BYTES to the other ranks (file, socket, env ...)
 *   every rank:  mc_comm_create(id, rank, world, device, &c)          -- ncclCommInitRank + the stream the collectives run on
 *                mc_engine_create(spec, cfg with shard_rank / shard_count, &e)
 *                mc_shard_run(e, c, &opts, &result)                    -- the whole search; every rank gets the global counters
 * replaces: TLC's worker pool ("Number of worker threads", examples/serializableSnapshotIsolation.tla:52-53) scaled past one
 * device.  Per level one collective for frontier sizes + verdicts + statuses (ncclAllGather); the rounds inside a level are
 * ncclSend / ncclRecv groups — of exact sizes after a small all-gather of the bucket counts (the default), or the fixed-capacity
 * exchanges above (mc_shard_*_pack, no host wait inside a level): MC_SHARD_* below.  Any RCCL failure returns MC_ERCCL
 * (mc_last_error() carries ncclGetErrorString).  A rank-local failure (a full exchange bucket, MC_ETABLEFULL, MC_EOVERFLOW,
 * an allocation) does not leave the others waiting in a collective: every rank's status travels with the per-level
 * all-gather and all ranks leave the loop together with the first failing rank's code.  Levels whose new states STAY where they
 * were generated need 9 bytes per routed candidate (the expand of round r+1 overlaps the exchange, the probes and the
 * materialisation of round r; the fixed-capacity forms also overlap the exchange of round r+1 with the probes of round r and move
 * their buckets whole); small or unbalanced levels MOVE the new states to
 * their owners (mc_shard_materialise_slot / mc_shard_ingest).  `mc X.tla -gpus P` is this API with forked ranks. */
#define MC_COMM_ID_BYTES 128
typedef struct mc_comm mc_comm;
#define MC_SHARD_NO_PREFIX 1u /* mc_shard_opts.flags: shard from Init on (mc_shard_begin) instead of the replicated prefix */
/* How a STAY level exchanges its candidates (mc_shard_opts.flags; `mc -gpus P -exchange exact | measured | packed`):
 *   default (neither flag): HOST-PACED rounds with exact sizes — the ranks exchange their P bucket counts (one small all-gather),
 *     then all_to_all_v moves exactly 8 bytes per routed candidate out and 1 byte back: the least the exchange can move, no
 *     capacity to guess and no bucket that could overflow, for one host wait + two small all-gathers per round (the counts, and "every
 *     rank has the buffers its announced sizes need": a rank that could not allocate must not leave its peers inside the all-to-all;
 *     one agreement covers both all-to-alls of the round) (the next round's expand is
 *     launched before them and overlaps the exchange, the probes and the materialisation of this one);
 *   MC_SHARD_PACKED: pipelined FIXED-CAPACITY rounds, counts in band, no host wait inside a level (mc_shard_*_pack); the buckets are
 *     moved whole, so their capacity is the exchange volume: it is sized from the previous level's MEASURED fill
 *     (the loop reads the in-band counts of every round back, asynchronously; cap_safety_pct) ...
 *   MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS: ... or from packed_fanout alone (rounds 2-3).  A search restarted after MC_EROUTE in a
 *     measured level runs so. */
#define MC_SHARD_FIXED_CAPS 2u
#define MC_SHARD_PACKED 4u
/* what one rank's level loop did (diagnostics; filled when mc_shard_opts.stats != NULL).  With TLAMC_SHARD_DEBUG set in the
 * environment rank 0 also prints one line per level to stderr: its kind, states, rounds, the fullest bucket per expanded state. */
typedef struct {
    uint64_t replicated_levels; /* levels every rank ran itself (mc_shard_begin_replicated)                                  */
    uint64_t stay_levels;       /* levels whose new states stayed on the generating rank (9 B per candidate cross xGMI)      */
    uint64_t move_levels;       /* levels whose new states moved to their owner (small frontiers, rebalancing)               */
    uint64_t rounds;            /* exchange rounds (one chunk per rank each)                                                 */
    uint64_t sent_bytes;        /* bytes THIS rank handed to the all-to-alls for other ranks (fingerprints, answers, states) */
    uint64_t distinct_local;    /* states resident on this rank at the end (its share)                                       */
    uint64_t max_frontier;      /* over the sharded levels: largest max-over-ranks frontier ...                              */
    uint64_t mean_frontier;     /* ... and the mean frontier of that same level (imbalance = max / mean)                     */
    uint64_t restarts;          /* times the search was started over from Init with twice the fan-out allowance (MC_EROUTE)  */
    uint64_t routed_candidates; /* candidates THIS rank sent to other ranks' seen-set slices (stay rounds: the in-band counts; move
                                 * rounds: the exact sizes): 9 bytes each (fingerprint out, answer back) is what the exchange needs */
    uint64_t fp_answer_bytes;   /* the part of sent_bytes that carried fingerprints and answers (the rest: moved states, parents)  */
    uint64_t measured_levels;   /* stay levels whose buckets were sized from the previous level's measured fill                   */
    uint64_t engine_ns;         /* host time this rank spent inside engine calls (mc_shard_*: launches AND the waits for its own streams) */
    uint64_t collective_ns;     /* host time inside the transport's collectives (the exchange itself + waiting for the slowest peer)  */
    uint64_t collectives;       /* how many collectives that was                                                                  */
} mc_shard_stats;
typedef struct {
    uint64_t chunk_states;    /* frontier states per round and rank (0 = 2^19); clamped to the engine's chunk_states         */
    uint64_t max_distinct;    /* budgets of the whole job (0 = none)                                                        */
    uint64_t max_levels;
    uint64_t replicate_until; /* states per rank a level must have before the search is sharded (0 = 2^15); below it every
                               * rank runs the same fused BFS (mc_shard_begin_replicated)                                    */
    uint64_t packed_fanout;   /* in-model successors per expanded state the fixed-capacity buckets allow for (0 = 16); a
                               * round that exceeds it is never truncated: the level fails on every rank (MC_EROUTE) and
                               * mc_shard_run* starts the search over with twice the allowance (at most 5 times)            */
    uint64_t stay_threshold;  /* states per rank a level needs before its new states STAY where they were generated
                               * (0 = 2^15); smaller levels MOVE every new state to its owner, which is what spreads a
                               * small frontier over the ranks                                                              */
    double rebalance_ratio;   /* a level whose largest per-rank frontier exceeds ratio x the mean is a MOVE level again:
                               * a drifting rank cannot stall the level loop (0 = 1.25)                                     */
    uint64_t move_fanout;     /* in-model successors per expanded state the buffers of a move round allow for (0 = 32)      */
    uint32_t flags;           /* MC_SHARD_*                                                                                 */
    uint32_t cap_safety_pct;  /* a stay level's buckets hold this many percent of what the previous level's fullest bucket held
                               * per expanded state (0 = 140; never more than packed_fanout allows; the first stay level and a level
                               * after one without packed rounds use packed_fanout).  Only the capacity — i.e. the bytes moved over
                               * xGMI — depends on it, never a count: a bucket that does not fit still fails the level (MC_EROUTE)   */
    mc_shard_stats *stats;    /* NULL or where to put this rank's loop statistics                                            */
} mc_shard_opts;
/* TEST-ONLY hooks of this part of the library (a one-GPU lab has no second device): the environment variable TLAMC_RCCL names the
 * file dlopen()ed instead of /opt/rocm/lib/librccl.so (tests/_fakerccl: the nine nccl* entry points over POSIX shared memory, so
 * that P ranks can share ONE device — RCCL itself refuses two ranks on a device); `mc X.tla -gpus P -samedevice` and
 * `bench.py --gpus N --share-gpu` put every rank on device 0 for it.  None of them belongs in a deployment: without TLAMC_RCCL
 * the real RCCL is loaded, and a run through the stand-in says so in its report / bench line ("NOT_A_MEASUREMENT").
 * TLAMC_TEST_FAIL_AT = "rank:level:code[,rank:level:code...]" makes that rank of mc_shard_run* fail that BFS level with that status,
 * once per process (tests/test_sharded_gloo.py: two DIFFERENT failures in one level must end every rank with ONE agreed code —
 * MC_EROUTE, i.e. a restart, only if every failing rank reports MC_EROUTE). */
int mc_comm_unique_id(uint8_t id_out[MC_COMM_ID_BYTES]);
int mc_comm_create(const uint8_t id[MC_COMM_ID_BYTES], uint32_t rank, uint32_t world, int32_t device, mc_comm **out);
void mc_comm_destroy(mc_comm *c);
/* host-side all-gather of `bytes` bytes per rank over the communicator (rank r's block at all_out + r * bytes): a barrier,
 * a max-over-ranks of wall times, the shares of a run — what a multi-process host needs besides the search itself */
int mc_comm_all_gather(mc_comm *c, const void *mine, void *all_out, uint64_t bytes);
int mc_shard_run(mc_engine *e, mc_comm *c, const mc_shard_opts *opts, mc_result *out);
/* Counterexample of a sharded run that ended in a violation (engines created with MC_F_TRACE): the behaviour is walked back
 * parent by parent ACROSS the ranks' arenas (mc_shard_violation / mc_shard_fetch on the rank that holds the state, one small
 * all-gather per step).  Collective: every rank calls it and receives the same trace.  states_out: *n_inout records of
 * mc_state_bytes() bytes (capacity in, count out), first the initial state; slots_out[k] = the action slot that led from state
 * k-1 to state k (-1 for k = 0): mc_state_action(spec, state k-1, slot) names it.  *final_slot >= 0: the violation is an
 * INVARIANT broken by a successor that is stored nowhere — the caller appends mc_state_apply(spec, last state, *final_slot);
 * -1 otherwise (Assert / evaluation error / deadlock / the last state itself violates).  *n_inout = 0: no rank found one. */
int mc_shard_trace(mc_engine *e, mc_comm *c, uint8_t *states_out, int32_t *slots_out, size_t *n_inout, int32_t *final_slot);

/* ---- bring your own collectives.  mc_shard_run / mc_shard_trace are written against this table; mc_comm fills it with RCCL
 * (ncclSend / ncclRecv groups, ncclAllGather).  A host that already owns a communicator — torch.distributed's process group
 * (tla_rust_amd/sharded.py: backend "nccl" = RCCL, or gloo staged through the host in the CPU tests), MPI — hands its own
 * functions over and gets the SAME level loop.  All buffers the loop passes to all_to_all / all_to_all_v come from alloc()
 * (device memory for a HIP engine) and are passed by their base address; every collective is ordered on `hip_stream` like a
 * kernel launch (the loop makes that stream wait for its producers with events and lets its consumers wait for it), may
 * return before it has completed, and must be called by every rank in the same order.  Return 0 or a negative MC_E* code. */
typedef struct mc_transport {
    void *user;
    uint32_t rank, world;
    void *hip_stream;  /* hipStream_t the collectives are ordered on; NULL: the collectives are synchronous host calls */
    void *(*alloc)(void *user, size_t bytes);
    void (*release)(void *user, void *p);
    /* equal split: bytes_per_peer bytes at send + p * bytes_per_peer go to rank p, rank p's block lands at recv + p * bytes_per_peer */
    int (*all_to_all)(void *user, const void *send, void *recv, uint64_t bytes_per_peer);
    /* variable split: send_bytes[p] bytes at send + send_off[p] go to rank p; recv_bytes[p] bytes from rank p land at recv + recv_off[p] */
    int (*all_to_all_v)(void *user, const void *send, const uint64_t *send_off, const uint64_t *send_bytes, void *recv,
                        const uint64_t *recv_off, const uint64_t *recv_bytes);
    /* HOST buffers, blocking: rank r's `bytes` bytes appear at all_out + r * bytes on every rank */
    int (*all_gather)(void *user, const void *mine, void *all_out, uint64_t bytes);
    /* OPTIONAL (may be NULL: a transport zero-initialises the struct and fills what it has): all_to_all without the rank's own
     * block — recv + rank * bytes_per_peer is left untouched.  The fingerprint exchange of a stay round uses it when present: a
     * rank's own bucket is empty by construction (candidates it owns are probed where they are generated), so moving it is a
     * device copy of the whole bucket for its count word; the loop zeroes that word itself. */
    int (*all_to_all_others)(void *user, const void *send, void *recv, uint64_t bytes_per_peer);
} mc_transport;
int mc_comm_transport(mc_comm *c, mc_transport *out);  /* the RCCL functions of a communicator; valid while c lives */
int mc_shard_run_transport(mc_engine *e, const mc_transport *t, const mc_shard_opts *opts, mc_result *out);
int mc_shard_trace_transport(mc_engine *e, const mc_transport *t, uint8_t *states_out, int32_t *slots_out, size_t *n_inout,
                             int32_t *final_slot);
/* what the level loop needs to know about an engine: its expand stream (the loop orders bucket compaction / its collectives
 * against it with events), the largest chunk one launch takes, whether parent pointers are kept (MC_F_TRACE) */
int mc_shard_info(mc_engine *e, void **main_stream_out, uint64_t *chunk_states_out, int32_t *traced_out);

/* ------------------------------------------------------------------ PlusCal front-end (host only)
 * The reference's workflow is `pcal2tla *tla` then `tlc *tla` (Makefile:3-7).  mc_pcal_translate is the first
 * half: it returns the module text with the TLA+ translation of its `--algorithm` inserted (p-manual.pdf App. B).
 * mc_program_compile is what lets the checker run a PlusCal spec nobody hand-lowered: the algorithm (p- or c-syntax;
 * labels, := , if/elsif/else, while, either/or, with, await/when, assert, skip, goto, define, macros, LET; integers, booleans, strings,
 * functions over constant sets, bounded sequences and arrays of them, sets of small naturals; procedures incl. recursion, records incl. nested
 * ones, sequences and sets of records: DESIGN.md section 9) becomes a bytecode program every GPU lane interprets on its own packed state
 * (tla_rust_amd/csrc/spec_vm.h).  Two bounds come from the environment at compile time: $TLAMC_PCAL_SEQ, the element cells per sequence /
 * set of records (1 .. 16, default 8; a longer one is MC_EOVERFLOW, never truncated), and $TLAMC_PCAL_STACK, the frames per recursive
 * procedure (default 4; a deeper call fails an assertion at the call).  cfg_text: CONSTANT(S) with integer / string / model-value / set values and
 * INVARIANT(S) naming zero-argument definitions of the module; NULL = no constants, no invariants. */
typedef struct mc_program mc_program;
int mc_pcal_translate(const char *tla_text, char *out, size_t cap);   /* >= 0: bytes needed (NUL excluded); < 0: MC_E* */
int mc_program_compile(const char *tla_text, const char *cfg_text, mc_program **out);
int mc_program_spec(const mc_program *p, mc_spec_desc *out);          /* valid while p lives */
const char *mc_program_translated(const mc_program *p);               /* the module text after translation */
/* the program as generated C++ (a header for tla_rust_amd/csrc/spec_gen.h: what MC_F_JIT compiles): the text's length, at most cap - 1
 * characters of it in buf; MC_EBADCFG (+ mc_last_error) when the program uses an instruction the translator does not cover */
long mc_program_codegen(const mc_program *p, char *buf, size_t cap);
const char *mc_program_invariant(const mc_program *p, int index);     /* name of INVARIANT number index */
int mc_program_assert_pos(const mc_program *p, int index, int *line, int *col);  /* source position of an assert */
void mc_program_free(mc_program *p);

/* ------------------------------------------------------------------ helpers (host only) */
size_t mc_state_bytes(const mc_spec_desc *spec);                    /* W, 0 if the spec is invalid   */
uint32_t mc_fp_owner(uint64_t fp, uint32_t shard_count);
/* canonical TLA+ text of one packed state ("/\ var = value" lines, README.md:272-276) */
int mc_state_format(const mc_spec_desc *spec, const uint8_t *state, char *buf, size_t cap);
const char *mc_action_name(const mc_spec_desc *spec, int32_t action);
/* host evaluation of one (packed state, slot) pair: the action id of the slot (for mc_action_name; negative MC_E* on error)
 * and the successor it produces — what a host needs to print a counterexample it assembled itself (sharded runs) */
int mc_state_action(const mc_spec_desc *spec, const uint8_t *state, int32_t slot);
int mc_state_apply(const mc_spec_desc *spec, const uint8_t *state, int32_t slot, uint8_t *successor_out);
const char *mc_strerror(int code);
const char *mc_last_error(void);                                    /* detail of the last failure    */
int mc_device_count(void);

/* ------------------------------------------------------------------ front-end
 * Replaces TLC's reading of X.cfg (grammar: examples/SpecifyingSystems/TLC/ConfigFileGrammar.tla:4-32)
 * and its selection of Init/Next/invariants; the spec itself is selected by MODULE name
 * (pcal_intro.tla:1) from the registry of hand-lowered specs. */
typedef struct mc_cfg mc_cfg;
int mc_cfg_parse(const char *text, size_t len, mc_cfg **out);
void mc_cfg_free(mc_cfg *c);
/* JSON rendering of the parsed cfg, for bindings and tests */
int mc_cfg_json(const mc_cfg *c, char *buf, size_t cap);
/* resolve module + cfg to a lowering; fails with MC_ENOSPEC for modules that are not lowered */
int mc_spec_resolve(const char *module_name, const mc_cfg *c, mc_spec_desc *out);
/* the front half of `tlc X.tla` (reference Makefile:6-7, README.md:262,356): read X.tla and the X.cfg beside it (or cfg_path),
 * pick the lowering (verified against the module text) or compile the PlusCal algorithm, and return the descriptor
 * mc_engine_create takes — what a multi-GPU host needs to create one engine per rank (mc_shard_*).  flags: MC_F_GENERIC.
 * *prog_out = the compiled program when the module went through the PlusCal compiler (the descriptor points into it: free
 * it with mc_program_free after the engines are destroyed), NULL otherwise. */
int mc_resolve_files(const char *tla_path, const char *cfg_path, unsigned flags, mc_spec_desc *out, mc_program **prog_out);
/* `tlc X.tla` end to end: read X.tla / X.cfg, run on `cfg->device`, write TLC's report text.
 * A TLA+ module that belongs to none of the lowered families (no PlusCal algorithm, not raft / the snapshot-isolation specs /
 * Voting / Paxos or a wrapper of them) is evaluated on the HOST by the general evaluator (csrc/tlaeval.cpp: the Specifying Systems
 * examples of the reference; `MCInnerSerial.tla` gives testout2:260-266): out->host_evaluated = 1, the report's first line says
 * so, cfg->max_levels / max_distinct / MC_F_DEADLOCK / MC_F_PROGRESS apply, $TLA_PATH (a ':'-separated list) names more module
 * directories.  A module of a lowered family is never evaluated on the host: what its lowering refuses stays MC_ENOSPEC. */
int mc_check_files(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report,
                   size_t report_cap, mc_result *out);
/* the same, and every distinct state found is written to dump_path in TLC's `-dump` layout ("State k:" + the
 * variables, blank line), in discovery order; dump_path NULL = mc_check_files */
int mc_check_files_dump(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report,
                        size_t report_cap, mc_result *out, const char *dump_path);
/* the same with TLC's -recover / checkpointing: recover_path (or NULL) is restored before the run, checkpoint_path (or
 * NULL) is written after a run that ended without an error; the report then carries TLC's line
 * "-- Checkpointing of run <path> completed." (testout1:10) */
int mc_check_files_ckpt(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report, size_t report_cap,
                        mc_result *out, const char *dump_path, const char *recover_path, const char *checkpoint_path);

#ifdef __cplusplus
}
#endif
#endif
