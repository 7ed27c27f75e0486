// engine.hip — the MI355X (gfx950) BFS engine behind include/tlamc.h.
//
// Per BFS level the frontier (a contiguous index range of the state arena in HBM) is processed in chunks of `chunk_states`
// (bench.py: 2^23) with no host synchronisation inside a level; the host reads one small counter block per level, and while
// levels are small it enqueues eight of them blind (LevelCtl).  The kernels, in the order they matter:
//
//   k_expand_family<Spec, ROUTE>   (specs with action families: raft)  one wavefront = one arena block of 64 parents, one
//                       workgroup = four.  Guards -> enabled (parent, slot) pairs, dense slots and the actions of in-flight
//                       messages evaluated by the parent's lane, the sparse fixed slots bucketed per family in LDS and evaluated
//                       64 pairs of ONE family at a time; successors that can never be stored are counted, not evaluated
//                       (S::GENERATED_ONLY); a candidate's fingerprint is the parent's plus O(delta) terms; a per-wavefront
//                       filter drops repeats; 64 candidates at a time probe the seen-set (open addressing over 32- or 64-byte
//                       buckets in HBM, agent-scope atomicCAS on write-once slots).  ROUTE = false (fused runs): the
//                       survivors wait in LDS and the WORKGROUP writes them at its end — pooled, counting-sorted by action
//                       class, one atomicAdd on the arena's fill level, row copy + patch from the parent's Summary (the
//                       in-wave tail; wave_write_survivors).  ROUTE = true (sharded runs): candidates of other owners go to
//                       exchange buckets, the rank's own are probed here and go through the new-list.
//   k_expand_insert<Spec, ROUTE>   the slot-by-slot form of the same (lane = parent, loop = action slot; every other spec).
//   k_materialise<Spec> one lane per entry of the new-list (the survivors a wavefront had no list space for, every survivor of
//                       a spec without in-wave writes, of a fast-growing level, or of a sharded run): re-evaluates its
//                       (parent, slot) and writes the full successor (coalesced: consecutive lanes own consecutive arena
//                       indices), plus (parent, slot) for counterexamples.  Runs on a second stream beside the next expand.
//   k_expand + k_insert the round-1 form — a slot-major candidate matrix between two kernels — kept behind MC_F_MATRIX for A/B.
//
// Algorithmic HBM bytes per distinct state = 2*W + 8*(G/D)  (SURVEY.md §8d): each state is written once and read once, each
// in-model successor touches one seen-set word; DESIGN.md §5 has what the kernels really move and what bounds them.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <chrono>
#include <memory>
#include <type_traits>
#include <string>
#include <vector>

#include "spec_registry.h"

namespace mc {

// The library is built from this one source compiled several times (MC_TU = 0: C ABI + host helpers;
// 1..5: the engine + kernels of one group of specs each), so the per-spec kernels compile in parallel.
#ifndef MC_TU
#define MC_TU -1  // single translation unit: everything
#endif
}  // namespace mc
extern "C" void mc_set_error_internal(const char *msg);
namespace mc {
static void set_error(const std::string &s) { mc_set_error_internal(s.c_str()); }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
            return MC_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

// Device-resident counters.  One word saturates at ~88 returning atomics/us (MI355X_MICROARCH.md
// "dequeue"), and atomics to the same cache line serialise, so everything the hot kernels bump
// once per wavefront is sharded 8 ways (shard = blockIdx.x & 7, i.e. roughly per XCD) with each
// shard on its own 128-byte line.
constexpr int NSHARD = 8;
struct alignas(128) PaddedCounter {
    unsigned long long v;
    unsigned long long pad[15];
};
struct DevCounters {
    PaddedCounter n_new[2 * NSHARD];  // survivors of the chunk in flight, per new-list segment; two parities so that
                                      // materialise(chunk c) overlaps expand(chunk c+1) on a second stream
    PaddedCounter generated[NSHARD];  // successors generated
    PaddedCounter cells[NSHARD];      // seen-set probes issued
    // next free arena index, on a line of its own.  Round 4: in a fused run (atomic_alloc != 0) it is bumped by the writers
    // themselves — the expand wavefront that appends its own survivors (one atomicAdd per wavefront, at its tail) and the
    // wavefronts of k_materialise (the overflow path) — so a new state's final index is known the moment it is written.
    // The sharded step calls keep the serialised form (atomic_alloc == 0: an appender reads arena_next, k_commit /
    // k_bump_arena_next add its count behind it; the appenders are chained by events, see append_begin).
    alignas(128) unsigned long long arena_next;
    unsigned long long pad_an[15];
    unsigned long long viol_key;      // min over (idx << 24 | slot << 8 | kind); ~0 = none
    unsigned long long via_list;      // fused runs: states that went through the new-list + k_materialise (the rest were written in-wave)
    unsigned int max_slots;           // rows of the candidate matrix written by the current chunk
    unsigned int error;               // DEV_E* bits
    unsigned int atomic_alloc;        // see arena_next
};
enum : unsigned { DEV_ETABLE = 1u, DEV_EARENA = 2u, DEV_EOVERFLOW = 4u, DEV_EROUTE = 8u /* an exchange bucket of a sharded round is full */ };
enum : unsigned { VK_INVARIANT = 1, VK_ASSERT = 2, VK_DEADLOCK = 3, VK_SPECERR = 4 };
static constexpr unsigned SLOT_NONE = 0xffffu;      // deadlock: no slot
static constexpr unsigned SLOT_INIT = 0xfffeu;      // an initial state violates an invariant
static constexpr unsigned SLOT_PARENT = 0xfffdu;    // the expanded state itself violates an invariant
static constexpr unsigned SLOT_COPY = 0xfffcu;      // sharded runs: this entry is a copy of state parent[i] (replicated prefix -> owned slice)

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) { unsigned t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) { unsigned long long t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

MC_HD CWordRef arena_cref(const uint64_t *arena, uint64_t idx, int words) {
    return CWordRef{arena + ((idx >> 6) * (uint64_t)words) * 64 + (idx & 63), 64};
}
// View of one state of the arena block a WAVEFRONT works on: the block's base address is wave-uniform (scalar registers), the
// state is a 32-bit lane offset, so every access is "global_load v, v_offset, s[base]" — no 64-bit per-lane pointer to keep
// (or spill), no 64-bit address arithmetic per access.  Block-relative word offsets fit 32 bits (a block is words * 512 bytes).
using GlobalWords = const __attribute__((address_space(1))) uint64_t *;  // (a generic pointer would make every access a flat_load)
struct BlockRef {
    GlobalWords base;  // arena + block * words * 64: uniform
    unsigned lane;     // the state inside the block
    // (loading the rows non-temporally — they are read once — was measured: 170.7 against 165.4 ms per step on the t3 graph)
    __device__ __forceinline__ uint64_t get(int w) const { return base[(unsigned)w * 64u + lane]; }
};
__device__ __forceinline__ GlobalWords uniform_ptr(const uint64_t *p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (GlobalWords)(((uint64_t)hi << 32) | lo);
}
MC_HD WordRef arena_ref(uint64_t *arena, uint64_t idx, int words) {
    return WordRef{arena + ((idx >> 6) * (uint64_t)words) * 64 + (idx & 63), 64};
}
// Order of the keys = order in which TLC would have met the errors of one level: a state that ITSELF violates an invariant
// (SLOT_PARENT: specs that check per stored state) was generated on the previous level, before anything of this level was
// expanded — bit 62 is clear for it and set for everything found while generating successors; then by arena index, then by slot.
static constexpr unsigned long long VIOL_LATER = 1ull << 62;
MC_HD unsigned long long viol_key(uint64_t idx, unsigned slot, unsigned kind, unsigned inv) {
    return ((slot & 0xffffu) == 0xfffdu ? 0ull : VIOL_LATER) | ((unsigned long long)idx << 24) | ((unsigned long long)(slot & 0xffffu) << 8) |
           ((inv & 31u) << 3) | kind;
}
MC_HD uint64_t viol_idx(unsigned long long key) { return (uint64_t)((key & ~VIOL_LATER) >> 24); }

// ------------------------------------------------------------------------------------- expand
template <class S>
__global__ void __launch_bounds__(256)
k_expand(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi,
         uint64_t *__restrict__ cand, uint64_t row_stride, uint64_t ncols, uint16_t *__restrict__ nsl,
         DevCounters *ctr, unsigned flags) {
    const uint64_t base = lo & ~63ull;
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;  // ncols is a multiple of 64: whole wavefronts leave together
    const uint64_t idx = base + col;
    const bool active = idx >= lo && idx < hi;
    const CWordRef s = arena_cref(arena, idx, S::words(prm));
    typename S::Local loc;
    int ns = 0;
    if (active) {
        S::load(prm, s, loc);
        ns = S::nslots(prm, loc);
    }
    const int wns = (int)wave_max_u32((unsigned)ns);
    unsigned gen = 0, err = 0;
    unsigned long long viol = ~0ull;
    for (int slot = 0; slot < wns; ++slot) {
        uint64_t fp = 0;
        if (slot < ns) {
            uint64_t f = 0;
            const unsigned st = S::eval(prm, loc, s, slot, f);
            if (st & ST_ENABLED) {
                ++gen;
                if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                else if (st & ST_ASSERT) viol = min(viol, viol_key(idx, (unsigned)slot, VK_ASSERT, 0));
                else if (st & ST_SPECERR) viol = min(viol, viol_key(idx, (unsigned)slot, VK_SPECERR, 0));
                else {
                    if (st & ST_INVARIANT) viol = min(viol, viol_key(idx, (unsigned)slot, VK_INVARIANT, st >> 8));
                    if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) fp = f;
                }
            }
        }
        cand[(uint64_t)slot * row_stride + col] = fp;
    }
    nsl[col] = (uint16_t)wns;
    if (active && gen == 0 && (flags & MC_F_DEADLOCK)) viol = min(viol, viol_key(idx, SLOT_NONE, VK_DEADLOCK, 0));
    const unsigned gsum = wave_sum_u32(gen);
    const unsigned long long vmin = wave_min_u64(viol);
    const unsigned eor = wave_or_u32(err);
    if ((threadIdx.x & 63) == 0) {
        if (gsum) atomicAdd(&ctr->generated[blockIdx.x & (NSHARD - 1)].v, (unsigned long long)gsum);
        if (wns) atomicMax(&ctr->max_slots, (unsigned)wns);
        if (vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
        if (eor) atomicOr(&ctr->error, eor);
    }
}

// initial states: one candidate row, column = index of the initial state inside the chunk; the
// states themselves are built once into `tmp` (plain records) and copied by k_init_materialise
template <class S>
__global__ void __launch_bounds__(256)
k_init_cand(typename S::Params prm, uint64_t first, uint64_t count, uint64_t *__restrict__ tmp, uint64_t *__restrict__ cand,
            uint64_t ncols, uint16_t *__restrict__ nsl, DevCounters *ctr, unsigned shard_rank, unsigned shard_count) {
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    uint64_t fp = 0;
    unsigned gen = 0;
    unsigned long long viol = ~0ull;
    if (col < count) {
        const int W = S::words(prm);
        S::init(prm, first + col, WordRef{tmp + col * (uint64_t)W, 1});
        const CWordRef st_ref{tmp + col * (uint64_t)W, 1};
        const unsigned st = S::init_status(prm, st_ref);
        gen = 1;
        if (st & ST_INVARIANT) viol = viol_key(first + col, SLOT_INIT, VK_INVARIANT, st >> 8);
        if (!(st & ST_OUT_OF_MODEL)) fp = S::fp_of(prm, st_ref);
        if (shard_count > 1) {  // every rank enumerates Init; each keeps (and counts) only what it owns
            const bool mine = fp ? fp_owner(fp, shard_count) == shard_rank : shard_rank == 0;
            if (!mine) { fp = 0; gen = 0; viol = ~0ull; }
        }
    }
    cand[col] = fp;
    nsl[col] = 1;
    const unsigned gsum = wave_sum_u32(gen);
    const unsigned long long vmin = wave_min_u64(viol);
    if ((threadIdx.x & 63) == 0) {
        if (gsum) atomicAdd(&ctr->generated[0].v, (unsigned long long)gsum);
        atomicMax(&ctr->max_slots, 1u);
        if (vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
    }
}

// ------------------------------------------------------------------------------------- seen-set
// Open addressing over BUCKETS of 8 slots (one 64-byte line), 64-bit fingerprints, EMPTY = 0.  A probe reads the whole
// bucket with four independent 16-byte loads — one memory round trip for 8 slots instead of one dependent 8-byte load
// per slot: at the load factors a complete graph needs (0.5 .. 0.8) linear probing slot by slot walks 3 to 9 slots per
// unsuccessful lookup, each a serialised trip to L2 / HBM.  Entries are write-once, so a non-zero word that was read is
// final whatever cache it came from; an EMPTY word may be stale (the per-XCD L2s are not coherent), so it is only ever
// taken by an agent-scope atomicCAS, whose return value is the truth.
// The table holds `nbuckets` buckets, ANY number of them (not a power of two: a seen-set is sized to the HBM that is left, and
// between 128 GiB and 256 GiB there is a lot of a 288 GB device): the home bucket is the multiply-shift of the fingerprint's low
// 32 bits (one v_mad_u64_u32; the owner rank of a sharded run comes from the high bits, fp_owner), the probe sequence is linear.
// SLOTS = 8: one 64-byte line per probe, for tables that fill up (the raft graphs: load 0.5 .. 0.8).  SLOTS = 4: 32 bytes per
// probe, for a SPARSE table (capacity >= 3 x the states it can ever hold): random HBM reads cost by the byte — 1.2-1.3 TB/s on
// this device whether they are 32- or 64-byte requests (atomic_add N = 28, 3.76 G probes into an 8 GB table: 94 ms with 32-byte,
// 197 ms with 64-byte probes) — and at load <= 1/3 a 4-slot bucket almost always decides in one request.  The engine picks
// the mode when it allocates the table (seen_arg()); bit 63 of the bucket count the kernels receive says which.
// MC_NT_PROBE (A/B): a probe reads its bucket past the L2 (`nt`: a random line of a 20 GB table is never read twice while cached,
// but it evicts a line of the parent rows the in-wave writer comes back for).  Bit 0: the synchronous prober, bit 1: the LDS-DMA.
#ifndef MC_NT_PROBE
#define MC_NT_PROBE 0
#endif
typedef unsigned long long mc_ull2 __attribute__((ext_vector_type(2)));
template <int SLOTS>
__device__ __forceinline__ bool seen_insert_t(uint64_t *table, uint64_t nbuckets, uint64_t fp, unsigned &err) {
    uint64_t bk = ((fp & 0xffffffffull) * nbuckets) >> 32;
    for (int probe = 0; probe < 2048; ++probe) {
        const uint64_t b = bk * SLOTS;
        unsigned long long slot[SLOTS];
#if MC_NT_PROBE & 1
        const mc_ull2 *line = reinterpret_cast<const mc_ull2 *>(table + b);
#pragma unroll
        for (int i = 0; i < SLOTS / 2; ++i) { const mc_ull2 v = __builtin_nontemporal_load(line + i); slot[2 * i] = v.x; slot[2 * i + 1] = v.y; }
#else
        const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + b);
#pragma unroll
        for (int i = 0; i < SLOTS / 2; ++i) { const ulonglong2 v = line[i]; slot[2 * i] = v.x; slot[2 * i + 1] = v.y; }
#endif
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            unsigned long long cur = slot[i];
            if (cur == 0) cur = atomicCAS((unsigned long long *)&table[b + i], 0ull, (unsigned long long)fp);
            if (cur == 0) return true;
            if (cur == fp) return false;
        }
        bk = bk + 1 == nbuckets ? 0 : bk + 1;
    }
    err |= DEV_ETABLE;
    return false;
}
constexpr uint64_t SEEN_SPARSE = 1ull << 63;
#ifndef MC_SPARSE_SLOTS
#define MC_SPARSE_SLOTS 4
#endif
// (the parameter is still called `mask` in the kernels' signatures: it carries the bucket count and the mode bit)
__device__ __forceinline__ bool seen_insert(uint64_t *table, uint64_t nbuckets, uint64_t fp, unsigned &err) {
    if (nbuckets & SEEN_SPARSE) return seen_insert_t<MC_SPARSE_SLOTS>(table, nbuckets & ~SEEN_SPARSE, fp, err);
    return seen_insert_t<8>(table, nbuckets, fp, err);
}

// The synchronous prober as a REAL function: the rare ways out of the split-phase probes of k_expand_family (a candidate whose
// home bucket is full, a compare-and-swap lost to another fingerprint) call it instead of carrying inlined copies of the loop.
// bit 0: the fingerprint is new (inserted here); bit 1: the table is full
__device__ __noinline__ unsigned seen_insert_slow(uint64_t *table, uint64_t nbuckets, uint64_t fp) {
    unsigned e = 0;
    const bool nw = seen_insert(table, nbuckets, fp, e);
    return (nw ? 1u : 0u) | (e ? 2u : 0u);
}

// checkpoint recovery: the seen-set is not part of a checkpoint — it is rebuilt from word 0 (the fingerprint) of the
// arena's states, one coalesced pass
template <class S>
__global__ void __launch_bounds__(256)
k_reseed_table(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t n, uint64_t *table, uint64_t mask, DevCounters *ctr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned err = 0;
    if (i < n) seen_insert(table, mask, S::fp_of(prm, arena_cref(arena, i, S::words(prm))), err);
    if (wave_or_u32(err) && (threadIdx.x & 63) == 0) atomicOr(&ctr->error, DEV_ETABLE);
}

static __global__ void __launch_bounds__(256)
k_insert(const uint64_t *__restrict__ cand, uint64_t row_stride, uint64_t ncols, const uint16_t *__restrict__ nsl,
         uint64_t *table, uint64_t mask, uint32_t *__restrict__ newlist, DevCounters *ctr) {
    const unsigned slot = blockIdx.y;
    if (slot >= ctr->max_slots) return;
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    bool is_new = false;
    unsigned err = 0, probed = 0;
    if (slot < nsl[col]) {
        const uint64_t fp = cand[(uint64_t)slot * row_stride + col];
        if (fp) {
            probed = 1;
            is_new = seen_insert(table, mask, fp, err);
        }
    }
    const unsigned long long ballot = __ballot(is_new);
    const unsigned lane = threadIdx.x & 63;
    unsigned long long base = 0;
    const unsigned np = wave_sum_u32(probed);
    if (lane == 0) {
        if (ballot) base = atomicAdd(&ctr->n_new[0].v, (unsigned long long)__popcll(ballot));
        if (np) atomicAdd(&ctr->cells[0].v, (unsigned long long)np);
    }
    base = __shfl(base, 0);
    if (is_new) {
        const unsigned rank = (unsigned)__popcll(ballot & ((1ull << lane) - 1ull));
        newlist[base + rank] = (uint32_t)col | ((uint32_t)slot << 24);
    }
    if (wave_or_u32(err) && lane == 0) atomicOr(&ctr->error, DEV_ETABLE);
}


// ------------------------------------------------------------------------------------- fused expand + insert
// Per-wavefront LDS ring queues turn the sparse stream of enabled successors into dense work:
//   q   : (fingerprint, source) of generated successors waiting to be probed.  As soon as 64 are
//         queued the whole wavefront probes the seen-set at once (64 independent HBM atomics in
//         flight per wave instead of a few divergent ones).
//   o   : sources of the survivors (new states); flushed to `newlist` 64 at a time with ONE
//         atomicAdd per flush, so the global cursor sees (new states)/64 atomics.
constexpr int QCAP = 128;  // ring capacity per wave (>= 2 * 64)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct WaveQueues {
    uint64_t q_fp[QCAP];
    uint32_t q_src[QCAP], o_src[QCAP];  // (slot << 24) | column: chunks hold <= 2^24 states, specs <= 255 slots
    uint64_t o_fp[QCAP];                // the survivors' fingerprints: k_materialise need not recompute them
};
constexpr int STAGE_MAX = 16;  // words of each parent state staged in LDS per lane (spec-chosen range)

// View of a parent state whose words [lo, lo+n) have been staged in LDS by the owning lane
// (lds points at this lane's column: word w of the range lives at lds[w * 64]).
struct StagedRef {
    // address-space-qualified pointers: the LDS branch must compile to ds_read_b64 and the HBM
    // branch to global_load (a generic pointer would make every access a flat_load)
    const __attribute__((address_space(1))) uint64_t *p;
    size_t stride;
    const __attribute__((address_space(3))) uint64_t *lds;
    int lo, hi;
    __device__ __forceinline__ uint64_t get(int w) const {
        if (w >= lo && w < hi) return lds[(w - lo) * 64];
        return p[(size_t)w * stride];
    }
};

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Sharded (multi-GPU) mode: instead of probing, a flush ROUTES the queued fingerprints to the
// rank that owns them (owner = fingerprint high bits): bucket [owner][shard] in HBM, one
// atomicAdd per (flush, owner present).  k_compact_buckets then makes each owner's bucket
// contiguous for the all-to-all.
// Device-driven levels: while the frontier is small, the host enqueues a batch of levels back to back and the
// kernels read the level's range from this block (no host round trip per level; see Engine::run).
constexpr int BLIND_BATCH = 8;
struct LevelCtl {
    unsigned long long lo, hi;        // frontier of the level about to be expanded
    unsigned long long max_states;    // a batched level handles at most this many states
    unsigned long long max_distinct;  // budget (0 = none)
    unsigned int stop;                // 0 run; 1 finished (empty frontier / violation / error / budget); 2 next level too large
    unsigned int nlev;                // levels completed since the host last looked
    unsigned int levels_left;         // max_levels budget: expansions still allowed (0 = unlimited)
    unsigned int pad;
    unsigned long long level_hi[BLIND_BATCH];  // arena fill level after each completed level
};
struct RouteArgs {
    unsigned nranks;
    PaddedCounter *cursors;   // [nranks * NSHARD]
    uint64_t *rt_fp;          // [nranks * NSHARD][subcap]
    uint32_t *rt_src;
    uint64_t subcap;
    const LevelCtl *lc = nullptr;  // non-null: [lo, hi) come from the device (batched small levels)
    uint64_t *new_fp = nullptr;    // non-null: fingerprints of the new-list entries (same segments, same positions)
    unsigned my_rank = 0;          // route mode: this rank (candidates it owns are probed locally)
    uint16_t *succ = nullptr;      // slot-sliced launch (gridDim.y > 1) with deadlock checking: one "has a successor" flag per column
    // IN-WAVE WRITES (round 4, fused runs of the by-family kernel): non-null = the expand wavefront appends its own survivors to
    // the arena at its tail (it still has the parent block in its caches; k_materialise's second read of every parent row is gone)
    uint64_t *arena_w = nullptr;
    uint64_t arena_cap = 0;
    uint32_t *parent = nullptr;    // MC_F_TRACE: parent pointers of the states written in-wave
    uint16_t *pslot = nullptr;
};

// specs that ask for a per-wavefront duplicate filter in front of the seen-set in the slot-by-slot kernel (S::WAVE_FILTER
// entries, a power of two; see k_expand_family's filter): a candidate found there was queued — hence probed — by this wavefront
template <class S, class = void>
struct WaveFilter : std::integral_constant<int, 0> {};
template <class S>
struct WaveFilter<S, decltype((void)S::WAVE_FILTER)> : std::integral_constant<int, S::WAVE_FILTER> {};

template <class S, bool ROUTE>
__global__ void __launch_bounds__(256)
k_expand_insert(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi, uint64_t ncols,
                uint64_t *table, uint64_t mask, uint32_t *__restrict__ newlist, uint64_t seg_cap, DevCounters *ctr, unsigned flags,
                RouteArgs rt, unsigned parity) {
    __shared__ WaveQueues wq[4];
    __shared__ uint64_t stage[4][S::STAGE_WORDS > 0 ? S::STAGE_WORDS : 1][64];
    constexpr int WF = WaveFilter<S>::value;
    __shared__ uint64_t wfilt[4][WF > 0 ? WF : 1];
    if (rt.lc) {
        if (rt.lc->stop) return;
        lo = rt.lc->lo;
        hi = rt.lc->hi;
        ncols = ((hi - (lo & ~63ull)) + 63) & ~63ull;
    }
    const unsigned lane = threadIdx.x & 63;
    WaveQueues &Q = wq[threadIdx.x >> 6];
    uint64_t *const filt = wfilt[threadIdx.x >> 6];
    const uint64_t base = lo & ~63ull;
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;  // whole wavefronts leave together (ncols % 64 == 0)
    if constexpr (WF > 0) {
#pragma unroll
        for (int t = 0; t < WF / 64; ++t) filt[t * 64 + lane] = 0;  // fingerprint 0 is never a candidate
        wave_lds_fence();
    }
    const uint64_t idx = base + col;
    const bool active = idx >= lo && idx < hi;
    const CWordRef g = arena_cref(arena, idx, S::words(prm));
    // Stage the spec-chosen word range of every parent (raft: the message slots, which each Send
    // scans) in LDS: the loads are independent and coalesced, the later scans hit LDS.
    using Ref = typename std::conditional<(S::STAGE_WORDS > 0), StagedRef, CWordRef>::type;
    Ref s;
    if constexpr (S::STAGE_WORDS > 0) {
        int slo = 0, sn = 0;
        if (active) S::stage_range(prm, g, slo, sn);
        const int wn = min((int)wave_max_u32((unsigned)sn), (int)S::STAGE_WORDS);
        slo = (int)wave_max_u32((unsigned)slo);
        uint64_t *col_lds = &stage[threadIdx.x >> 6][0][lane];
        uint64_t tmp[S::STAGE_WORDS];
#pragma unroll
        for (int w = 0; w < S::STAGE_WORDS; w++) tmp[w] = (active && w < wn) ? g.get(slo + w) : 0;
#pragma unroll
        for (int w = 0; w < S::STAGE_WORDS; w++) if (w < wn) col_lds[w * 64] = tmp[w];
        s = StagedRef{(const __attribute__((address_space(1))) uint64_t *)g.p, g.stride,
                      (const __attribute__((address_space(3))) uint64_t *)col_lds, slo, slo + wn};
    } else {
        s = g;
    }
    typename S::Local loc;
    int ns = 0;
    unsigned long long viol = ~0ull;
    // SLOT SLICES (gridDim.y = SG > 1; specs without unrolled slots only): slice sy evaluates the slots FIX_SLOTS + sy,
    // + SG, ... of the same parents.  A small frontier of a spec with many slots per state (the witness enumeration of the
    // Paxos family: 210; compiled PlusCal programs) is otherwise a handful of wavefronts each walking its slots one after the
    // other — a level costs slots x eval latency while the device idles.  Slice 0 alone evaluates the parent's own status.
    const unsigned sy = blockIdx.y, SG = gridDim.y;
    if (active) {
        S::load(prm, s, loc);
        ns = S::nslots(prm, loc);
        if (sy == 0) {
            const unsigned ps = S::parent_status(prm, loc, s);  // specs that check invariants per expanded state
            if (ps & ST_INVARIANT) viol = viol_key(idx, SLOT_PARENT, VK_INVARIANT, ps >> 8);
        }
    }
    const int wns = (flags & 64u) ? 0 : (int)wave_max_u32((unsigned)ns);  // 64 = ablation: load the parents only
    unsigned gen = 0, err = 0, probes = 0;
    unsigned qhead = 0, qn = 0, ohead = 0, on = 0;  // wave-uniform ring state
    const unsigned shard = blockIdx.x & (NSHARD - 1), pshard = parity * NSHARD + shard;
    uint32_t *__restrict__ seg = newlist + (uint64_t)pshard * seg_cap;  // this shard's new-list segment

    auto flush_out = [&](unsigned take) {  // append `take` survivors to the global new-list
        unsigned long long pos = 0;
        if (lane == 0) pos = atomicAdd(&ctr->n_new[pshard].v, (unsigned long long)take);
        pos = __shfl(pos, 0);
        if (lane < take) {
            seg[pos + lane] = Q.o_src[(ohead + lane) & (QCAP - 1)];
            if (rt.new_fp) rt.new_fp[(uint64_t)pshard * seg_cap + pos + lane] = Q.o_fp[(ohead + lane) & (QCAP - 1)];
        }
        ohead = (ohead + take) & (QCAP - 1);
        on -= take;
    };
    auto flush_probe = [&](unsigned take) {  // probe (or route) `take` queued fingerprints, one per lane
        bool is_new = false;
        uint32_t src = 0;
        uint64_t qfp = 0;
        if (lane < take) {
            const unsigned k = (qhead + lane) & (QCAP - 1);
            src = Q.q_src[k];
            qfp = Q.q_fp[k];
            if constexpr (!ROUTE) is_new = (flags & 16u) ? false : seen_insert(table, mask, qfp, err);  // 16 = ablation: no probes
        }
        qhead = (qhead + take) & (QCAP - 1);
        qn -= take;
        probes += take;
        if constexpr (ROUTE) {
            // LOCAL-OWNER SHORTCUT: a candidate this rank owns is probed right here, like on one GPU, and a new one goes to the
            // rank's own new-list (materialised locally, it never travels); only candidates of OTHER owners are routed.  On P
            // ranks 1/P of the candidates skip the exchange; on one rank the sharded engine does exactly the fused engine's work.
            unsigned owner = lane < take ? fp_owner(qfp, rt.nranks) : 0xffffffffu;
            if (owner == rt.my_rank) {
                is_new = (flags & 16u) ? false : seen_insert(table, mask, qfp, err);
                owner = 0xffffffffu;
            }
            {
                const unsigned long long b = __ballot(is_new);
                if (is_new) {
                    const unsigned k = (ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (QCAP - 1);
                    Q.o_src[k] = src;
                    Q.o_fp[k] = qfp;
                }
                on += (unsigned)__popcll(b);
                wave_lds_fence();
                if (on >= 64) flush_out(64);
            }
            // one round trip for all remote owners: lane t reserves the bucket space of owner t (the P atomics issue together
            // instead of one after the other), then every candidate takes its owner's base from that lane
            unsigned my_rank = 0, my_cnt = 0;
            for (unsigned t = 0; t < rt.nranks; ++t) {
                const unsigned long long b = __ballot(owner == t);
                if (owner == t) my_rank = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
                if (lane == t) my_cnt = (unsigned)__popcll(b);
            }
            unsigned long long base = 0;
            if (lane < rt.nranks && my_cnt) base = atomicAdd(&rt.cursors[lane * NSHARD + shard].v, (unsigned long long)my_cnt);
            base = __shfl(base, (int)(owner < rt.nranks ? owner : 0u));
            if (owner < rt.nranks) {
                const unsigned bucket = owner * NSHARD + shard;
                const unsigned long long pos = base + my_rank;
                if (pos < rt.subcap) {
                    rt.rt_fp[(uint64_t)bucket * rt.subcap + pos] = qfp;
                    rt.rt_src[(uint64_t)bucket * rt.subcap + pos] = src;
                } else {
                    err |= DEV_EROUTE;  // a full route sub-bucket is "more candidates than the allowance" (restart), not a full arena
                }
            }
        } else {
            const unsigned long long b = __ballot(is_new);
            if (is_new) {
                const unsigned k = (ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (QCAP - 1);
                Q.o_src[k] = src;
                Q.o_fp[k] = qfp;
            }
            on += (unsigned)__popcll(b);
            wave_lds_fence();
            if (on >= 64) flush_out(64);
        }
    };

    auto body = [&](int slot) __attribute__((always_inline)) {
        uint64_t fp = 0;
        if (slot < ns) {
            uint64_t f = 0;
            const unsigned st = S::eval(prm, loc, s, slot, f);
            if (st & ST_ENABLED) {
                ++gen;
                if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                else if (st & ST_ASSERT) viol = min(viol, viol_key(idx, (unsigned)slot, VK_ASSERT, 0));
                else if (st & ST_SPECERR) viol = min(viol, viol_key(idx, (unsigned)slot, VK_SPECERR, 0));
                else {
                    if (st & ST_INVARIANT) viol = min(viol, viol_key(idx, (unsigned)slot, VK_INVARIANT, st >> 8));
                    if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) fp = f;
                }
            }
        }
        if constexpr (WF > 0) {
            if (fp && !(flags & 8192u)) {  // 8192 = A/B: no duplicate filter
                const unsigned h = (unsigned)(fp >> 20) & (unsigned)(WF - 1);
                if (filt[h] == fp) fp = 0;  // this wavefront has queued it before
                else filt[h] = fp;
            }
        }
        const unsigned long long b = __ballot(fp != 0);
        if (b) {
            if (fp) {
                const unsigned k = (qhead + qn + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (QCAP - 1);
                Q.q_fp[k] = fp;
                Q.q_src[k] = (uint32_t)col | ((uint32_t)slot << 24);
            }
            qn += (unsigned)__popcll(b);
            wave_lds_fence();
            if (qn >= 64) flush_probe(64);
        }
    };
    // slots whose action / server indices are compile-time constants: fully unrolled, so the
    // spec's dispatch and register-array indexing fold away; the rest (per-message slots) loops
    if (wns > 0) static_for<0, S::FIX_SLOTS>([&](auto c) __attribute__((always_inline)) { body(decltype(c)::value); });
    for (int slot = S::FIX_SLOTS + (int)sy; slot < wns; slot += (int)SG) body(slot);
    if (qn) flush_probe(qn);
    if (on) flush_out(on);

    if (SG == 1) {
        if (active && gen == 0 && (flags & MC_F_DEADLOCK)) viol = min(viol, viol_key(idx, SLOT_NONE, VK_DEADLOCK, 0));
    } else if (rt.succ && active && gen) {
        rt.succ[col] = 1;  // a deadlock is the absence of a successor in EVERY slice: k_deadlock_slices looks at the flags
    }
    const unsigned gsum = wave_sum_u32(gen);
    const unsigned long long vmin = wave_min_u64(viol);
    const unsigned eor = wave_or_u32(err);
    if (lane == 0) {
        if (gsum) atomicAdd(&ctr->generated[shard].v, (unsigned long long)gsum);
        if (probes) atomicAdd(&ctr->cells[shard].v, (unsigned long long)probes);
        if (vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
        if (eor) atomicOr(&ctr->error, eor);
    }
}


// ------------------------------------------------------------------------------------- expand BY ACTION FAMILY
// For specs with many action kinds (raft).  In k_expand_insert every slot body runs for the
// whole wavefront as soon as ONE lane is enabled — about a quarter of the lanes do useful work, and the
// Receive slot executes every message handler in turn.  Here the work is split three ways:
//   dense    lane = parent: the slots nearly every state enables (raft: Restart / Timeout), evaluated in pairs;
//   inline   lane = parent: the actions on a parent's IN-FLIGHT messages (raft: Receive / Duplicate / Drop of at most MaxMsgs
//            messages, whatever the size of the bag), the message word and its hash shared by the three;
//   phase A  lane = parent: cheap, exact guards of the sparse fixed slots; each enabled (parent lane, slot) pair is appended
//            to the LDS ring queue of its action family;
//   phase B  as soon as a family has 64 pairs queued, the wavefront evaluates 64 pairs of THAT family — every
//            lane busy, one code path — reading the pair's parent straight from the arena (the 64 parents of a
//            wavefront are one arena block, so lanes reading word w of different parents hit one 512-byte row).
// The fingerprints then go through the same probe / route queues as in k_expand_insert.
constexpr int FQCAP = 128;

// Phase profile of k_expand_family (build with -DMC_PHASE_PROF: profiles/phase_prof.sh — rocprofv3's PC sampling is not
// available for gfx950 in this image).  Every wavefront accumulates shader-clock cycles per phase — nested phases are exclusive:
// switching to a phase charges the time since the last switch to the phase that was current — and adds them to g_phase at exit.
//   0 load_expand + summarize | 1 dense pairs (Restart / Timeout) | 2 enqueue (filter, probe ring) | 3 flush_probe (seen-set)
//   4 flush_out (new-list) | 5 push loop of the fixed slots | 6 push loop of the message slots | 7 epilogue
//   8 + f: phase B of family f (eval_pair) | 24 + f: pairs evaluated of family f | 40: wavefronts
#ifdef MC_PHASE_PROF
__device__ unsigned long long g_phase[48];
#define MC_PROF_DECL unsigned long long pf_t = wall_clock64(), pf_acc[24] = {}; int pf_cur = 0; unsigned long long pf_pairs[16] = {};
#define MC_PROF(ph) do { const unsigned long long pf_n = wall_clock64(); pf_acc[pf_cur] += pf_n - pf_t; pf_t = pf_n; pf_cur = (ph); } while (0)
#define MC_PROF_PAIRS(f, n) do { pf_pairs[(f)] += (n); } while (0)
#define MC_PROF_END do { MC_PROF(7); if (lane == 0) { for (int q_ = 0; q_ < 24; ++q_) if (pf_acc[q_]) atomicAdd(&g_phase[q_], pf_acc[q_]); \
    for (int q_ = 0; q_ < 16; ++q_) if (pf_pairs[q_]) atomicAdd(&g_phase[24 + q_], pf_pairs[q_]); atomicAdd(&g_phase[40], 1ull); } } while (0)
#else
#define MC_PROF_DECL
#define MC_PROF(ph) do { } while (0)
#define MC_PROF_PAIRS(f, n) do { } while (0)
#define MC_PROF_END do { } while (0)
#endif

// number of leading fixed slots a by-family spec wants evaluated inline, lane = parent (S::DENSE_SLOTS; 0 if absent)
template <class S, class = void>
struct DenseSlots : std::integral_constant<int, 0> {};
template <class S>
struct DenseSlots<S, decltype((void)S::DENSE_SLOTS)> : std::integral_constant<int, S::DENSE_SLOTS> {};

// The wavefront's own duplicate filter: a direct-mapped table of the fingerprints it has already queued for the seen-set.  The
// successors of 64 neighbouring parents repeat each other (two actions that commute reach the same state from two siblings:
// 30 % of the candidates of a wavefront, measured on the bench model in BFS order); a candidate found here was probed — found or
// inserted — by this very wavefront, so it is dropped before it costs a random 64-byte read of HBM.  Sound: an entry is only
// ever a fingerprint this wavefront handed to the seen-set.
constexpr int WFILT = 256;
// specs whose message actions are evaluated inline, lane = parent (S::inflight_slots; see k_expand_family)
template <class S, class = void>
struct InlineMsgs : std::false_type {};
template <class S>
struct InlineMsgs<S, decltype((void)&S::inflight_slots)> : std::true_type {};

// specs whose expand kernel hands the successor's fingerprint to the writer (S::apply_known_fp)
template <class S, class = void>
struct HasKnownFp : std::false_type {};
template <class S>
struct HasKnownFp<S, decltype((void)S::KNOWN_FP)> : std::true_type {};
// The writer of the in-wave tail, a REAL function (not inlined): the copy-and-patch writer wants 160+ VGPRs on its own, and
// inlined into k_expand_family it drags the register allocation of the whole kernel down with it (101 spilled VGPRs against 1).
// Behind a call it is allocated by itself, and at the call site — the wavefront's tail — nothing is live that would have to be
// saved.  Arguments of a device function travel in vector registers, so the wave-uniform ones are made scalar again here.
template <class T>
__device__ __forceinline__ T wave_uniform_copy(const T &v) {
    static_assert(sizeof(T) % 4 == 0, "copied in 32-bit words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}
// specs that name successors which are generated (counted) but provably never stored, so that the kernel need not evaluate them
template <class S, class = void>
struct HasGeneratedOnly : std::false_type {};
template <class S>
struct HasGeneratedOnly<S, decltype((void)S::GENERATED_ONLY)> : std::true_type {};
// specs whose writer starts from what the parent's lane derived (S::Summary in LDS) instead of walking the row again
template <class S, class = void>
struct HasSummaryWriter : std::false_type {};
template <class S>
struct HasSummaryWriter<S, decltype((void)S::SUMMARY_WRITER)> : std::true_type {};
template <class S>
__device__ __noinline__ void wave_write_survivors(typename S::Params prm_v, const uint64_t *arena_v, uint64_t pidx, bool mine, unsigned slot, uint64_t fp,
                                                  uint64_t *arena_w_v, uint64_t oidx, typename S::Summary q) {
    const typename S::Params prm = wave_uniform_copy(prm_v);
    const uint64_t *arena = (const uint64_t *)uniform_ptr(arena_v);
    uint64_t *arena_w = (uint64_t *)uniform_ptr(arena_w_v);
    if (!mine) return;
    const int W = S::words(prm);
    const CWordRef sp = arena_cref(arena, pidx, W);
    if constexpr (HasSummaryWriter<S>::value) S::apply_summary_patch(prm, q, sp, (int)slot, fp, arena_ref(arena_w, oidx, W));
    else if constexpr (HasKnownFp<S>::value) S::apply_known_fp(prm, sp, (int)slot, fp, arena_ref(arena_w, oidx, W));
    else S::apply(prm, sp, (int)slot, arena_ref(arena_w, oidx, W));
}
// classes of action slots whose successor construction shares a code path (S::NCLS, S::slot_class): the workgroup's tail sorts
// its survivors by class, so that the 64 lanes of a batch walk one or two branches of the writer instead of all of them
template <class S, class = void>
struct SlotClasses : std::integral_constant<int, 1> {
    __device__ __forceinline__ static int of(int) { return 0; }
};
template <class S>
struct SlotClasses<S, decltype((void)S::NCLS)> : std::integral_constant<int, S::NCLS> {
    __device__ __forceinline__ static int of(int slot) { return S::slot_class(slot); }
};

// Probe ring and survivor list of a by-family wavefront.  An entry names its (parent, slot) pair inside the wavefront's own
// arena block: (slot << 6) | parent lane, 16 bits.  The survivor list holds up to OCAP entries: with in-wave writes the
// survivors wait here until the wavefront's tail (one per parent on average; 64 are moved to the global new-list — the
// overflow path, k_materialise — only when the list is about to fill up).
constexpr int OCAP = 256;
// SPLIT-PHASE PROBES (round 5; MC_ASYNC_PROBE: 0 = off, 1 = loads, 2 = loads + compare-and-swaps).  A seen-set probe is two dependent
// trips to HBM — read the bucket, then compare-and-swap the fingerprint into its first empty slot — and until round 4 a wavefront
// sat through both with nothing else to do (flush_probe: 6 of the ~10 HBM-class waits of a wavefront's life).  Now the 64 queued
// candidates of a batch ISSUE their read as an LDS-DMA (global_load_lds_dwordx4: 16 bytes = the first two slots of the 32-byte
// bucket per lane, no VGPRs held while it flies) and the wavefront goes on generating; when the next 64 candidates are queued the
// batch is RESOLVED from LDS — match: dropped; an empty slot: compare-and-swap; both slots taken by others: the candidate stays at
// the head of the ring with its displacement bumped and reads the next 16 bytes with the next batch (the probe sequence and the
// "first empty slot" rule are the synchronous prober's: the table format does not change).  With MC_ASYNC_PROBE = 2 the
// compare-and-swap is not waited for either: the candidate goes to the survivor list as TENTATIVE, the returned word stays in two
// VGPRs, and the next resolve step confirms it (0: new; its own fingerprint: somebody else inserted it first — the entry becomes a
// tombstone, O_DEAD; anything else: the synchronous prober decides).  Only sparse tables (32-byte buckets) of fused runs.
#ifndef MC_ASYNC_PROBE
#define MC_ASYNC_PROBE 0
#endif
#ifndef MC_FOLD_MSG
#define MC_FOLD_MSG 0   // (one loop over the message slots instead of two: 14 KB less code, but 14 spilled VGPRs)
#endif
#ifndef MC_FOLD_FIX
#define MC_FOLD_FIX 0   // (the drain of the family queues as the last step of the fixed-slot loop: 14 KB less code, no spills — and
#endif                  //  147.3 -> 153.6 ms per step on the t3 graph, profiles/r05b_ab.jsonl: not adopted)
constexpr unsigned O_DEAD = 0xffffu;    // survivor-list tombstone: a tentative survivor that turned out to be known
constexpr unsigned Q_DSP_SHIFT = 14;    // probe-ring entries: bits [14, 16) = 16-byte steps already taken past the home bucket's first half
struct FamQueues {
    uint64_t q_fp[QCAP];
    uint64_t o_fp[OCAP];
    uint16_t q_ent[QCAP], o_ent[OCAP];
};

template <class S, int NB>
struct FamLds {
    uint16_t fq[S::NFAM][FQCAP];   // (slot << 8) | (block << 6) | parent lane
    typename S::Summary sum[NB * 64];
    uint64_t filt[WFILT];
    // (deadlock check: "this parent has a successor" is a register of the parent's own lane for everything that lane evaluates,
    //  and bit 31 of a word of its Summary — S::succ_word — for the pairs another lane evaluates in a family batch)
};

template <class S, int F, class Fn>
__device__ __forceinline__ void family_dispatch(int fam, Fn &&fn) {
    if constexpr (F < S::NFAM) {
        if (fam == F) fn(std::integral_constant<int, F>{});
        else family_dispatch<S, F + 1>(fam, fn);
    }
}

// NB = arena blocks (of 64 parents) one wavefront works through.  The family queues live across the blocks and are
// drained once at the end, so the partially filled batches of the drain (up to one per family) are paid once per
// NB * 64 parents instead of once per 64: phase B's lane utilisation goes from ~80 % (NB = 1) towards 95 % (NB = 4).
// MINW = wavefronts per SIMD the register allocation leaves room for: 4 = at most 128 VGPRs (no spills), 5 = at most 96
// (a few dozen spilled VGPRs, one more wavefront per SIMD to hide the probe / gather latency behind)
#ifndef MC_EXPAND_MINW
#define MC_EXPAND_MINW 4
#endif
// WAVES = wavefronts per workgroup.  The workgroup only matters to the in-wave tail (its wavefronts pool their survivors behind a
// barrier): 4 = the widest pool (a batch of 64 sorted survivors holds one or two action classes) but four wavefronts wait for the
// slowest; 2 = a tail per PAIR of wavefronts.
#ifndef MC_EXPAND_WAVES
#define MC_EXPAND_WAVES 4
#endif
template <class S, bool ROUTE, int NB, int MINW = MC_EXPAND_MINW, int WAVES = MC_EXPAND_WAVES>
__global__ void __launch_bounds__(64 * WAVES, MINW)
k_expand_family(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi, uint64_t ncols,
                uint64_t *table, uint64_t mask, uint32_t *__restrict__ newlist, uint64_t seg_cap, DevCounters *ctr, unsigned flags,
                RouteArgs rt, unsigned parity) {
    static_assert(NB >= 1 && NB <= 4, "the queue entry has two bits for the block");
    static_assert(WAVES == 1 || WAVES == 2 || WAVES == 4, "wavefronts per workgroup");
    __shared__ FamQueues wq[WAVES];
    __shared__ FamLds<S, NB> fls[WAVES];
    if (rt.lc) {
        if (rt.lc->stop) return;
        lo = rt.lc->lo;
        hi = rt.lc->hi;
        ncols = ((hi - (lo & ~63ull)) + 63) & ~63ull;
    }
    const unsigned lane = threadIdx.x & 63;
    FamQueues &Q = wq[threadIdx.x >> 6];
    FamLds<S, NB> &FL = fls[threadIdx.x >> 6];
    const uint64_t base = lo & ~63ull;
    // (An XCD-aware tile order — XCD k = workgroup id % 8 walks the k-th eighth of the chunk's tiles, so that neighbouring arena
    //  blocks, which generate many of the same successors, probe the seen-set through ONE L2 — was measured on the t3 and K = 10
    //  graphs: 168.8 against 166.4 ms and 35.0 against 34.6 ms per step, i.e. nothing: the probes that repeat within a
    //  neighbourhood are already caught by the wavefront's own filter, the rest miss every L2.)
    // this wavefront's first column: NB consecutive arena blocks
    const uint64_t wave_col0 = ((uint64_t)blockIdx.x * WAVES + (threadIdx.x >> 6)) * (64ull * NB);
    const bool inwave = !ROUTE && rt.arena_w != nullptr;  // wave-uniform (a kernel argument)
    if (wave_col0 >= ncols && !inwave) return;  // (in-wave writes: the workgroup's tail has barriers — a wavefront without parents walks through the empty loops below)
    const uint64_t wave_idx0 = base + wave_col0;
    const int W = S::words(prm);
    static_assert(NB == 1, "BlockRef addresses ONE arena block per wavefront");
    const GlobalWords blk_base = uniform_ptr(arena + (wave_idx0 >> 6) * (uint64_t)W * 64);
    unsigned long long viol = ~0ull;
    unsigned gen = 0, err = 0, probes = 0, cands = 0;  // cands: in-model successors (the algorithmic look-ups); probes: after the filter
    unsigned qhead = 0, qn = 0, ohead = 0, on = 0;  // wave-uniform ring state of the probe / survivor queues
#pragma unroll
    for (int t = 0; t < WFILT / 64; ++t) FL.filt[t * 64 + lane] = 0;  // fingerprint 0 is never a candidate
    bool track_succ = (flags & MC_F_DEADLOCK) != 0;  // cleared once the dense slots gave every parent of the block a successor
    bool lane_succ = false;                          // this lane's parent has a successor (NB == 1: one parent per lane)
    // ring state of the family queues, wave-uniform, packed 8 bits per family so that a run-time family index is a
    // scalar shift (no LDS round trip): heads and counts of families 0..7 in *A, 8.. in *B
    uint64_t fheadA = 0, fheadB = 0, fcntA = 0, fcntB = 0;
    auto fget = [](uint64_t a, uint64_t b, int f) -> unsigned { return (unsigned)((f < 8 ? a >> (8 * f) : b >> (8 * (f - 8))) & 255u); };
    auto fset = [](uint64_t &a, uint64_t &b, int f, unsigned v) {
        if (f < 8) a = (a & ~(255ull << (8 * f))) | ((uint64_t)v << (8 * f));
        else b = (b & ~(255ull << (8 * (f - 8)))) | ((uint64_t)v << (8 * (f - 8)));
    };
    const unsigned shard = blockIdx.x & (NSHARD - 1), pshard = parity * NSHARD + shard;
    uint32_t *__restrict__ seg = newlist + (uint64_t)pshard * seg_cap;
    // survivors kept in LDS before a batch of 64 goes to the global new-list: all the list holds minus one probe batch (in-wave
    // writes: the global list is the overflow path), or one batch (everything goes through the new-list)
    const unsigned okeep = inwave ? (unsigned)(OCAP - 64) : 63u;
    // split-phase probes (MC_ASYNC_PROBE above): wave-uniform state
    constexpr bool ASYNC_BUILD = !ROUTE && MC_ASYNC_PROBE > 0 && MC_SPARSE_SLOTS == 4;
    constexpr bool ASYNC_CAS = ASYNC_BUILD && MC_ASYNC_PROBE > 1;
    __shared__ __attribute__((aligned(16))) uint64_t probe_land[WAVES][ASYNC_BUILD ? 128 : 2];  // 16 bytes per lane: where the DMA lands
    constexpr bool async_probe = ASYNC_BUILD;
    const bool async_cas = ASYNC_CAS && inwave;  // (a tentative survivor must not reach the global new-list)
    const unsigned hshift = (mask & SEEN_SPARSE) ? 1u : 2u;        // 16-byte halves per bucket: 2 (32-byte buckets) or 4 (64-byte)
    const uint64_t nhalves = (mask & ~SEEN_SPARSE) << hshift;
    const bool probe_at_once = (flags & MC_F_SYNCPROBE) != 0;      // A/B: resolve a batch at the next candidate instead of a batch later
    unsigned pend = 0;               // the first `pend` entries of the probe ring have their 16 bytes on the way to LDS
    unsigned long long casmask = 0;  // lanes whose compare-and-swap is in flight; lane's tentative survivor: list position cas_obase + rank
    unsigned cas_obase = 0;
    unsigned long long casret = 0;   // what that compare-and-swap returns (per lane; waited for at its first use, one resolve step later)
    uint64_t *const land = probe_land[threadIdx.x >> 6];
    MC_PROF_DECL

    auto flush_out = [&](unsigned take) __attribute__((always_inline)) {
        MC_PROF(4);
        // (the `take` oldest entries: all confirmed — tentative ones are the newest — but some may be tombstones)
        const unsigned e = lane < take ? Q.o_ent[(ohead + lane) & (OCAP - 1)] : O_DEAD;
        const unsigned long long bl = __ballot(e != O_DEAD);
        unsigned long long pos = 0;
        if (lane == 0 && bl) pos = atomicAdd(&ctr->n_new[pshard].v, (unsigned long long)__popcll(bl));
        pos = __shfl(pos, 0) + (unsigned)__popcll(bl & ((1ull << lane) - 1ull));
        if (e != O_DEAD) {
            seg[pos] = (uint32_t)(wave_col0 + (e & 63u)) | ((uint32_t)(e >> 6) << 24);
            if (rt.new_fp) rt.new_fp[(uint64_t)pshard * seg_cap + pos] = Q.o_fp[(ohead + lane) & (OCAP - 1)];
        }
        ohead = (ohead + take) & (OCAP - 1);
        on -= take;
        MC_PROF(3);
    };
    auto flush_probe = [&](unsigned take) __attribute__((always_inline)) {
      if constexpr (!ASYNC_BUILD) {
        MC_PROF(3);
        bool is_new = false;
        unsigned src = 0;  // (slot << 6) | parent lane
        uint64_t qfp = 0;
        if (lane < take) {
            const unsigned k = (qhead + lane) & (QCAP - 1);
            src = Q.q_ent[k];
            qfp = Q.q_fp[k];
            if constexpr (!ROUTE) is_new = (flags & 16u) ? false : seen_insert(table, mask, qfp, err);
        }
        qhead = (qhead + take) & (QCAP - 1);
        qn -= take;
        probes += take;
        if constexpr (ROUTE) {
            // LOCAL-OWNER SHORTCUT: a candidate this rank owns is probed right here, like on one GPU, and a new one goes to the
            // rank's own new-list (materialised locally, it never travels); only candidates of OTHER owners are routed.  On P
            // ranks 1/P of the candidates skip the exchange; on one rank the sharded engine does exactly the fused engine's work.
            unsigned owner = lane < take ? fp_owner(qfp, rt.nranks) : 0xffffffffu;
            if (owner == rt.my_rank) {
                is_new = (flags & 16u) ? false : seen_insert(table, mask, qfp, err);
                owner = 0xffffffffu;
            }
            {
                const unsigned long long b = __ballot(is_new);
                if (is_new) {
                    const unsigned k = (ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (OCAP - 1);
                    Q.o_ent[k] = (uint16_t)src;
                    Q.o_fp[k] = qfp;
                }
                on += (unsigned)__popcll(b);
                wave_lds_fence();
                if (on > okeep) flush_out(64);
            }
            // one round trip for all remote owners: lane t reserves the bucket space of owner t (the P atomics issue together
            // instead of one after the other), then every candidate takes its owner's base from that lane
            unsigned my_rank = 0, my_cnt = 0;
            for (unsigned t = 0; t < rt.nranks; ++t) {
                const unsigned long long b = __ballot(owner == t);
                if (owner == t) my_rank = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
                if (lane == t) my_cnt = (unsigned)__popcll(b);
            }
            unsigned long long base = 0;
            if (lane < rt.nranks && my_cnt) base = atomicAdd(&rt.cursors[lane * NSHARD + shard].v, (unsigned long long)my_cnt);
            base = __shfl(base, (int)(owner < rt.nranks ? owner : 0u));
            if (owner < rt.nranks) {
                const unsigned bucket = owner * NSHARD + shard;
                const unsigned long long pos = base + my_rank;
                if (pos < rt.subcap) {
                    rt.rt_fp[(uint64_t)bucket * rt.subcap + pos] = qfp;
                    rt.rt_src[(uint64_t)bucket * rt.subcap + pos] = (uint32_t)(wave_col0 + (src & 63u)) | ((uint32_t)(src >> 6) << 24);
                } else {
                    err |= DEV_EROUTE;  // a full route sub-bucket is "more candidates than the allowance" (restart), not a full arena
                }
            }
        } else {
            const unsigned long long b = __ballot(is_new);
            if (is_new) {
                const unsigned k = (ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (OCAP - 1);
                Q.o_ent[k] = (uint16_t)src;
                Q.o_fp[k] = qfp;
            }
            on += (unsigned)__popcll(b);
            wave_lds_fence();
            if (on > okeep) flush_out(64);
        }
        MC_PROF(2);
      }
    };
    // ---- split-phase probes: issue / resolve (MC_ASYNC_PROBE above)
    auto half_of_entry = [&](uint64_t fp, unsigned ent) __attribute__((always_inline)) -> uint64_t {  // the 16-byte half this ring entry looks at next
        uint64_t h = ((((fp & 0xffffffffull) * (mask & ~SEEN_SPARSE)) >> 32) << hshift) + (ent >> Q_DSP_SHIFT);
        if (h >= nhalves) h -= nhalves;
        return h;
    };
    auto slow_insert = [&](uint64_t fp) __attribute__((always_inline)) -> bool {
        const unsigned r = seen_insert_slow(table, mask, fp);
        if (r & 2u) err |= DEV_ETABLE;
        return (r & 1u) != 0;
    };
    auto probe_issue = [&](unsigned take) __attribute__((always_inline)) {
        if constexpr (ASYNC_BUILD) {
            MC_PROF(3);
            if (lane < take) {
                const unsigned k = (qhead + lane) & (QCAP - 1);
                const uint64_t h = half_of_entry(Q.q_fp[k], Q.q_ent[k]);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(table + h * 2),
                                                 (__attribute__((address_space(3))) void *)land, 16, 0, (MC_NT_PROBE & 2) ? 2 : 0);
            }
            pend = take;
            MC_PROF(2);
        }
    };
    // resolve the batch whose reads were issued a flush ago; at most `room` candidates that must look at the next 16 bytes stay in
    // the ring (at its head, in front of the fresh ones), the others fall back to the synchronous prober.  First, the tentative
    // survivors of the PREVIOUS resolve step are confirmed: 0 came back = new; the fingerprint itself = somebody else inserted it
    // in between (tombstone); another fingerprint took the slot = the synchronous prober decides.
    auto probe_resolve = [&](unsigned room) __attribute__((always_inline)) {
        if constexpr (ASYNC_BUILD) {
            MC_PROF(3);
            if constexpr (ASYNC_CAS) {
                if (casmask) {
                    if ((casmask >> lane & 1ull) && casret != 0ull) {
                        const unsigned pos = (cas_obase + (unsigned)__popcll(casmask & ((1ull << lane) - 1ull))) & (OCAP - 1);
                        const uint64_t f = Q.o_fp[pos];
                        bool nw = false;
                        if (casret != f) nw = slow_insert(f);
                        if (!nw) Q.o_ent[pos] = (uint16_t)O_DEAD;
                    }
                    casmask = 0;
                    wave_lds_fence();
                }
            }
            bool is_new = false, again = false, cas = false, slow = false;
            uint64_t fp = 0, half = 0;
            unsigned ent = 0, which = 0;
            // the DMA is NOT tracked by the compiler's wait-count insertion (the landing area is read by ordinary LDS loads): wait
            // for it here — and keep memory accesses from moving across the wait
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane < pend) {
                const unsigned k = (qhead + lane) & (QCAP - 1);
                ent = Q.q_ent[k];
                fp = Q.q_fp[k];
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(land + lane * 2);
                if (v.x != fp && v.y != fp && !(flags & 16u)) {  // (16 = ablation: no probes)
                    if (v.x == 0ull || v.y == 0ull) { cas = true; which = v.x != 0ull ? 1u : 0u; half = half_of_entry(fp, ent); }
                    else if ((ent >> Q_DSP_SHIFT) < 3u) again = true;
                    else slow = true;
                }
            }
            const unsigned long long ba = __ballot(again);
            unsigned r = (unsigned)__popcll(ba);
            const unsigned arank = (unsigned)__popcll(ba & ((1ull << lane) - 1ull));
            if (r > room) {  // (rare: the ring has no room for all of them)
                if (again && arank >= room) { again = false; slow = true; }
                r = room;
            }
            if (again) {
                const unsigned k2 = (qhead + pend - r + arank) & (QCAP - 1);
                Q.q_fp[k2] = fp;
                Q.q_ent[k2] = (uint16_t)(ent + (1u << Q_DSP_SHIFT));
            }
            qhead = (qhead + pend - r) & (QCAP - 1);
            qn -= pend - r;
            probes += pend;
            pend = 0;
            unsigned long long cur = 0;
            if (cas) cur = atomicCAS((unsigned long long *)(table + half * 2 + which), 0ull, (unsigned long long)fp);
            if (!async_cas) {
                // (the empty asm keeps the compiler from evaluating `cur` speculatively outside this branch — it did, as a select —
                //  which would wait for the compare-and-swap right where it was issued)
                asm volatile("" : "+v"(cur));
                if (cas) {
                    if (cur == 0ull) is_new = true;
                    else if (cur != fp) slow = true;
                }
            }
            if (slow) is_new = slow_insert(fp);
            {
                const unsigned long long bn = __ballot(is_new);
                if (bn) {
                    if (is_new) {
                        const unsigned k = (ohead + on + (unsigned)__popcll(bn & ((1ull << lane) - 1ull))) & (OCAP - 1);
                        Q.o_ent[k] = (uint16_t)(ent & ((1u << Q_DSP_SHIFT) - 1u));
                        Q.o_fp[k] = fp;
                    }
                    on += (unsigned)__popcll(bn);
                }
            }
            if constexpr (ASYNC_CAS) {
                if (async_cas) {  // tentative survivors: the newest entries of the list, confirmed by the next resolve step
                    const unsigned long long bc = __ballot(cas);
                    if (cas) {
                        const unsigned k = (ohead + on + (unsigned)__popcll(bc & ((1ull << lane) - 1ull))) & (OCAP - 1);
                        Q.o_ent[k] = (uint16_t)(ent & ((1u << Q_DSP_SHIFT) - 1u));
                        Q.o_fp[k] = fp;
                        casret = cur;
                    }
                    casmask = bc;
                    cas_obase = (ohead + on) & (OCAP - 1);
                    on += (unsigned)__popcll(bc);
                }
            }
            wave_lds_fence();
            if (on > okeep) flush_out(64);
            MC_PROF(2);
        }
    };
    // append the lanes of `b` (each with its slot) to family f's queue; returns true when it holds >= 64 pairs
    auto fam_push = [&](int f, unsigned long long b, bool mine, unsigned entry) __attribute__((always_inline)) -> bool {
        const unsigned h = fget(fheadA, fheadB, f), c = fget(fcntA, fcntB, f);
        if (mine) FL.fq[f][(h + c + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (FQCAP - 1)] = (uint16_t)entry;
        const unsigned nc = c + (unsigned)__popcll(b);
        fset(fcntA, fcntB, f, nc);
        return nc >= 64;
    };
    // a lane's candidate (fp != 0) joins the probe ring; 64 queued candidates are probed (or routed) at once
    auto enqueue = [&](uint64_t fp, unsigned src /* (slot << 6) | parent lane */, int back_to) __attribute__((always_inline)) {
        MC_PROF(2);
        const unsigned long long b0 = __ballot(fp != 0);
        if (b0) {
            cands += (unsigned)__popcll(b0);
            if (fp && !(flags & 8192u)) {  // 8192 = A/B: no duplicate filter
                const unsigned h = (unsigned)(fp >> 20) & (WFILT - 1);
                if (FL.filt[h] == fp) fp = 0;  // this wavefront has queued it before
                else FL.filt[h] = fp;
            }
            // (measured and NOT adopted, round 3: touching the candidate's home bucket here, so that the probe a few microseconds
            //  later hits L2 — 40.9 ms per step against 38.5 without, profiles/r03g: the extra request per candidate costs more
            //  than the latency it hides)
            const unsigned long long b = __ballot(fp != 0);
            const unsigned nb = (unsigned)__popcll(b);
            // split-phase: the ring holds the batch in flight (`pend` entries) and the fresh candidates behind it; as soon as the
            // fresh ones make a batch, the batch in flight is resolved (its reads were issued ~64 candidates ago) and the next one issued
            if (async_probe && pend && (qn - pend + nb >= 64 || probe_at_once)) probe_resolve((unsigned)QCAP - (qn - pend) - nb);
            if (fp) {
                const unsigned k = (qhead + qn + (unsigned)__popcll(b & ((1ull << lane) - 1ull))) & (QCAP - 1);
                Q.q_fp[k] = fp;
                Q.q_ent[k] = (uint16_t)src;
            }
            qn += nb;
            wave_lds_fence();
            if (async_probe) {
                if (!pend && qn >= 64) probe_issue(64);
            } else if (qn >= 64) flush_probe(64);
        }
        MC_PROF(back_to);
        (void)back_to;
    };
    // phase B: evaluate `take` queued pairs of family f (f is wave-uniform)
    auto run_family = [&](int f, unsigned take, int back_to) __attribute__((always_inline)) {
        MC_PROF(8 + f);
        MC_PROF_PAIRS(f, take);
        const unsigned h = fget(fheadA, fheadB, f);
        wave_lds_fence();  // queue entries written by fam_push are visible
        uint64_t fp = 0;
        unsigned src = 0;
        if (lane < take) {
            const unsigned e = FL.fq[f][(h + lane) & (FQCAP - 1)];
            const unsigned p = e & 255u;  // (block << 6) | lane of the parent
            const int slot = (int)(e >> 8);
            const uint64_t pidx = wave_idx0 + p;
            const typename S::Summary q = FL.sum[p];
            const BlockRef sp{blk_base, p & 63u};  // NB == 1: the pair's parent is in this wavefront's block
            unsigned st = 0;
            uint64_t fv = 0;
            family_dispatch<S, 0>(f, [&](auto fc) { st = S::template eval_pair<decltype(fc)::value>(prm, q, sp, slot, fv); });
            if (st & ST_ENABLED) {
                ++gen;
                if (track_succ) atomicOr(S::succ_word(FL.sum[p]), 0x80000000u);
                if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                else if (st & ST_ASSERT) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_ASSERT, 0));
                else if (st & ST_SPECERR) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_SPECERR, 0));
                else {
                    if (st & ST_INVARIANT) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_INVARIANT, st >> 8));
                    if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) { fp = fv; src = ((unsigned)slot << 6) | (p & 63u); }
                }
            }
        }
        fset(fheadA, fheadB, f, (h + take) & (FQCAP - 1));
        fset(fcntA, fcntB, f, fget(fcntA, fcntB, f) - take);
        enqueue(fp, src, back_to);
        wave_lds_fence();
    };
    auto run_full = [&](unsigned fullmask, bool drain, int back_to) __attribute__((always_inline)) {
        while (fullmask) {
            const int f = __ffs((int)fullmask) - 1;
            const unsigned c = fget(fcntA, fcntB, f);
            run_family(f, drain ? (c < 64 ? c : 64u) : 64u, back_to);
            if (fget(fcntA, fcntB, f) < (drain ? 1u : 64u)) fullmask &= ~(1u << f);
        }
    };

    for (int blk = 0; blk < NB; ++blk) {
        const uint64_t idx = wave_idx0 + (uint64_t)blk * 64 + lane;
        if (wave_col0 + (uint64_t)blk * 64 >= ncols) break;  // wave-uniform
        const bool active = idx >= lo && idx < hi;
        const unsigned pl = (unsigned)blk * 64 + lane;       // this lane's parent inside the wavefront's NB blocks
        const BlockRef g{blk_base, lane};
        typename S::Guards gd;
        gd.fixed = gd.fixed_hi = 0;
        gd.infl = 0;
        int nm = 0;
        typename S::Local loc;
        MC_PROF(0);
        if (active) {
            S::load_expand(prm, g, loc, gd);  // the whole row in one round trip; guards and per-message codes included
            nm = loc.nm;
            typename S::Summary q;
            S::summarize(loc, q);
            FL.sum[pl] = q;
            const unsigned ps = S::parent_status(prm, loc, g);
            if (ps & ST_INVARIANT) viol = min(viol, viol_key(idx, SLOT_PARENT, VK_INVARIANT, ps >> 8));
        }
        wave_lds_fence();
        const int wnm = (int)wave_max_u32((unsigned)nm);
        // DENSE slots (raft: Restart(i), Timeout(i) — enabled for nearly every parent, half of all successors): evaluated right
        // here, lane = parent, the parent's words in this lane's registers and the server index a compile-time constant; no
        // family queue, no gather.  Only the sparse slots below go through the by-family queues.
        if constexpr (DenseSlots<S>::value > 0) {
            if (!(flags & 64u)) {
                // S::eval_dense(i): the two dense slots of server i together (shared hash terms).  The loops are NOT unrolled: the
                // server index is wave-uniform (scalar registers), and the probe / flush code below exists twice, not 2 * NS times.
                bool mysucc = false;
                MC_PROF(1);
#pragma clang loop unroll(disable)
                for (int i = 0; i < S::DENSE_PAIRS; ++i) {
                    unsigned st2[2] = {0u, 0u};
                    uint64_t f2[2] = {0ull, 0ull};
                    if (active) S::eval_dense(prm, loc, g, i, st2[0], f2[0], st2[1], f2[1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned st = st2[h];
                        const unsigned slot = (unsigned)S::dense_slot(i, h);
                        uint64_t fp = 0;
                        if (st & ST_ENABLED) {
                            ++gen;
                            mysucc = true;
                            if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                            else if (st & ST_ASSERT) viol = min(viol, viol_key(idx, slot, VK_ASSERT, 0));
                            else if (st & ST_SPECERR) viol = min(viol, viol_key(idx, slot, VK_SPECERR, 0));
                            else {
                                if (st & ST_INVARIANT) viol = min(viol, viol_key(idx, slot, VK_INVARIANT, st >> 8));
                                if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) fp = f2[h];
                            }
                        }
                        enqueue(fp, (slot << 6) | pl, 1);
                    }
                }
                S::fixed_clear_dense(gd);
                if (track_succ) {  // deadlock check: one bit per parent; when every parent already has a successor, nothing more to track
                    lane_succ = mysucc;
                    if (!__ballot(active && !mysucc)) track_succ = false;
                }
            }
        }
        // (flag 64 = ablation: load the parents only)
        // IN-FLIGHT MESSAGES, inline, lane = parent (round 3): Receive(m), DuplicateMessage(m), DropMessage(m) need count(m) > 0, and the model
        // bounds the copies in flight (MaxMsgs = 1 in the bench models): a parent has at most MaxMsgs such messages whatever the
        // size of its bag.  Lane = parent evaluates the three actions of ITS j-th in-flight message together — the message word
        // and H(key) shared by the three (Local::cache_hm) — instead of pushing three pairs per bag slot through the family
        // queues: no queue traffic, no gather of the pair's parent, no partially filled batches at the drain; 73 % of the lanes
        // are busy on the bench model (measured share of the wave time before: push loop 10 % + phase B of the message families
        // 15 %, profiles/r03e_phase_profile; step 48.2 -> 41.1 ms on the 102.6 M-state graph, profiles/r03f).  S::eval is the
        // slot-by-slot evaluation the oracle comparisons of tests/_shim run on.
        static_assert(InlineMsgs<S>::value, "a by-family spec evaluates its message actions inline (S::inflight_slots)");
        {
            if (!(flags & 64u)) {
                MC_PROF(6);
                unsigned infl = active ? S::inflight_slots(gd) : 0u;  // bit k: count(message k) > 0, k < GUARD_SLOTS
                auto eval_three = [&](bool on, int k) __attribute__((always_inline)) {
#pragma clang loop unroll(disable)
                    for (int kind = 0; kind < 3; ++kind) {
                        const unsigned slot = (unsigned)(S::FIX + 3 * k + kind);
                        uint64_t fv = 0, fp = 0;
                        bool ev = on;
                        if constexpr (HasGeneratedOnly<S>::value) {
                            if (on && S::message_generated_only(prm, loc, g, (int)slot)) {  // (DuplicateMessage of a full bag: counted, not evaluated)
                                ++gen;
                                lane_succ = true;
                                ev = false;
                            }
                        }
                        const unsigned st = ev ? S::eval(prm, loc, g, (int)slot, fv) : 0u;
                        if (st & ST_ENABLED) {
                            ++gen;
                            lane_succ = true;
                            if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                            else if (st & ST_ASSERT) viol = min(viol, viol_key(idx, slot, VK_ASSERT, 0));
                            else if (st & ST_SPECERR) viol = min(viol, viol_key(idx, slot, VK_SPECERR, 0));
                            else {
                                if (st & ST_INVARIANT) viol = min(viol, viol_key(idx, slot, VK_INVARIANT, st >> 8));
                                if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) fp = fv;
                            }
                        }
                        enqueue(fp, (slot << 6) | pl, 6);
                    }
                };
                // ONE loop, one inlined copy of eval_three (and of the probe code behind enqueue): first every lane's classified
                // in-flight messages (its own k), then the bags beyond the classified slots (the count is read from the row)
#if MC_FOLD_MSG
                for (int kx = S::GUARD_SLOTS;;) {
                    bool on;
                    int k;
                    if (__ballot(infl != 0)) {
                        on = infl != 0;
                        k = on ? __ffs((int)infl) - 1 : 0;
                        infl &= infl - 1;
                    } else if (kx < wnm) {
                        k = kx++;
                        on = k < nm && S::m_count(S::rd_msg(g, k)) > 0;
                        if (!__ballot(on)) continue;
                    } else break;
                    eval_three(on, k);
                }
#else
                while (__ballot(infl != 0)) {
                    const bool on = infl != 0;
                    const int k = on ? __ffs((int)infl) - 1 : 0;
                    infl &= infl - 1;
                    eval_three(on, k);
                }
                for (int k = S::GUARD_SLOTS; k < wnm; ++k) {  // bags beyond the classified slots: the count is read from the row
                    const bool on = k < nm && S::m_count(S::rd_msg(g, k)) > 0;
                    if (__ballot(on)) eval_three(on, k);
                }
#endif
            }
        }
        MC_PROF(5);
        const int lane_inflight = active ? loc.inflight : 0;
        // the fixed slots, and — as the last step of the wavefront's last block — the drain of the family queues (their partially
        // filled batches): ONE loop, so that run_full (family batches, the probe code behind enqueue) is inlined once
#if !MC_FOLD_FIX
        for (int step = 0; step < ((flags & 64u) ? 0 : S::FIX); ++step) {
            const int f = S::fixed_family(step);
            bool en = S::fixed_bit(gd, step);
            if constexpr (HasGeneratedOnly<S>::value) {
                if (en && S::fixed_generated_only(prm, lane_inflight, step)) {
                    ++gen;
                    lane_succ = true;
                    en = false;
                }
            }
            const unsigned long long b = __ballot(en);
            if (b && fam_push(f, b, en, ((unsigned)step << 8) | pl)) run_full(1u << f, false, 5);
        }
    }
    MC_PROF(7);
    {
        unsigned fullmask = 0;
#pragma unroll
        for (int f = 0; f < S::NFAM; ++f) if (fget(fcntA, fcntB, f)) fullmask |= 1u << f;
        run_full(fullmask, true, 7);
    }
    if (false) { int blk = 0; typename S::Guards gd; unsigned pl = 0; int lane_inflight = 0;
#endif
        const int nfix = (flags & 64u) ? 0 : S::FIX;
        for (int step = 0; step <= nfix; ++step) {
            unsigned fullmask = 0;
            const bool drain = step == nfix;
            if (!drain) {
                const int f = S::fixed_family(step);
                bool en = S::fixed_bit(gd, step);
                if constexpr (HasGeneratedOnly<S>::value) {
                    // enabled but never storable, known from the guard and the parent's in-flight count (S::fixed_generated_only):
                    // counted as generated — TLC counts it — and neither queued nor evaluated
                    if (en && S::fixed_generated_only(prm, lane_inflight, step)) {
                        ++gen;
                        lane_succ = true;
                        en = false;
                    }
                }
                const unsigned long long b = __ballot(en);
                if (!(b && fam_push(f, b, en, ((unsigned)step << 8) | pl))) continue;
                fullmask = 1u << f;
            } else {
                if (blk != NB - 1) break;
                MC_PROF(7);
#pragma unroll
                for (int f = 0; f < S::NFAM; ++f) if (fget(fcntA, fcntB, f)) fullmask |= 1u << f;
            }
            run_full(fullmask, drain, drain ? 7 : 5);
        }
    }
    MC_PROF(7);
    if (async_probe) {
        // (nothing left to overlap with: what must look further does so synchronously; the last pass confirms the last
        //  tentative survivors)
        bool first = true;
        while (pend || qn || casmask) {
            if (!pend && qn) probe_issue(qn < 64 ? qn : 64u);
            probe_resolve(first ? (unsigned)QCAP - (qn - pend) : 0u);
            first = false;
        }
    } else if (qn) flush_probe(qn);
    if (inwave) {
        // THE TAIL, by WORKGROUP: the four wavefronts pool their survivors (one per parent on average: 250-400 per workgroup),
        // sort them by action class, take their arena indices with ONE atomicAdd and write them — parent row (this workgroup's
        // own four arena blocks: read a few microseconds ago, L2 / Infinity Cache, not HBM) + patch, lanes = consecutive arena
        // indices, i.e. whole rows of the word-major blocks.  Why sorted: the writer re-evaluates (parent, slot), and with 64
        // survivors in discovery order a wavefront walks EVERY branch of the next-state relation for every batch (~2600
        // instructions per 64 states: k_materialise is bound by instruction issue, 44 ms of the 159 of round 3); sorted, a batch
        // holds one or two classes.  The sort is a counting sort over S::NCLS classes in LDS that is dead by now (the duplicate
        // filters of the wavefronts): no LDS beyond the generation phase's.
        MC_PROF(4);
        constexpr int NCLS = SlotClasses<S>::value;
        if (flags & MC_F_WAVETAIL) {
            // A/B: the tail by WAVEFRONT — no barrier (nobody waits for the slowest wavefront of the workgroup), one atomicAdd per
            // wavefront, the wavefront's own 64-100 survivors sorted by class (a batch still holds about half of the classes)
            MC_PROF(16);
            uint16_t *order = reinterpret_cast<uint16_t *>(FL.filt);  // this wavefront's own filter: dead, its generation is over
            unsigned ccnt[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) ccnt[c] = 0;
            for (unsigned t = 0; t < on; t += 64) {
                const bool valid = t + lane < on;
                const unsigned e_ = valid ? Q.o_ent[(ohead + t + lane) & (OCAP - 1)] : O_DEAD;
                const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
                for (int c = 0; c < NCLS; ++c) ccnt[c] += (unsigned)__popcll(__ballot(cls == c));
            }
            unsigned nlive = 0;  // (tombstones of the split-phase probes are not survivors)
            {
#pragma unroll
                for (int c = 0; c < NCLS; ++c) { const unsigned n = ccnt[c]; ccnt[c] = nlive; nlive += n; }
            }
            wave_lds_fence();
            for (unsigned t = 0; t < on; t += 64) {
                const bool valid = t + lane < on;
                const unsigned k = (ohead + t + lane) & (OCAP - 1);
                const unsigned e_ = valid ? Q.o_ent[k] : O_DEAD;
                const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
                for (int c = 0; c < NCLS; ++c) {
                    const unsigned long long b = __ballot(cls == c);
                    if (cls == c) order[ccnt[c] + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)k;
                    ccnt[c] += (unsigned)__popcll(b);
                }
            }
            wave_lds_fence();
            MC_PROF(17);
            if (nlive) {
                unsigned long long out0 = 0;
                if (lane == 0) out0 = atomicAdd(&ctr->arena_next, (unsigned long long)nlive);
                out0 = __shfl(out0, 0);
                if (out0 + nlive > rt.arena_cap) {
                    err |= DEV_EARENA;
                } else {
                    for (unsigned t = 0; t < nlive; t += 64) {
                        const bool mine = t + lane < nlive;
                        const unsigned k = mine ? order[t + lane] : 0u;
                        const unsigned e = mine ? Q.o_ent[k] : 0u;
                        const uint64_t sfp = mine ? Q.o_fp[k] : 0ull;
                        const uint64_t pidx = wave_idx0 + (e & 63u), oidx = out0 + t + lane;
                        wave_write_survivors<S>(prm, arena, pidx, mine, e >> 6, sfp, rt.arena_w, oidx, FL.sum[e & 63u]);
                        if (mine && rt.parent) { rt.parent[oidx] = (uint32_t)pidx; rt.pslot[oidx] = (uint16_t)(e >> 6); }
                    }
                }
            }
        } else {
        static_assert(NCLS * WAVES <= 64, "one lane per (class, wavefront) in the prefix sum");
        static_assert(WAVES * OCAP * sizeof(uint16_t) <= sizeof(fls[0].filt) && OCAP <= 256, "the sorted order aliases one duplicate filter");
        uint16_t *order = reinterpret_cast<uint16_t *>(fls[0].filt);        // [WAVES * OCAP]: (wavefront << 8) | position in its list
        // [NCLS][WAVES] class counts + the workgroup's first arena index.  Written BEFORE barrier (1), while sibling wavefronts still
        // generate: in LDS of its own, or — split-phase builds — in the writing wavefront's own probe landing area, dead by then
        // (its last probe is resolved); every wavefront's slice lies in ITS landing area: hist of wave w' = probe_land[w'][...]
        static_assert(NCLS * sizeof(unsigned) <= 512, "class counts + first index fit a landing area");
        unsigned *hist_base;             // class c of wavefront ww: hist_base[ww * hist_stride + c]
        unsigned hist_stride;
        unsigned long long *wg_out0;
        if constexpr (ASYNC_BUILD) {
            hist_base = reinterpret_cast<unsigned *>(&probe_land[0][0]);
            hist_stride = (unsigned)((ASYNC_BUILD ? 128 : 2) * 2);  // (32-bit words of one wavefront's landing area)
            wg_out0 = reinterpret_cast<unsigned long long *>(&probe_land[0][64]);
        } else {
            __shared__ unsigned wg_hist_s[WAVES * NCLS];
            __shared__ unsigned long long wg_out0_s;
            hist_base = wg_hist_s;
            hist_stride = NCLS;
            wg_out0 = &wg_out0_s;
        }
        auto hist_at = [&](unsigned c, unsigned ww) -> unsigned & { return hist_base[ww * hist_stride + c]; };
        const unsigned w = threadIdx.x >> 6;
        // a wavefront counts its own survivors per class as soon as IT has finished — in the shadow of the wait for its siblings
        MC_PROF(16);      // (profiling builds: 16 = the counting sort, 4 = waiting at barrier (1), 17 = the writes)
        unsigned ccnt[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) ccnt[c] = 0;
        for (unsigned t = 0; t < on; t += 64) {
            const bool valid = t + lane < on;
            const unsigned e_ = valid ? Q.o_ent[(ohead + t + lane) & (OCAP - 1)] : O_DEAD;
            const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) ccnt[c] += (unsigned)__popcll(__ballot(cls == c));
        }
#pragma unroll
        for (int c = 0; c < NCLS; ++c) if (lane == 0) hist_at((unsigned)c, w) = ccnt[c];
        MC_PROF(4);
        __syncthreads();  // (1) no wavefront of the workgroup generates any more: the filters are free, the lists final, the counts there
        MC_PROF(16);
        const unsigned h = lane < (unsigned)(NCLS * WAVES) ? hist_at(lane / WAVES, lane % WAVES) : 0u;
        unsigned incl = h;
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if ((int)lane >= o) incl += u; }
        const unsigned excl = incl - h, total = __shfl(incl, 63);
        if (w == 0 && lane == 0) *wg_out0 = total ? atomicAdd(&ctr->arena_next, (unsigned long long)total) : 0ull;
#pragma unroll
        for (int c = 0; c < NCLS; ++c) ccnt[c] = __shfl(excl, c * WAVES + (int)w);  // where this wavefront's class-c survivors go
        for (unsigned t = 0; t < on; t += 64) {
            const bool valid = t + lane < on;
            const unsigned k = (ohead + t + lane) & (OCAP - 1);
            const unsigned e_ = valid ? Q.o_ent[k] : O_DEAD;
            const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const unsigned long long b = __ballot(cls == c);
                if (cls == c) order[ccnt[c] + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)((w << 8) | k);
                ccnt[c] += (unsigned)__popcll(b);
            }
        }
        __syncthreads();  // (2) the order and the workgroup's first arena index are visible
        MC_PROF(17);
        const unsigned long long out0 = *wg_out0;
        if (total && out0 + total > rt.arena_cap) {
            err |= DEV_EARENA;
        } else {
            const uint64_t wg_idx0 = base + (uint64_t)blockIdx.x * (64u * WAVES);
            for (unsigned bt = w * 64u; bt < total; bt += 64u * WAVES) {  // batch of 64 sorted survivors; the wavefronts take turns
                const bool mine = bt + lane < total;
                const unsigned ref = mine ? order[bt + lane] : 0u;
                const unsigned e = mine ? wq[ref >> 8].o_ent[ref & 255u] : 0u;
                const uint64_t sfp = mine ? wq[ref >> 8].o_fp[ref & 255u] : 0ull;
                const uint64_t pidx = wg_idx0 + (ref >> 8) * 64u + (e & 63u), oidx = out0 + bt + lane;
                wave_write_survivors<S>(prm, arena, pidx, mine, e >> 6, sfp, rt.arena_w, oidx, fls[ref >> 8].sum[e & 63u]);
                if (mine && rt.parent) { rt.parent[oidx] = (uint32_t)pidx; rt.pslot[oidx] = (uint16_t)(e >> 6); }
            }
        }
        }  // (workgroup tail)
    } else if (on) {
        flush_out(on);
    }
    MC_PROF(7);

    if (flags & MC_F_DEADLOCK) {
        wave_lds_fence();
        for (int blk = 0; blk < NB; ++blk) {
            const uint64_t idx = wave_idx0 + (uint64_t)blk * 64 + lane;
            if (wave_col0 + (uint64_t)blk * 64 >= ncols) break;
            if (idx >= lo && idx < hi && !lane_succ && !(*S::succ_word(FL.sum[blk * 64 + lane]) >> 31)) viol = min(viol, viol_key(idx, SLOT_NONE, VK_DEADLOCK, 0));
        }
    }
    const unsigned gsum = wave_sum_u32(gen);
    const unsigned long long vmin = wave_min_u64(viol);
    const unsigned eor = wave_or_u32(err);
    if (lane == 0) {
        if (gsum) atomicAdd(&ctr->generated[shard].v, (unsigned long long)gsum);
        (void)probes;  // `cells` counts the in-model successors, i.e. the seen-set look-ups the algorithm asks for (before the filter)
        if (cands) atomicAdd(&ctr->cells[shard].v, (unsigned long long)cands);
        if (vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
        if (eor) atomicOr(&ctr->error, eor);
    }
    MC_PROF_END;
}

// specs with a copy + patch writer (S::PATCH_WORDS, eval_pair_delta, write_patched) run the one-kernel form

// specs that define action families (S::NFAM) are expanded by family, the others slot by slot
template <class S, class = void>
struct WantsSlices : std::false_type {};
template <class S>
struct WantsSlices<S, decltype((void)S::SLICE_SLOTS)> : std::true_type {};
template <class S, class = void>
struct UsesFamilies : std::false_type {};
template <class S>
struct UsesFamilies<S, decltype((void)S::NFAM)> : std::true_type {};

// (several arena blocks per wavefront, NB = 2 / 4, and a 5-waves-per-SIMD register budget were measured slower in round 2 —
// DESIGN.md §5 — and are no longer compiled)
// after a slot-sliced expand: the parents none of whose slices produced a successor
static __global__ void __launch_bounds__(256)
k_deadlock_slices(const uint16_t *__restrict__ succ, uint64_t lo, uint64_t hi, uint64_t ncols, const LevelCtl *lc, DevCounters *ctr) {
    if (lc) {
        if (lc->stop) return;
        lo = lc->lo;
        hi = lc->hi;
        ncols = ((hi - (lo & ~63ull)) + 63) & ~63ull;
    }
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    const uint64_t idx = (lo & ~63ull) + col;
    unsigned long long viol = ~0ull;
    if (idx >= lo && idx < hi && !succ[col]) viol = viol_key(idx, SLOT_NONE, VK_DEADLOCK, 0);
    const unsigned long long vmin = wave_min_u64(viol);
    if ((threadIdx.x & 63) == 0 && vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
}
template <class S, bool ROUTE, class... A>
static void launch_expand(bool by_family, unsigned flags, uint64_t ncols, hipStream_t stream, unsigned slices, A... args) {
    if constexpr (UsesFamilies<S>::value) {
        if (by_family) {
            // MC_F_OCC3 (A/B): the register budget of 3 wavefronts per SIMD (no spills) instead of 4 (a dozen spilled VGPRs)
            constexpr unsigned WG = 64u * MC_EXPAND_WAVES;   // columns (parents) per workgroup
            if (flags & MC_F_OCC3)
                hipLaunchKernelGGL((k_expand_family<S, ROUTE, 1, 3>), dim3((unsigned)((ncols + WG - 1) / WG)), dim3(WG), 0, stream, args...);
            else
                hipLaunchKernelGGL((k_expand_family<S, ROUTE, 1>), dim3((unsigned)((ncols + WG - 1) / WG)), dim3(WG), 0, stream, args...);
            return;
        }
    }
    hipLaunchKernelGGL((k_expand_insert<S, ROUTE>), dim3((unsigned)((ncols + 255) / 256), slices ? slices : 1u), dim3(256), 0, stream, args...);
}

// ------------------------------------------------------------------------------------- materialise
template <class S>
__global__ void __launch_bounds__(256)
k_materialise(typename S::Params prm, uint64_t *arena, uint64_t chunk_base, const uint32_t *__restrict__ newlist, uint64_t seg_cap,
              uint64_t arena_cap, uint32_t *__restrict__ parent, uint16_t *__restrict__ pslot, DevCounters *ctr, unsigned parity,
              const LevelCtl *lc, const uint64_t *__restrict__ newfp) {
    if (lc) {
        if (lc->stop) return;
        chunk_base = lc->lo & ~63ull;
    }
    const unsigned sh = blockIdx.y;  // new-list segment
    const uint64_t n = ctr->n_new[parity * NSHARD + sh].v;
    const bool atomic_alloc = ctr->atomic_alloc != 0;  // fused runs: every writer takes its indices from arena_next itself
    uint64_t out0 = 0;
    if (!atomic_alloc) {
        out0 = ctr->arena_next;
        for (unsigned t = 0; t < sh; t++) out0 += ctr->n_new[parity * NSHARD + t].v;
    }
    const uint32_t *__restrict__ seg = newlist + (uint64_t)(parity * NSHARD + sh) * seg_cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t src = seg[j];
        const uint64_t pidx = chunk_base + (src & 0xffffffu);
        const int slot = (int)(src >> 24);
        uint64_t oidx = out0 + j;
        if (atomic_alloc) {  // (the lanes of a wavefront hold consecutive j: the active ones are a prefix)
            const unsigned long long act = __ballot(true);
            unsigned long long w0 = 0;
            if ((threadIdx.x & 63u) == 0) w0 = atomicAdd(&ctr->arena_next, (unsigned long long)__popcll(act));
            oidx = __shfl(w0, 0) + (threadIdx.x & 63u);
        }
        if (oidx >= arena_cap) { atomicOr(&ctr->error, DEV_EARENA); continue; }
        if constexpr (HasKnownFp<S>::value) {  // the expand kernel hands the successor's fingerprint over: no second delta_fp
            if (newfp) S::apply_known_fp(prm, arena_cref(arena, pidx, S::words(prm)), slot, newfp[(uint64_t)(parity * NSHARD + sh) * seg_cap + j], arena_ref(arena, oidx, S::words(prm)));
            else S::apply(prm, arena_cref(arena, pidx, S::words(prm)), slot, arena_ref(arena, oidx, S::words(prm)));
        } else {
            S::apply(prm, arena_cref(arena, pidx, S::words(prm)), slot, arena_ref(arena, oidx, S::words(prm)));
        }
        if (parent) { parent[oidx] = (uint32_t)pidx; pslot[oidx] = (uint16_t)slot; }
    }
}
template <class S>
__global__ void __launch_bounds__(256)
k_init_materialise(typename S::Params prm, uint64_t *arena, uint64_t first, const uint64_t *__restrict__ tmp,
                   const uint32_t *__restrict__ newlist, uint64_t arena_cap, uint32_t *__restrict__ parent,
                   uint16_t *__restrict__ pslot, DevCounters *ctr) {
    const uint64_t n = ctr->n_new[0].v, out0 = ctr->arena_next;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const int W = S::words(prm);
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint64_t col = newlist[j] & 0xffffffu;
        const uint64_t oidx = out0 + j;
        if (oidx >= arena_cap) { atomicOr(&ctr->error, DEV_EARENA); continue; }
        const WordRef o = arena_ref(arena, oidx, W);
        for (int w = 0; w < W; w++) o.set(w, tmp[col * (uint64_t)W + w]);
        if (parent) { parent[oidx] = 0xffffffffu; pslot[oidx] = (uint16_t)((first + col) & 0xffffu); }
    }
}
// arena (blocked, word-major) -> plain records, for read-back and for the exchange buffers
static __global__ void __launch_bounds__(256)
k_gather_states(const uint64_t *__restrict__ arena, int words, uint64_t first, uint64_t count, uint64_t *__restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * (uint64_t)words) return;
    const uint64_t j = t / (uint64_t)words, w = t % (uint64_t)words, idx = first + j;
    out[t] = arena[((idx >> 6) * (uint64_t)words + w) * 64 + (idx & 63)];
}

// ------------------------------------------------------------------------------------- sharded step kernels
// sub-buckets [owner][shard] -> one contiguous range per owner (order inside an owner: by shard)
static __global__ void __launch_bounds__(256)
k_compact_buckets(RouteArgs rt, uint64_t *__restrict__ send_fp, uint32_t *__restrict__ pend_src) {
    const unsigned bucket = blockIdx.y;  // owner * NSHARD + shard
    const uint64_t n = rt.cursors[bucket].v < rt.subcap ? rt.cursors[bucket].v : rt.subcap;
    uint64_t off = 0;
    for (unsigned b = 0; b < bucket; ++b) off += rt.cursors[b].v < rt.subcap ? rt.cursors[b].v : rt.subcap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        send_fp[off + j] = rt.rt_fp[(uint64_t)bucket * rt.subcap + j];
        pend_src[off + j] = rt.rt_src[(uint64_t)bucket * rt.subcap + j];
    }
}
// FIXED-CAPACITY exchange (no size message, no host in the round): owner t's candidates go to send_fp[t * cap + 1 ...] and their
// NUMBER into send_fp[t * cap] — in band, so the receiver learns it from the bucket itself; pend_src uses the same positions.
// A bucket that does not fit raises DEV_EROUTE (reported at the end of the level: raise the caller's fan-out allowance).
static __global__ void __launch_bounds__(256)
k_compact_packed(RouteArgs rt, uint64_t cap, uint64_t *__restrict__ send_fp, uint32_t *__restrict__ pend_src, DevCounters *ctr) {
    const unsigned bucket = blockIdx.y, owner = bucket / NSHARD;  // bucket = owner * NSHARD + shard
    const uint64_t n = rt.cursors[bucket].v < rt.subcap ? rt.cursors[bucket].v : rt.subcap;
    uint64_t off = 1, total = 0;
    bool over = false;
    for (unsigned b = owner * NSHARD; b < (owner + 1) * NSHARD; ++b) {
        const uint64_t c = rt.cursors[b].v;
        over |= c > rt.subcap;
        const uint64_t cc = c < rt.subcap ? c : rt.subcap;
        if (b < bucket) off += cc;
        total += cc;
    }
    over |= total + 1 > cap;
    if (blockIdx.x == 0 && threadIdx.x == 0 && bucket == owner * NSHARD) {
        send_fp[(uint64_t)owner * cap] = over ? 0ull : total;
        if (over) atomicOr(&ctr->error, DEV_EROUTE);
    }
    if (over) return;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, base = (uint64_t)owner * cap + off;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        send_fp[base + j] = rt.rt_fp[(uint64_t)bucket * rt.subcap + j];
        pend_src[base + j] = rt.rt_src[(uint64_t)bucket * rt.subcap + j];
    }
}
// owner side of the fixed-capacity exchange: bucket s of recv_fp came from rank s; answers keep the positions (0 outside a bucket's
// count, so the sender can scan the whole buffer without knowing the counts)
static __global__ void __launch_bounds__(256)
k_probe_packed(const uint64_t *__restrict__ fps, uint64_t cap, unsigned nranks, uint64_t *table, uint64_t mask, uint8_t *__restrict__ answers,
               DevCounters *ctr) {
    // Workgroup b probes 256 consecutive entries of the bucket of source rank b % nranks: the sources are walked INTERLEAVED.
    // A fingerprint that several ranks generated in the same round is "new" for whichever candidate reaches the table first,
    // and its state then lives on that rank: with the buckets walked one after the other (source 0 first) the lower ranks won
    // those ties systematically and their frontiers grew level after level (8 ranks, 10^8 states: 17.2 M on rank 0 against
    // 11.1 M on rank 5; measured, profiles/r03a) — every other level became a rebalancing level.
    const unsigned s = blockIdx.x % nranks;
    const uint64_t j = (uint64_t)(blockIdx.x / nranks) * blockDim.x + threadIdx.x;
    unsigned err = 0;
    if (j < cap) {
        const uint64_t s0 = (uint64_t)s * cap, i = s0 + j;
        const uint64_t n = fps[s0] < cap ? fps[s0] : 0;  // (a count that cannot be: an overflowed bucket, already reported by its sender)
        bool is_new = false;
        if (j >= 1 && j <= n) is_new = seen_insert(table, mask, fps[i], err);
        answers[i] = is_new ? 1 : 0;
    }
    if (wave_or_u32(err) && (threadIdx.x & 63) == 0) atomicOr(&ctr->error, DEV_ETABLE);
}
// owner side: insert received fingerprints, answer 1 = new
static __global__ void __launch_bounds__(256)
k_probe(const uint64_t *__restrict__ fps, uint64_t n, uint64_t *table, uint64_t mask, uint8_t *__restrict__ answers, DevCounters *ctr) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned err = 0;
    bool is_new = false;
    if (i < n) {
        is_new = seen_insert(table, mask, fps[i], err);
        answers[i] = is_new ? 1 : 0;
    }
    const unsigned long long b = __ballot(is_new);
    (void)b;
    if (wave_or_u32(err) && (threadIdx.x & 63) == 0) atomicOr(&ctr->error, DEV_ETABLE);
}
// sender side: per-owner count of positive answers (owner ranges given by off[0..nranks])
struct OwnerOffsets { uint64_t off[9]; };
__device__ __forceinline__ unsigned owner_of_index(const OwnerOffsets &o, unsigned nranks, uint64_t i) {
    unsigned t = 0;
    while (t + 1 < nranks && i >= o.off[t + 1]) ++t;
    return t;
}
// sender side, step 1: compact the sources of the positively answered candidates per owner.
// incl[] = inclusive prefix sum of the answers (hipcub::DeviceScan), so positions need no atomics;
// start[t] = number of positive answers before owner t's range.
struct AnswerCast {
    __host__ __device__ uint32_t operator()(const uint8_t &a) const { return a ? 1u : 0u; }
};
static __global__ void __launch_bounds__(256)
k_gather_range_ends(const uint32_t *__restrict__ incl, OwnerOffsets offs, unsigned nranks, unsigned long long *__restrict__ ends) {
    const unsigned t = threadIdx.x;
    if (t < nranks) ends[t] = offs.off[t + 1] ? incl[offs.off[t + 1] - 1] : 0;
}
static __global__ void __launch_bounds__(256)
k_compact_new(const uint8_t *__restrict__ answers, const uint32_t *__restrict__ incl, const uint32_t *__restrict__ pend_src,
              uint64_t total, OwnerOffsets offs, OwnerOffsets start, unsigned nranks, uint32_t *__restrict__ new_src) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || !answers[i]) return;
    const unsigned t = owner_of_index(offs, nranks, i);
    // new states of owner t are a subset of its pending range, so they fit at the range's start
    new_src[offs.off[t] + (incl[i] - 1 - start.off[t])] = pend_src[i];
}
// Exchange format of full states: per owner a whole number of 64-state BLOCKS, word-major inside a
// block exactly like the arena, so that both the sender's writes and the receiver's reads are
// coalesced.  blk_off[t] = first block of owner t in the send buffer, cnt[t] = its states.
struct BlockPlan { uint64_t blk_off[9]; uint64_t cnt[8]; };
template <class S>
__global__ void __launch_bounds__(256)
k_send_materialise(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t chunk_base, const uint32_t *__restrict__ new_src,
                   OwnerOffsets offs, BlockPlan plan, unsigned nranks, uint64_t *__restrict__ send_states) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // global lane over all blocks
    const uint64_t blk = g >> 6;
    if (blk >= plan.blk_off[nranks]) return;
    unsigned t = 0;
    while (t + 1 < nranks && blk >= plan.blk_off[t + 1]) ++t;
    const uint64_t j = ((blk - plan.blk_off[t]) << 6) + (g & 63);  // index inside owner t's bucket
    const int W = S::words(prm);
    const WordRef out{send_states + blk * (uint64_t)W * 64 + (g & 63), 64};
    if (j < plan.cnt[t]) {
        const uint32_t src = new_src[offs.off[t] + j];
        S::apply(prm, arena_cref(arena, chunk_base + (src & 0xffffffu), W), (int)(src >> 24), out);
    } else {
        for (int w = 0; w < W; w++) out.set(w, 0);  // padding lanes of the owner's last block
    }
}
// "stay" mode of the sharded engine: new states are materialised on the rank that generated them
// (only fingerprints travelled); `list` holds the compacted sources of the positive answers
template <class S>
__global__ void __launch_bounds__(256)
k_materialise_list(typename S::Params prm, uint64_t *arena, uint64_t chunk_base, const uint32_t *__restrict__ list,
                   const uint32_t *__restrict__ n_dev, uint64_t arena_cap, uint32_t *__restrict__ parent, uint16_t *__restrict__ pslot,
                   DevCounters *ctr) {
    // n (= last element of the inclusive scan) and the output base stay on the device: no host round trip per round
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = *n_dev, out0 = ctr->arena_next;
    if (j >= n) return;
    const uint32_t src = list[j];
    const uint64_t pidx = chunk_base + (src & 0xffffffu), oidx = out0 + j;
    if (oidx >= arena_cap) { atomicOr(&ctr->error, DEV_EARENA); return; }
    const int W = S::words(prm);
    S::apply(prm, arena_cref(arena, pidx, W), (int)(src >> 24), arena_ref(arena, oidx, W));
    if (parent) { parent[oidx] = (uint32_t)pidx; pslot[oidx] = (uint16_t)(src >> 24); }
}
// owner side: append the `n` states of one received bucket (blocked layout) to the arena
static __global__ void __launch_bounds__(256)
k_ingest(uint64_t *arena, int words, const uint64_t *__restrict__ recv_blocks, uint64_t n, uint64_t arena_cap,
         uint32_t *__restrict__ parent, DevCounters *ctr) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t oidx = ctr->arena_next + j;
    if (oidx >= arena_cap) { atomicOr(&ctr->error, DEV_EARENA); return; }
    const WordRef o = arena_ref(arena, oidx, words);
    const uint64_t *src = recv_blocks + (j >> 6) * (uint64_t)words * 64 + (j & 63);
    for (int w = 0; w < words; w++) o.set(w, src[(uint64_t)w * 64]);
    if (parent) parent[oidx] = 0xfffffffeu;  // produced on another rank: no local parent
}
// Counterexamples across ranks: a state that MOVES to its owner takes (index of its parent on the sending rank, slot) with it.
// sender side: per moved state, in the order of the state blocks (owner by owner), parent = chunk_base + column
static __global__ void __launch_bounds__(256)
k_send_parents(const uint32_t *__restrict__ new_src, uint64_t chunk_base, OwnerOffsets offs, BlockPlan plan, unsigned nranks,
               uint64_t *__restrict__ send_parents) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t before = 0;
    for (unsigned t = 0; t < nranks; ++t) {
        if (g < before + plan.cnt[t]) {
            const uint32_t src = new_src[offs.off[t] + (g - before)];
            send_parents[g] = ((chunk_base + (src & 0xffffffu)) << 16) | (uint64_t)(src >> 24);
            return;
        }
        before += plan.cnt[t];
    }
}
// owner side: the n states ingested last (arena_next - n ...) get their remote parent
static __global__ void __launch_bounds__(256)
k_ingest_parents(const uint64_t *__restrict__ recv_parents, uint64_t n, unsigned src_rank, uint32_t *__restrict__ parent,
                 uint16_t *__restrict__ pslot, uint8_t *__restrict__ prank, const DevCounters *ctr) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t oidx = ctr->arena_next - n + j;
    parent[oidx] = (uint32_t)(recv_parents[j] >> 16);
    pslot[oidx] = (uint16_t)(recv_parents[j] & 0xffffu);
    prank[oidx] = (uint8_t)src_rank;
}
// runs alone on its stream after the kernel that appended: arena_next += n (n on the device when n_dev != null)
static __global__ void k_bump_arena_next(DevCounters *ctr, const uint32_t *n_dev, unsigned long long n, unsigned long long arena_cap) {
    const unsigned long long v = ctr->arena_next + (n_dev ? (unsigned long long)*n_dev : n);
    if (v > arena_cap) atomicOr(&ctr->error, DEV_EARENA);
    else ctr->arena_next = v;
}
static __global__ void k_commit(DevCounters *ctr, unsigned parity) {
    unsigned long long n = 0;
    for (int t = 0; t < NSHARD; t++) { n += ctr->n_new[parity * NSHARD + t].v; ctr->n_new[parity * NSHARD + t].v = 0; }
    ctr->via_list += n;
    if (!ctr->atomic_alloc) ctr->arena_next += n;  // (atomic_alloc: k_materialise took the indices itself)
    ctr->max_slots = 0;
}

// replicated prefix -> sharded continuation: rank r keeps the states of the last replicated level whose fingerprint it
// owns.  (Not "every nranks-th state": the ORDER of a level in the arena differs from rank to rank, its SET does not.)
template <class S>
__global__ void __launch_bounds__(256)
k_take_owned(typename S::Params prm, uint64_t *arena, uint64_t lo, uint64_t hi, unsigned rank, unsigned nranks, uint64_t dst0,
             uint64_t arena_cap, uint32_t *__restrict__ parent, uint16_t *__restrict__ pslot, DevCounters *ctr) {
    const uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const int W = S::words(prm);
    const CWordRef in = arena_cref(arena, i, W);
    if (fp_owner(S::fp_of(prm, in), nranks) != rank) return;
    const uint64_t dst = dst0 + atomicAdd(&ctr->n_new[0].v, 1ull);
    if (dst >= arena_cap) { atomicOr(&ctr->error, DEV_EARENA); return; }
    const WordRef out = arena_ref(arena, dst, W);
    for (int w = 0; w < W; w++) out.set(w, in.get(w));
    if (parent) { parent[dst] = (uint32_t)i; pslot[dst] = (uint16_t)SLOT_COPY; }
}
static __global__ void k_set_alloc_mode(DevCounters *ctr, unsigned atomic_alloc) { ctr->atomic_alloc = atomic_alloc; }
static __global__ void k_after_prefix(DevCounters *ctr, unsigned long long dst0, int zero_counts) {
    ctr->atomic_alloc = 0;  // the sharded rounds append in stream order (k_commit / k_bump_arena_next)
    ctr->arena_next = dst0 + ctr->n_new[0].v;
    ctr->n_new[0].v = 0;
    if (zero_counts) for (int t = 0; t < NSHARD; t++) { ctr->generated[t].v = 0; ctr->cells[t].v = 0; }
}

// Specs that check their invariants when a state is EXPANDED (S::CHECK_ON_EXPAND: the SI models — one evaluation per stored
// state instead of one per generated successor) have not yet looked at the level a run stops on.  TLC checks a state when it
// is generated, so before a run that leaves an unexpanded frontier reports "budget", that frontier is checked here.
template <class S>
__global__ void __launch_bounds__(256)
k_check_frontier(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi, DevCounters *ctr) {
    const uint64_t idx = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long viol = ~0ull;
    if (idx < hi) {
        const CWordRef g = arena_cref(arena, idx, S::words(prm));
        typename S::Local loc;
        S::load(prm, g, loc);
        const unsigned ps = S::parent_status(prm, loc, g);
        if (ps & ST_INVARIANT) viol = viol_key(idx, SLOT_PARENT, VK_INVARIANT, ps >> 8);
    }
    const unsigned long long vmin = wave_min_u64(viol);
    if ((threadIdx.x & 63) == 0 && vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
}
template <class S, class = void>
struct ChecksOnExpand : std::false_type {};
template <class S>
struct ChecksOnExpand<S, decltype((void)S::CHECK_ON_EXPAND)> : std::true_type {};

// closes a batched level on the device: advances [lo, hi), records the fill level, decides whether the next one may run
static __global__ void k_end_level(DevCounters *ctr, LevelCtl *lc) {
    if (lc->stop) return;
    {  // k_commit of new-list parity 0, folded in (one launch less per level)
        unsigned long long n = 0;
        for (int t = 0; t < NSHARD; t++) { n += ctr->n_new[t].v; ctr->n_new[t].v = 0; }
        ctr->via_list += n;
        if (!ctr->atomic_alloc) ctr->arena_next += n;
        ctr->max_slots = 0;
    }
    const unsigned long long hi_new = ctr->arena_next;
    lc->level_hi[lc->nlev++] = hi_new;
    lc->lo = lc->hi;
    lc->hi = hi_new;
    if (hi_new == lc->lo || ctr->viol_key != ~0ull || ctr->error) lc->stop = 1;
    else if (lc->max_distinct && hi_new >= lc->max_distinct) lc->stop = 1;
    else if (lc->levels_left && --lc->levels_left == 0) lc->stop = 1;
    else if (hi_new - lc->lo > lc->max_states) lc->stop = 2;
}

// ------------------------------------------------------------------------------------- host side
struct EngineBase {
    void *owned_device_blob = nullptr;  // program image of a compiled PlusCal spec (spec_vm.h)
    uint64_t program_hash = 0;          // ... and a hash of that image + its entry points: what a checkpoint of it is matched by
    mc_progress_fn progress_fn = nullptr;  // mc_engine_set_progress
    void *progress_user = nullptr;
    double progress_interval = 1.0;
    std::chrono::steady_clock::time_point progress_last;
    void report_progress(uint32_t levels, uint64_t generated, uint64_t distinct, uint64_t queue) {
        if (!progress_fn) return;
        const auto now = std::chrono::steady_clock::now();
        if (std::chrono::duration<double>(now - progress_last).count() < progress_interval) return;
        progress_last = now;
        progress_fn(progress_user, levels, generated, distinct, queue);
    }
    virtual ~EngineBase() { if (owned_device_blob) hipFree(owned_device_blob); }
    virtual int run(mc_result *out) = 0;
    virtual int step(uint32_t levels, mc_result *out) = 0;
    virtual int trace(uint8_t *states_out, int32_t *actions_out, size_t *n_inout) = 0;
    virtual int kernel_stats(mc_kernel_stats *out) = 0;
    virtual int read_states(uint64_t first, uint64_t count, uint8_t *out) = 0;
    virtual int debug_reexpand(unsigned extra_flags, double *ms) = 0;
    virtual int debug_phases(uint64_t *out48, int reset) = 0;
    virtual int checkpoint(const char *path) = 0;
    virtual int restore(const char *path) = 0;
    virtual int shard_begin() = 0;
    virtual int shard_begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) = 0;
    virtual int shard_level_size(uint64_t *n) = 0;
    virtual int shard_set_stream(void *hip_stream, int enable) = 0;
    virtual int shard_expand_launch(unsigned slot, uint64_t first, uint64_t count, uint64_t send_cap) = 0;
    virtual int shard_expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int shard_probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) = 0;
    virtual int shard_expand_pack(unsigned slot, uint64_t *send_fp, uint64_t cap) = 0;
    virtual int shard_probe_pack(const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) = 0;
    virtual int shard_keep_pack(unsigned slot, const uint8_t *answers_back, uint64_t cap) = 0;
    virtual int shard_wait_keep(unsigned slot) = 0;
    virtual int shard_materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) = 0;
    virtual int shard_ingest(const uint8_t *recv_states, uint64_t n) = 0;
    virtual int shard_keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) = 0;
    virtual int shard_end_level(uint64_t *new_local) = 0;
    virtual int shard_counters(uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) = 0;
    virtual int shard_check_frontier() = 0;
    virtual int shard_materialise_parents(unsigned slot, uint64_t *send_parents) = 0;
    virtual int shard_ingest_parents(const uint64_t *recv_parents, uint64_t n, unsigned src_rank) = 0;
    virtual int shard_violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) = 0;
    virtual int shard_fetch(uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) = 0;
    virtual int shard_info(void **main_stream, uint64_t *chunk_states, int32_t *traced) = 0;
    virtual int shard_note_levels(const uint64_t *levels, uint32_t n, int32_t verdict) = 0;
    virtual int shard_resume(uint64_t *levels_out, uint32_t *nlevels) = 0;
    virtual int shard_checkpoint(const char *path) = 0;
    virtual int shard_restore(const char *path) = 0;
    virtual size_t state_bytes() const = 0;
};

static uint64_t round_pow2(uint64_t v) {
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

struct KTimer {
    std::vector<hipEvent_t> pool;
    struct Pending { int which; size_t e0, e1; uint64_t units; };
    std::vector<Pending> pending;
    size_t used = 0;
    bool enabled = false;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; hipEventCreate(&e); pool.push_back(e); }
        return pool[used++];
    }
    void resolve(mc_kernel_stat *stats /*[3]*/) {
        for (auto &p : pending) {
            float ms = 0;
            hipEventElapsedTime(&ms, pool[p.e0], pool[p.e1]);
            stats[p.which].launches++;
            stats[p.which].ms_total += ms;
            stats[p.which].units += p.units;
        }
        pending.clear();
        used = 0;
    }
    ~KTimer() { for (auto e : pool) hipEventDestroy(e); }
};

template <class S>
struct Engine : EngineBase {
    using Params = typename S::Params;
    int W = 1;  // 64-bit words per packed state
    unsigned max_slots = 1;
    Params prm;
    mc_spec_desc desc;
    mc_config cfg;
    hipStream_t stream = nullptr, stream2 = nullptr;  // expand+insert on `stream`, materialise on `stream2`
    hipEvent_t ev_e[2] = {nullptr, nullptr}, ev_m[2] = {nullptr, nullptr};
    uint64_t *d_arena = nullptr, *d_table = nullptr, *d_cand = nullptr;
    uint32_t *d_newlist = nullptr;
    uint64_t *d_newfp = nullptr;  // fingerprints of the new-list entries (specs with apply_known_fp)
    uint16_t *d_nsl = nullptr, *d_pslot = nullptr;
    uint64_t *d_inittmp = nullptr;
    uint32_t *d_parent = nullptr;
    uint8_t *d_prank = nullptr;  // sharded runs with MC_F_TRACE: rank a state's parent lives on (0xff = this rank)
    DevCounters *d_ctr = nullptr, *h_ctr = nullptr;
    LevelCtl *d_lc = nullptr, *h_lc = nullptr;
    uint64_t table_cap = 0, arena_cap = 0, chunk = 0, row_stride = 0, seg_cap = 0;
    bool seen_sparse = false;
    // slot slices of a generic-kernel launch (k_expand_insert): as many as it takes to give the device a few thousand wavefronts,
    // for specs whose slots all go through the loop (no unrolled prefix) and that have enough of them; TLAMC_NOSLICE=1 = A/B
    unsigned slices_for(uint64_t ncols, bool blind) const {
        // (a lowering asks for it with SLICE_SLOTS: worth it when one slot evaluation is expensive — the Paxos family's witness
        // enumeration, the bytecode interpreter; for atomic_add's two-instruction slots the slices only repeat the parent loads)
        if (!WantsSlices<S>::value || S::FIX_SLOTS != 0 || UsesFamilies<S>::value || use_matrix || max_slots < 16 || no_slices) return 1;
        const uint64_t waves = (ncols + 63) / 64;
        uint64_t sg = blind ? 32 : (16384 + waves - 1) / waves;  // a blind level launches its CAPACITY: the frontier itself is smaller
        if (sg > 32) sg = 32;
        if (sg > max_slots / 4) sg = max_slots / 4;
        return sg < 1 ? 1u : (unsigned)sg;
    }
    bool no_slices = getenv("TLAMC_NOSLICE") != nullptr;
    uint64_t seen_arg() const { return seen_sparse ? ((table_cap / MC_SPARSE_SLOTS) | SEEN_SPARSE) : table_cap / 8; }
    KTimer timer;
    mc_kernel_stat kstat[3];
    // counterexample of the last run
    bool have_viol = false;
    unsigned long long last_viol = ~0ull;
    std::vector<uint64_t> level_start;

    int alloc() {
        W = S::words(prm);
        max_slots = (unsigned)S::max_slots(prm);
        use_matrix = (cfg.flags & MC_F_MATRIX) != 0;
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        {   // A/B knob (measured in DESIGN.md section 4): TLAMC_PRIO=1 gives the materialise stream the highest priority, =2 the lowest
            const char *pe = getenv("TLAMC_PRIO");
            int lo_p = 0, hi_p = 0;
            hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);  // lo_p = least, hi_p = greatest priority (numerically smaller)
            if (pe && *pe == '1') HIP_TRY(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, hi_p));
            else if (pe && *pe == '2') HIP_TRY(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, lo_p));
            else HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        }
        for (int i = 0; i < 2; i++) { HIP_TRY(hipEventCreateWithFlags(&ev_e[i], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ev_m[i], hipEventDisableTiming)); }
        table_cap = cfg.table_capacity ? cfg.table_capacity : (1ull << 24);
        table_cap = (table_cap + 63) / 64 * 64;  // whole 8-slot buckets; any size (seen_insert), at most 2^32 buckets
        if (table_cap / 8 > 0xffffffffull) { set_error("table_capacity: at most 2^35 - 8 slots per device"); return MC_EBADCFG; }
        arena_cap = cfg.arena_capacity ? cfg.arena_capacity : (1ull << 22);
        arena_cap = (arena_cap + 63) & ~63ull;
        if (arena_cap >= (1ull << 32) - 1) { set_error("arena_capacity must be < 2^32 states"); return MC_EBADCFG; }
        // a table that can never be more than a third full is probed 32 bytes at a time (seen_insert)
        seen_sparse = table_cap >= 3 * arena_cap && table_cap / MC_SPARSE_SLOTS <= 0xffffffffull && !getenv("TLAMC_DENSE_TABLE");
        chunk = cfg.chunk_states ? cfg.chunk_states : (1ull << 18);
        chunk = (chunk + 255) & ~255ull;
        if (chunk > (1ull << 23)) chunk = 1ull << 23;  // a column index must fit 24 bits
        if (max_slots > 255) { set_error("spec has more than 255 action slots per state"); return MC_EBADCFG; }
        row_stride = chunk + 256;
        HIP_TRY(hipMalloc(&d_arena, arena_cap * W * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_table, table_cap * sizeof(uint64_t)));
        HIP_TRY(hipMalloc(&d_cand, (size_t)(use_matrix ? max_slots : 1) * row_stride * sizeof(uint64_t)));
        seg_cap = (row_stride / NSHARD + 256) * max_slots;
        HIP_TRY(hipMalloc(&d_newlist, (size_t)2 * NSHARD * seg_cap * sizeof(uint32_t)));
        if (HasKnownFp<S>::value && !use_matrix) HIP_TRY(hipMalloc(&d_newfp, (size_t)2 * NSHARD * seg_cap * sizeof(uint64_t)));
        {
            const uint64_t ni = S::num_init(prm);
            HIP_TRY(hipMalloc(&d_inittmp, (size_t)(ni < chunk ? ni : chunk) * W * sizeof(uint64_t)));
        }
        HIP_TRY(hipMalloc(&d_nsl, row_stride * sizeof(uint16_t)));
        if (cfg.flags & MC_F_TRACE) {
            HIP_TRY(hipMalloc(&d_parent, arena_cap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&d_pslot, arena_cap * sizeof(uint16_t)));
            if (cfg.shard_count > 1) {
                HIP_TRY(hipMalloc(&d_prank, arena_cap));
                HIP_TRY(hipMemset(d_prank, 0xff, arena_cap));
            }
        }
        HIP_TRY(hipMalloc(&d_ctr, sizeof(DevCounters)));
        HIP_TRY(hipHostMalloc(&h_ctr, sizeof(DevCounters)));
        HIP_TRY(hipMalloc(&d_lc, sizeof(LevelCtl)));
        HIP_TRY(hipHostMalloc(&h_lc, sizeof(LevelCtl)));
        timer.enabled = (cfg.flags & MC_F_TIMING) != 0;
        return MC_OK;
    }
    ~Engine() override {
        if (d_arena) hipFree(d_arena);
        if (d_table) hipFree(d_table);
        if (d_cand) hipFree(d_cand);
        if (d_newlist) hipFree(d_newlist);
        if (d_newfp) hipFree(d_newfp);
        if (d_nsl) hipFree(d_nsl);
        if (d_inittmp) hipFree(d_inittmp);
        for (auto &q : sl) {
            if (q.rt_fp) { hipFree(q.rt_fp); hipFree(q.rt_src); hipFree(q.pend_src); }
            if (q.rt_cur) hipFree(q.rt_cur);
        }
        if (d_new_src) { hipFree(d_new_src); hipFree(d_incl); }
        if (d_ends) { hipFree(d_ends); hipHostFree(h_ends); hipHostFree(h_cur); }
        for (auto &ev : ev_slot) if (ev) hipEventDestroy(ev);
        for (auto &ev : ev_keep) if (ev) hipEventDestroy(ev);
        if (ev_ans) hipEventDestroy(ev_ans);
        if (ev_append) hipEventDestroy(ev_append);
        if (d_scan_tmp) hipFree(d_scan_tmp);
        if (d_parent) hipFree(d_parent);
        if (d_pslot) hipFree(d_pslot);
        if (d_prank) hipFree(d_prank);
        if (d_ctr) hipFree(d_ctr);
        if (h_ctr) hipHostFree(h_ctr);
        if (d_lc) hipFree(d_lc);
        if (h_lc) hipHostFree(h_lc);
        for (int i = 0; i < 2; i++) { if (ev_e[i]) hipEventDestroy(ev_e[i]); if (ev_m[i]) hipEventDestroy(ev_m[i]); }
        if (stream2) hipStreamDestroy(stream2);
        if (stream) hipStreamDestroy(stream);
    }

    template <class F>
    void timed(int which, uint64_t units, F &&launch, hipStream_t on = nullptr) {
        if (!timer.enabled) { launch(); return; }
        if (!on) on = stream;
        hipEvent_t a = timer.get(), b = timer.get();
        size_t ia = timer.used - 2, ib = timer.used - 1;
        hipEventRecord(a, on);
        launch();
        hipEventRecord(b, on);
        timer.pending.push_back({which, ia, ib, units});
    }

    int read_counters() {
        HIP_TRY(hipStreamSynchronize(stream2));
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (timer.enabled) timer.resolve(kstat);
        return MC_OK;
    }
    int check_dev_error() {
        if (h_ctr->error & DEV_EOVERFLOW) { set_error("packed-state capacity exceeded (raft messages / elections / allLogs slots, or a PlusCal sequence longer than its cells)"); return MC_EOVERFLOW; }
        if (h_ctr->error & DEV_ETABLE) { set_error("seen-set full: raise table_capacity"); return MC_ETABLEFULL; }
        if (h_ctr->error & DEV_EROUTE) { set_error("sharded round: an exchange bucket is full (raise the fan-out allowance / send capacity)"); return MC_EROUTE; }
        if (h_ctr->error & DEV_EARENA) { set_error("state arena full: raise arena_capacity"); return MC_EARENA; }
        return MC_OK;
    }

    bool use_matrix = false;
    // fused runs of a by-family spec: the expand wavefronts write their own survivors (MC_F_NOINWAVE = A/B: everything through the
    // new-list and k_materialise, as in rounds 1-3)
    double inwave_growth_limit = getenv("TLAMC_INWAVE_GROWTH") ? atof(getenv("TLAMC_INWAVE_GROWTH")) : 2.3;  // (A/B knob; profiles/r04l: t3 — levels
    // grow by at most 2.2 x — is best with every level in-wave, 159.6 against 161.8 ms at 1.7; the 5-server model — 2.4 x and more on all
    // 18 levels — with none, 230.6 against 239.0 ms)
    bool inwave_ok() const { return UsesFamilies<S>::value && !use_matrix && !(cfg.flags & (MC_F_NOFAMILY | MC_F_NOINWAVE)); }
    void set_inwave(RouteArgs &rt) const {
        if (!inwave_ok()) return;
        rt.arena_w = d_arena;
        rt.arena_cap = arena_cap;
        rt.parent = d_parent;
        rt.pslot = d_pslot;
    }
    // materialise + commit of the chunk whose survivors are in new-list `parity`, on the second stream: it overlaps
    // the expansion of the next chunk (memory-bound next to latency-bound)
    void finish_materialise(uint64_t chunk_base, uint64_t ncols, unsigned parity) {
        const unsigned bx = (unsigned)((ncols + 255) / 256);
        const unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
        hipEventRecord(ev_e[parity], stream);
        hipStreamWaitEvent(stream2, ev_e[parity], 0);
        timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise<S>, dim3(gm, NSHARD), dim3(256), 0, stream2, prm, d_arena, chunk_base, d_newlist,
                               seg_cap, arena_cap, d_parent, d_pslot, d_ctr, parity, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
        }, stream2);
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, parity);
        hipEventRecord(ev_m[parity], stream2);
        // measurement knob: TLAMC_SERIAL=1 lets no expand kernel run beside a materialise kernel (each kernel's stand-alone time)
        static const bool serial = getenv("TLAMC_SERIAL") != nullptr;
        if (serial) hipStreamSynchronize(stream2);
    }
    // one batched level (LevelCtl): the same pair of kernels, ranges read on the device, then the level is closed there
    void enqueue_blind_level(uint64_t max_states) {
        // all four kernels on ONE stream: for a frontier this small the two-stream overlap buys nothing and the
        // cross-stream events would cost more than the kernels
        const uint64_t ncols = max_states + 64;
        RouteArgs rt{};
        rt.lc = d_lc;
        rt.new_fp = d_newfp;
        set_inwave(rt);
        const unsigned sg = slices_for(ncols, true);
        const bool dl = sg > 1 && (cfg.flags & MC_F_DEADLOCK);
        if (dl) { rt.succ = d_nsl; hipMemsetAsync(d_nsl, 0, ncols * sizeof(uint16_t), stream); }
        timed(0, 0, [&] {
            launch_expand<S, false>(!(cfg.flags & MC_F_NOFAMILY), cfg.flags, ncols, stream, sg, prm, (const uint64_t *)d_arena,
                                    (uint64_t)0, (uint64_t)0, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr, cfg.flags, rt, 0u);
            if (dl) hipLaunchKernelGGL(k_deadlock_slices, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, (const uint16_t *)d_nsl,
                                       (uint64_t)0, (uint64_t)0, ncols, (const LevelCtl *)d_lc, d_ctr);
        });
        timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise<S>, dim3(32, NSHARD), dim3(256), 0, stream, prm, d_arena, (uint64_t)0, d_newlist, seg_cap,
                               arena_cap, d_parent, d_pslot, d_ctr, 0u, (const LevelCtl *)d_lc, (const uint64_t *)d_newfp);
        });
        hipLaunchKernelGGL(k_end_level, dim3(1), dim3(1), 0, stream, d_ctr, d_lc);
    }

    // insert + materialise + commit for the candidate matrix just written
    template <bool INIT>
    void finish_chunk(uint64_t chunk_base_or_first, uint64_t ncols, unsigned rows) {
        const unsigned bx = (unsigned)((ncols + 255) / 256);
        timed(1, ncols * rows, [&] {
            hipLaunchKernelGGL(k_insert, dim3(bx, rows), dim3(256), 0, stream, d_cand, row_stride, ncols, d_nsl, d_table,
                               seen_arg(), d_newlist, d_ctr);
        });
        const unsigned gm = bx < 2048 ? bx : 2048;
        timed(2, 0, [&] {
            if (INIT)
                hipLaunchKernelGGL(k_init_materialise<S>, dim3(gm), dim3(256), 0, stream, prm, d_arena, chunk_base_or_first,
                                   d_inittmp, d_newlist, arena_cap, d_parent, d_pslot, d_ctr);
            else
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm, 1), dim3(256), 0, stream, prm, d_arena, chunk_base_or_first,
                                   d_newlist, seg_cap, arena_cap, d_parent, d_pslot, d_ctr, 0u, (const LevelCtl *)nullptr, (const uint64_t *)nullptr);
        });
        hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream, d_ctr, 0u);
    }

    int run(mc_result *out) override {
        memset(out, 0, sizeof *out);
        out->violated_invariant = -1;
        memset(kstat, 0, sizeof kstat);
        have_viol = false;
        have_run = false;  // set again on the success path only: a run that fails half-way (MC_EARENA, MC_ETABLEFULL, MC_EOVERFLOW)
                           // leaves fingerprints of an unfinished level in the seen-set — the next step / checkpoint must not continue it
        level_start.clear();
        HIP_TRY(hipSetDevice(cfg.device));
        const bool in_place = ck_pending && ck_in_place;  // mc_engine_step: the seen-set of the stopped run is still valid
        ck_in_place = false;
        if (!in_place) HIP_TRY(hipMemsetAsync(d_table, 0, table_cap * sizeof(uint64_t), stream));
        HIP_TRY(hipMemsetAsync(d_ctr, 0, sizeof(DevCounters), stream));
        DevCounters init_c;
        memset(&init_c, 0, sizeof init_c);
        init_c.viol_key = ~0ull;
        const bool resuming = ck_pending;
        ck_pending = false;
        if (resuming) {  // counters of the checkpointed run; its states are already in the arena (restore())
            init_c.arena_next = ck_distinct;
            init_c.generated[0].v = ck_generated;
            init_c.cells[0].v = ck_cells;
        }
        HIP_TRY(hipMemcpyAsync(d_ctr, &init_c, sizeof init_c, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const auto t0 = std::chrono::steady_clock::now();
        progress_last = t0;
        auto progress = [&](uint32_t lv, uint64_t lo_, uint64_t hi_) {
            if (!progress_fn) return;
            uint64_t g = 0;
            for (int t = 0; t < NSHARD; t++) g += h_ctr->generated[t].v;
            report_progress(lv, g, hi_, hi_ - lo_);
        };

        // level 1: Init
        const uint64_t ninit = resuming ? 0 : S::num_init(prm);
        if (resuming && ck_distinct && !in_place)
            hipLaunchKernelGGL(k_reseed_table<S>, dim3((unsigned)((ck_distinct + 255) / 256)), dim3(256), 0, stream, prm,
                               (const uint64_t *)d_arena, ck_distinct, d_table, seen_arg(), d_ctr);
        for (uint64_t first = 0; first < ninit; first += chunk) {
            const uint64_t count = ninit - first < chunk ? ninit - first : chunk;
            const uint64_t ncols = (count + 63) & ~63ull;
            hipLaunchKernelGGL(k_init_cand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, first, count,
                               d_inittmp, d_cand, ncols, d_nsl, d_ctr, 0u, 1u);  // run() is never owner-filtered
            finish_chunk<true>(first, ncols, 1);
        }
        // with in-wave writes every writer takes its arena indices from arena_next itself from here on (DevCounters::atomic_alloc).
        // Init above, the candidate-matrix form and every spec whose states all go through k_materialise keep appending in stream
        // order: one returning atomicAdd per wavefront of k_materialise on ONE word is pure cost there (atomic_add N = 28: 145
        // against 108 ms per run, profiles/r04h / r04i; raft through k_materialise only: 203 against 159 ms)
        if (inwave_ok()) hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, 1u);
        bool alloc_atomic = inwave_ok();
        uint64_t prev_frontier = ~0ull >> 12;  // (the first level is never "fast growing")
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        uint64_t lo = 0, hi = h_ctr->arena_next;
        uint32_t level = 1;
        out->level_distinct[0] = hi;
        level_start.push_back(0);
        if (resuming) {  // continue with the frontier [lo, hi) = the last, unexpanded level of the checkpointed run
            level_start = ck_level_start;
            level = (uint32_t)level_start.size();
            lo = ck_lo;
            for (uint32_t k = 0; k < level && k < MC_MAX_LEVELS; k++)
                out->level_distinct[k] = (k + 1 < level ? level_start[k + 1] : hi) - level_start[k];
        }
        int budget = 0;
        const uint64_t blind_max = chunk < (1ull << 16) ? chunk : (1ull << 16);
        while (hi > lo) {
            if (h_ctr->viol_key != ~0ull) break;
            if (stop_frontier && hi - lo >= stop_frontier) break;  // the caller continues this level sharded
            if (cfg.max_levels && level >= cfg.max_levels) { budget = 1; break; }
            if (cfg.max_distinct && hi >= cfg.max_distinct) { budget = 1; break; }
            if (!use_matrix && !(cfg.flags & MC_F_NOBATCH) && hi - lo <= blind_max) {
                // Small frontier: BLIND_BATCH levels are enqueued back to back; the kernels take each level's range from
                // LevelCtl and k_end_level applies the same stopping rules as this loop, so one host round trip covers
                // up to BLIND_BATCH levels instead of one (a level of a few thousand states is pure launch latency).
                memset(h_lc, 0, sizeof *h_lc);
                h_lc->lo = lo;
                h_lc->hi = hi;
                h_lc->max_states = blind_max;
                h_lc->max_distinct = cfg.max_distinct;
                h_lc->levels_left = cfg.max_levels ? (unsigned)(cfg.max_levels - level) : 0u;
                HIP_TRY(hipMemcpyAsync(d_lc, h_lc, sizeof *h_lc, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipStreamSynchronize(stream2));  // (nothing of an earlier level is still being materialised)
                if (inwave_ok() && !alloc_atomic) {  // the batched small levels always write in-wave
                    hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, 1u);
                    alloc_atomic = true;
                }
                for (int k = 0; k < BLIND_BATCH; k++) enqueue_blind_level(blind_max);
                HIP_TRY(hipMemcpyAsync(h_lc, d_lc, sizeof *h_lc, hipMemcpyDeviceToHost, stream));
                if ((rc = read_counters())) return rc;
                if ((rc = check_dev_error())) return rc;
                for (unsigned k = 0; k < h_lc->nlev; k++) {
                    kstat[0].units += hi - lo;  // states expanded by this level (the launches were timed with 0 units)
                    lo = hi;
                    hi = h_lc->level_hi[k];
                    if (hi > lo) {
                        level_start.push_back(lo);
                        if (level < MC_MAX_LEVELS) out->level_distinct[level] = hi - lo;
                        level++;
                    }
                    if (level >= MC_MAX_LEVELS) { set_error("too many BFS levels"); return MC_EBADCFG; }
                }
                progress(level, lo, hi);
                prev_frontier = hi > lo ? hi - lo : 1;
                continue;  // the loop head re-checks violation / budgets / frontier with the host's copies
            }
            // IN-WAVE WRITES, level by level: they pay where a wavefront has about one survivor per parent (its list holds them
            // all, one workgroup tail per 256 parents); in a level that grows fast (config 4's model: x 2.4 on every one of its 18
            // levels) most survivors overflow into the new-list anyway and k_materialise alone is the better writer (924 M states:
            // 234.7 against 239.0 ms, profiles/r04k).  The level before this one says which kind it is; the allocation mode
            // (DevCounters::atomic_alloc) follows — nothing is in flight between two levels.
            const bool lvl_inwave = inwave_ok() && (double)(hi - lo) <= inwave_growth_limit * (double)prev_frontier;
            if (inwave_ok() && lvl_inwave != alloc_atomic) {
                hipLaunchKernelGGL(k_set_alloc_mode, dim3(1), dim3(1), 0, stream, d_ctr, lvl_inwave ? 1u : 0u);
                alloc_atomic = lvl_inwave;
            }
            prev_frontier = hi - lo;
            unsigned chunk_no = 0;
            for (uint64_t c0 = lo; c0 < hi; ++chunk_no) {
                const unsigned parity = chunk_no & 1u;
                const uint64_t base = c0 & ~63ull;
                uint64_t c1 = base + chunk;  // chunk boundaries stay 64-aligned
                if (c1 > hi) c1 = hi;
                const uint64_t ncols = ((c1 - base) + 63) & ~63ull;
                if (use_matrix) {  // two-kernel form: sparse candidate matrix + k_insert (kept for A/B measurements)
                    timed(0, c1 - c0, [&] {
                        hipLaunchKernelGGL(k_expand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, d_arena,
                                           c0, c1, d_cand, row_stride, ncols, d_nsl, d_ctr, cfg.flags);
                    });
                    finish_chunk<false>(base, ncols, max_slots);
                } else {
                    if (chunk_no >= 2) hipStreamWaitEvent(stream, ev_m[parity], 0);  // new-list `parity` is free again
                    RouteArgs rt_new{};
                    rt_new.new_fp = d_newfp;
                    if (lvl_inwave) set_inwave(rt_new);
                    const unsigned sg = slices_for(ncols, false);
                    const bool dl = sg > 1 && (cfg.flags & MC_F_DEADLOCK);
                    if (dl) { rt_new.succ = d_nsl; hipMemsetAsync(d_nsl, 0, ncols * sizeof(uint16_t), stream); }
                    timed(0, c1 - c0, [&] {
                        launch_expand<S, false>(!(cfg.flags & MC_F_NOFAMILY), cfg.flags, ncols, stream, sg, prm,
                                                (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr,
                                                cfg.flags, rt_new, parity);
                        if (dl) hipLaunchKernelGGL(k_deadlock_slices, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream,
                                                   (const uint16_t *)d_nsl, c0, c1, ncols, (const LevelCtl *)nullptr, d_ctr);
                    });
                    finish_materialise(base, ncols, parity);
                }
                c0 = c1;
            }
            if ((rc = read_counters())) return rc;
            if ((rc = check_dev_error())) return rc;
            lo = hi;
            hi = h_ctr->arena_next;
            if (hi > lo) {
                level_start.push_back(lo);
                if (level < MC_MAX_LEVELS) out->level_distinct[level] = hi - lo;
                level++;
            }
            if (level >= MC_MAX_LEVELS) { set_error("too many BFS levels"); return MC_EBADCFG; }
            progress(level, lo, hi);
        }
        // a run that stops with an unexpanded frontier (budget) has not evaluated that level's check-on-expand invariants yet
        if (hi > lo && h_ctr->viol_key == ~0ull && !stop_frontier && (rc = check_frontier(lo, hi))) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        out->seconds = std::chrono::duration<double>(t1 - t0).count();
        out->distinct = h_ctr->arena_next;
        last_distinct = h_ctr->arena_next;
        kstat[2].units = h_ctr->arena_next;
        out->generated = 0;
        kstat_cells = 0;
        for (int t = 0; t < NSHARD; t++) { out->generated += h_ctr->generated[t].v; kstat_cells += h_ctr->cells[t].v; }
        last_generated = out->generated;
        // (a resumed run counts from the checkpoint on: via_list starts at 0 there)
        kstat_inwave = inwave_ok() && h_ctr->arena_next >= h_ctr->via_list + (resuming ? ck_distinct : 0)
                           ? h_ctr->arena_next - h_ctr->via_list - (resuming ? ck_distinct : 0) : 0;  // (Init's states went through the list too)
        have_run = true;
        out->queue_left = hi - lo;
        out->depth = level;
        out->levels = level;
        run_lo = lo;
        run_hi = hi;
        if (h_ctr->viol_key != ~0ull) {
            have_viol = true;
            last_viol = h_ctr->viol_key;
            const unsigned kind = (unsigned)(last_viol & 7u);
            out->verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT
                         : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
            if (kind == VK_INVARIANT) out->violated_invariant = (int)(last_viol >> 3 & 31u);
            std::vector<uint64_t> chain;
            if (build_chain(chain) == MC_OK) {
                const unsigned slot = (unsigned)(last_viol >> 8 & 0xffffu);
                const bool extra = kind == VK_INVARIANT && slot < SLOT_PARENT;  // violating successor itself
                out->trace_len = (uint32_t)chain.size() + (extra ? 1u : 0u);
            }
        } else {
            out->verdict = budget ? MC_V_BUDGET : MC_V_OK;
        }
        return MC_OK;
    }
    uint64_t kstat_cells = 0, kstat_inwave = 0;
    uint64_t stop_frontier = 0, run_lo = 0, run_hi = 0;  // run() stops before a level of >= stop_frontier states (0 = never)

    // ------------------------------------------------------------------------------- checkpoint / recover
    // TLC checkpoints a run into its states/ directory and continues it with -recover (testout1:10: "-- Checkpointing of
    // run states/01-08-03-18-14-01 completed."; .gitignore:2).  Here the resident arena already IS the state store: a
    // checkpoint is the arena's used blocks as they lie in HBM, the level boundaries, the counters and (with MC_F_TRACE)
    // the parent pointers.  The seen-set is not written: recovery re-inserts word 0 of every state (k_reseed_table).
    struct CkHeader {
        char magic[8];
        uint32_t spec_id, nparams;
        int64_t params[16];
        uint32_t words, has_trace;
        uint64_t distinct, generated, cells, lo, hi, nlevels;
    };
    struct FileCloser {
        FILE *f;
        ~FileCloser() { if (f) fclose(f); }
    };
    bool have_run = false, ck_pending = false, ck_in_place = false;
    // mc_engine_step: `levels` more BFS levels.  The first call (or a call after a run that ended: complete, or with an error)
    // starts from Init; a call after a budget stop continues IN PLACE — arena, seen-set and parent pointers stay where they
    // are in HBM, only the counters and the level table are handed over (the same hand-over as restore(), without the file
    // and without rebuilding the seen-set).
    int step(uint32_t levels, mc_result *out) override {
        if (!levels) return MC_EBADCFG;
        const uint64_t saved = cfg.max_levels;
        const bool cont = have_run && !have_viol && run_hi > run_lo && !ck_pending;
        if (cont) {
            ck_distinct = last_distinct; ck_generated = last_generated; ck_cells = kstat_cells; ck_lo = run_lo;
            ck_level_start = level_start;
            ck_pending = ck_in_place = true;
            cfg.max_levels = (uint64_t)level_start.size() + levels;
        } else if (ck_pending) {  // restore() handed a checkpoint over: `levels` more levels beyond the checkpointed ones
            cfg.max_levels = (uint64_t)ck_level_start.size() + levels;
        } else {
            cfg.max_levels = levels;
        }
        const int rc = run(out);
        cfg.max_levels = saved;
        if (rc) ck_pending = ck_in_place = false;  // after a failure the next step starts over
        return rc;
    }
    uint64_t last_generated = 0, ck_distinct = 0, ck_generated = 0, ck_cells = 0, ck_lo = 0;
    std::vector<uint64_t> ck_level_start;
    bool ck_params_comparable() const { return desc.spec_id != MC_SPEC_PCAL; }  // a compiled program's parameter is a host pointer
    int dev_to_file(const void *dev, size_t bytes, FILE *f) {
        std::vector<char> buf(bytes < (64u << 20) ? bytes : (64u << 20));
        for (size_t off = 0; off < bytes; off += buf.size()) {
            const size_t n = bytes - off < buf.size() ? bytes - off : buf.size();
            HIP_TRY(hipMemcpy(buf.data(), (const char *)dev + off, n, hipMemcpyDeviceToHost));
            if (fwrite(buf.data(), 1, n, f) != n) { set_error("checkpoint: short write"); return MC_EBADCFG; }
        }
        return MC_OK;
    }
    int file_to_dev(void *dev, size_t bytes, FILE *f) {
        std::vector<char> buf(bytes < (64u << 20) ? bytes : (64u << 20));
        for (size_t off = 0; off < bytes; off += buf.size()) {
            const size_t n = bytes - off < buf.size() ? bytes - off : buf.size();
            if (fread(buf.data(), 1, n, f) != n) { set_error("restore: the checkpoint file is truncated"); return MC_EPARSE; }
            HIP_TRY(hipMemcpy((char *)dev + off, buf.data(), n, hipMemcpyHostToDevice));
        }
        return MC_OK;
    }
    int checkpoint(const char *path) override {
        if (cfg.shard_count > 1) { set_error("checkpoint: not available for a sharded engine"); return MC_EBADCFG; }
        if (!have_run || have_viol) { set_error("checkpoint: needs a completed mc_engine_run that stopped on a budget (or finished) without a violation"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        FILE *f = fopen(path, "wb");
        if (!f) { set_error(std::string("checkpoint: cannot write ") + path); return MC_EBADCFG; }
        FileCloser closer{f};  // early returns (HIP errors) must not leak the handle
        CkHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, "TLAMCCK1", 8);
        h.spec_id = desc.spec_id;
        h.nparams = ck_params_comparable() ? desc.nparams : 1;
        for (uint32_t i = 0; i < h.nparams && i < 16; i++) h.params[i] = desc.params[i];
        if (!ck_params_comparable()) h.params[0] = (int64_t)program_hash;  // a compiled program is identified by its image
        h.words = (uint32_t)W;
        h.has_trace = d_parent ? 1u : 0u;
        h.distinct = last_distinct; h.generated = last_generated; h.cells = kstat_cells;
        h.lo = run_lo; h.hi = run_hi; h.nlevels = level_start.size();
        int rc = MC_OK;
        if (fwrite(&h, sizeof h, 1, f) != 1 || fwrite(level_start.data(), sizeof(uint64_t), level_start.size(), f) != level_start.size()) {
            set_error("checkpoint: short write");
            rc = MC_EBADCFG;
        }
        const size_t blocks = (size_t)((last_distinct + 63) >> 6);
        if (!rc) rc = dev_to_file(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_parent, (size_t)last_distinct * sizeof(uint32_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_pslot, (size_t)last_distinct * sizeof(uint16_t), f);
        closer.f = nullptr;
        if (fclose(f) != 0 && !rc) { set_error("checkpoint: close failed"); rc = MC_EBADCFG; }
        return rc;
    }
    int restore(const char *path) override {
        if (cfg.shard_count > 1) { set_error("restore: not available for a sharded engine"); return MC_EBADCFG; }
        FILE *f = fopen(path, "rb");
        if (!f) { set_error(std::string("restore: cannot read ") + path); return MC_EPARSE; }
        FileCloser closer{f};
        CkHeader h;
        int rc = MC_OK;
        auto fail = [&](int code, const char *msg) { set_error(msg); rc = code; };
        if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "TLAMCCK1", 8) != 0) fail(MC_EPARSE, "restore: not a checkpoint file");
        else if (h.spec_id != desc.spec_id || h.words != (uint32_t)W) fail(MC_EBADCFG, "restore: the checkpoint belongs to another spec");
        else if (ck_params_comparable() && (h.nparams != desc.nparams || memcmp(h.params, desc.params, sizeof(int64_t) * (h.nparams < 16 ? h.nparams : 16)) != 0))
            fail(MC_EBADCFG, "restore: the checkpoint was written with other constants / invariants");
        else if (!ck_params_comparable() && (h.nparams != 1 || (uint64_t)h.params[0] != program_hash))
            fail(MC_EBADCFG, "restore: the checkpoint was written by another compiled program (algorithm, constants, invariants or constraints differ)");
        // a run may legitimately stop on a budget at a seen-set load of 0.5 - 0.8 (the bench model at 2^27 slots ends at 0.76): its
        // checkpoint must restore into the same configuration; beyond 0.9 the reseeding itself could not finish (DEV_ETABLE)
        else if (h.distinct * 10 > table_cap * 9) fail(MC_ETABLEFULL, "restore: table_capacity cannot hold the checkpoint's states (load > 0.9)");
        else if (h.distinct > arena_cap) fail(MC_EARENA, "restore: arena_capacity is smaller than the checkpoint");
        else if (h.hi != h.distinct || h.lo > h.hi || h.nlevels == 0 || h.nlevels >= MC_MAX_LEVELS) fail(MC_EPARSE, "restore: inconsistent header");
        else if (d_parent && !h.has_trace) fail(MC_EBADCFG, "restore: the checkpoint holds no parent pointers (written without MC_F_TRACE); run without MC_F_TRACE");
        if (!rc) {
            ck_level_start.assign((size_t)h.nlevels, 0);
            if (fread(ck_level_start.data(), sizeof(uint64_t), ck_level_start.size(), f) != ck_level_start.size()) fail(MC_EPARSE, "restore: the checkpoint file is truncated");
            else {  // level table: starts at 0, strictly increasing, below the fill level; an unexpanded frontier IS the last level
                bool ok = ck_level_start[0] == 0 && ck_level_start.back() < h.hi && (h.lo == h.hi || ck_level_start.back() == h.lo);
                for (size_t k = 1; k < ck_level_start.size() && ok; k++) ok = ck_level_start[k] > ck_level_start[k - 1];
                if (!ok) fail(MC_EPARSE, "restore: inconsistent level table");
            }
        }
        if (!rc) {
            HIP_TRY(hipSetDevice(cfg.device));
            const size_t blocks = (size_t)((h.distinct + 63) >> 6);
            rc = file_to_dev(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
            if (!rc && d_parent && h.has_trace) rc = file_to_dev(d_parent, (size_t)h.distinct * sizeof(uint32_t), f);
            if (!rc && d_parent && h.has_trace) rc = file_to_dev(d_pslot, (size_t)h.distinct * sizeof(uint16_t), f);
        }
        if (rc) return rc;
        ck_distinct = h.distinct; ck_generated = h.generated; ck_cells = h.cells; ck_lo = h.lo;
        ck_pending = true;  // the next run() continues from here
        return MC_OK;
    }

    int fetch_state(uint64_t idx, uint64_t *words) {
        const uint64_t *src = d_arena + ((idx >> 6) * (uint64_t)W) * 64 + (idx & 63);
        HIP_TRY(hipMemcpy2D(words, sizeof(uint64_t), src, 64 * sizeof(uint64_t), sizeof(uint64_t), W, hipMemcpyDeviceToHost));
        return MC_OK;
    }
    // arena indices from an initial state to the state the violation was found in / from
    int build_chain(std::vector<uint64_t> &chain) {
        chain.clear();
        if (!have_viol) return MC_ESTATE;
        const unsigned slot = (unsigned)(last_viol >> 8 & 0xffffu);
        if (slot == SLOT_INIT) return MC_OK;  // violated by an initial state: handled by the caller
        if (!d_parent) { set_error("engine created without MC_F_TRACE"); return MC_ESTATE; }
        uint64_t idx = viol_idx(last_viol);
        for (int guard = 0; guard < MC_MAX_LEVELS; ++guard) {
            chain.push_back(idx);
            uint32_t p;
            HIP_TRY(hipMemcpy(&p, d_parent + idx, sizeof p, hipMemcpyDeviceToHost));
            if (p == 0xffffffffu) break;
            idx = p;
        }
        std::vector<uint64_t> rev(chain.rbegin(), chain.rend());
        chain.swap(rev);
        return MC_OK;
    }
    int trace(uint8_t *states_out, int32_t *actions_out, size_t *n_inout) override {
        if (!have_viol) { *n_inout = 0; return MC_OK; }
        HIP_TRY(hipSetDevice(cfg.device));
        const unsigned kind = (unsigned)(last_viol & 7u), slot = (unsigned)(last_viol >> 8 & 0xffffu);
        std::vector<uint64_t> words;
        std::vector<int32_t> acts;
        if (slot == SLOT_INIT) {  // an initial state violates an invariant
            words.resize(W);
            S::init(prm, viol_idx(last_viol), WordRef{words.data(), 1});
            acts.push_back(-1);
        } else {
            std::vector<uint64_t> chain;
            int rc = build_chain(chain);
            if (rc) return rc;
            words.resize(chain.size() * W);
            for (size_t k = 0; k < chain.size(); ++k) {
                if ((rc = fetch_state(chain[k], &words[k * W]))) return rc;
                if (k == 0) acts.push_back(-1);
                else {
                    uint16_t ps;
                    HIP_TRY(hipMemcpy(&ps, d_pslot + chain[k], sizeof ps, hipMemcpyDeviceToHost));
                    acts.push_back(S::action_of(prm, &words[(k - 1) * W], (int)ps));
                }
            }
            if (kind == VK_INVARIANT && slot < SLOT_PARENT) {  // append the violating successor
                std::vector<uint64_t> last(words.end() - W, words.end());
                words.resize(words.size() + W);
                S::apply(prm, CWordRef{last.data(), 1}, (int)slot, WordRef{&words[words.size() - W], 1});
                acts.push_back(S::action_of(prm, last.data(), (int)slot));
            }
        }
        const size_t n = acts.size();
        if (n > *n_inout) { *n_inout = n; set_error("trace buffer too small"); return MC_EBADCFG; }
        memcpy(states_out, words.data(), n * W * sizeof(uint64_t));
        memcpy(actions_out, acts.data(), n * sizeof(int32_t));
        *n_inout = n;
        return MC_OK;
    }
    int read_states(uint64_t first, uint64_t count, uint8_t *out) override {
        if (!count) return MC_OK;
        if (first + count > last_distinct) { set_error("read_states: range beyond the states found by the last run"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        uint64_t *tmp = nullptr;
        const uint64_t piece = 1ull << 20;
        HIP_TRY(hipMalloc(&tmp, (count < piece ? count : piece) * W * sizeof(uint64_t)));
        for (uint64_t off = 0; off < count; off += piece) {
            const uint64_t n = count - off < piece ? count - off : piece;
            hipLaunchKernelGGL(k_gather_states, dim3((unsigned)((n * W + 255) / 256)), dim3(256), 0, stream, d_arena, W, first + off, n, tmp);
            hipError_t e1 = hipMemcpyAsync(out + off * W * sizeof(uint64_t), tmp, n * W * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
            hipError_t e2 = hipStreamSynchronize(stream);
            if (e1 != hipSuccess || e2 != hipSuccess) { hipFree(tmp); set_error("read_states: copy failed"); return MC_EHIP; }
        }
        hipFree(tmp);
        return MC_OK;
    }
    uint64_t last_distinct = 0;
    // Profiling aid: expand every resident state again (seen-set already full, so every probe hits)
    // with optional ablation flags; returns the kernel time.  State counts are not changed.
    int debug_phases(uint64_t *out48, int reset) override {
#ifdef MC_PHASE_PROF
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_phase), 48 * sizeof(uint64_t)));
        if (reset) { static const unsigned long long zero[48] = {}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof zero)); }
        return MC_OK;
#else
        (void)out48; (void)reset;
        set_error("the library was built without MC_PHASE_PROF");
        return MC_ESTATE;
#endif
    }
    int debug_reexpand(unsigned extra_flags, double *ms) override {
        HIP_TRY(hipSetDevice(cfg.device));
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        HIP_TRY(hipEventRecord(a, stream));
        for (uint64_t c0 = 0; c0 < last_distinct; c0 += chunk) {
            const uint64_t c1 = c0 + chunk < last_distinct ? c0 + chunk : last_distinct;
            const uint64_t ncols = ((c1 - c0) + 63) & ~63ull;
            launch_expand<S, false>(!((cfg.flags | extra_flags) & MC_F_NOFAMILY), cfg.flags | extra_flags, ncols, stream, 0u, prm,
                                    (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr,
                                    cfg.flags | extra_flags, RouteArgs{}, 0u);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream, d_ctr, 0u);
        }
        HIP_TRY(hipEventRecord(b, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, a, b));
        *ms = t;
        hipEventDestroy(a);
        hipEventDestroy(b);
        // restore arena_next (k_commit added nothing: no probe can be new) and leave counters as they were
        return MC_OK;
    }
    // ------------------------------------------------------------------ sharded step API
    uint64_t sh_lo = 0, sh_hi = 0, sh_next = 0;  // local frontier [sh_lo, sh_hi), arena fill level
    // Two expand slots: the route-mode expand of chunk r+1 (on `stream`) overlaps the exchange, probe and keep /
    // materialise of chunk r (kernels on `stream2`, collectives on the caller's stream).
    struct ShSlot {
        uint64_t *rt_fp = nullptr;
        uint32_t *rt_src = nullptr, *pend_src = nullptr;
        PaddedCounter *rt_cur = nullptr;  // [nranks*NSHARD] route cursors
        uint64_t rt_subcap = 0, pend_cap = 0, pend_total = 0, chunk_base = 0, count = 0, ncols = 0;
        BlockPlan plan;          // of the slot's last shard_materialise (for shard_materialise_parents)
        uint64_t moved = 0;
        OwnerOffsets pend_off;
        bool launched = false, keep_pending = false;
    } sl[2];
    uint32_t *d_new_src = nullptr, *d_incl = nullptr;
    uint64_t new_cap = 0;
    unsigned long long *d_ends = nullptr, *h_ends = nullptr;
    PaddedCounter *h_cur = nullptr;  // pinned copy of one slot's route cursors
    void *d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    unsigned nranks() const { return cfg.shard_count > 1 ? cfg.shard_count : 1; }

    int shard_alloc(ShSlot &q, uint64_t send_cap) {
        const unsigned P = nranks();
        if (send_cap > q.pend_cap) {
            if (q.rt_fp) { hipFree(q.rt_fp); hipFree(q.rt_src); hipFree(q.pend_src); q.rt_fp = nullptr; }
            q.rt_subcap = (send_cap / (P * NSHARD)) * 2 + 4096;
            HIP_TRY(hipMalloc(&q.rt_fp, (size_t)P * NSHARD * q.rt_subcap * sizeof(uint64_t)));
            HIP_TRY(hipMalloc(&q.rt_src, (size_t)P * NSHARD * q.rt_subcap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&q.pend_src, send_cap * sizeof(uint32_t)));
            q.pend_cap = send_cap;
        }
        if (!q.rt_cur) HIP_TRY(hipMalloc(&q.rt_cur, (size_t)(P * NSHARD) * sizeof(PaddedCounter)));
        if (send_cap > new_cap) {
            if (d_new_src) { hipFree(d_new_src); hipFree(d_incl); d_new_src = nullptr; }
            HIP_TRY(hipMalloc(&d_new_src, send_cap * sizeof(uint32_t)));
            HIP_TRY(hipMalloc(&d_incl, send_cap * sizeof(uint32_t)));
            new_cap = send_cap;
        }
        if (!d_ends) {
            HIP_TRY(hipMalloc(&d_ends, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipHostMalloc(&h_ends, 8 * sizeof(unsigned long long)));
            HIP_TRY(hipHostMalloc(&h_cur, (size_t)8 * NSHARD * sizeof(PaddedCounter)));
        }
        return MC_OK;
    }
    int scan_answers(const uint8_t *answers_back, uint64_t n, hipStream_t on) {
        hipcub::TransformInputIterator<uint32_t, AnswerCast, const uint8_t *> in(answers_back, AnswerCast());
        size_t need = 0;
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(nullptr, need, in, d_incl, (int)n, on));
        if (need > scan_tmp_bytes) {
            if (d_scan_tmp) hipFree(d_scan_tmp);
            HIP_TRY(hipMalloc(&d_scan_tmp, need));
            scan_tmp_bytes = need;
        }
        HIP_TRY(hipcub::DeviceScan::InclusiveSum(d_scan_tmp, need, in, d_incl, (int)n, on));
        return MC_OK;
    }
    // a sharded run that failed mid-level (a full route bucket, MC_ETABLEFULL ...) leaves its slots as they were: the remedy the
    // error message names — run the same engine again with a larger allowance — must not trip over an expand "still in flight"
    void shard_reset_slots() {
        hipSetDevice(cfg.device);
        hipDeviceSynchronize();  // (whatever the failed run left on the streams has drained; its events are complete)
        for (auto &q : sl) { q.launched = false; q.keep_pending = false; q.pend_total = 0; q.moved = 0; q.count = 0; }
    }
    int shard_begin() override {
        if (nranks() > 8) { set_error("at most 8 shards"); return MC_EBADCFG; }
        sh_dup = 0;
        sh_resume = sh_ck_ok = false;
        shard_reset_slots();
        HIP_TRY(hipSetDevice(cfg.device));
        memset(kstat, 0, sizeof kstat);
        HIP_TRY(hipMemsetAsync(d_table, 0, table_cap * sizeof(uint64_t), stream));
        DevCounters init_c;
        memset(&init_c, 0, sizeof init_c);
        init_c.viol_key = ~0ull;
        HIP_TRY(hipMemcpyAsync(d_ctr, &init_c, sizeof init_c, hipMemcpyHostToDevice, stream));
        const uint64_t ninit = S::num_init(prm);
        for (uint64_t first = 0; first < ninit; first += chunk) {
            const uint64_t count = ninit - first < chunk ? ninit - first : chunk;
            const uint64_t ncols = (count + 63) & ~63ull;
            hipLaunchKernelGGL(k_init_cand<S>, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, prm, first, count,
                               d_inittmp, d_cand, ncols, d_nsl, d_ctr, cfg.shard_rank, cfg.shard_count);
            finish_chunk<true>(first, ncols, 1);
        }
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        sh_lo = 0;
        sh_hi = sh_next = h_ctr->arena_next;
        last_distinct = sh_next;
        return MC_OK;
    }
    // The first levels of a run are tiny: sharding them costs several collectives per level and balances nothing.
    // Every rank therefore runs the SAME fused BFS (run()) until a level has at least min_frontier states, keeps every
    // nranks-th state of that level as its local frontier, and the sharded rounds start there.  The seen-set of each
    // rank then holds ALL prefix fingerprints (a superset of the ones it owns: harmless); rank 0 alone reports the
    // prefix's `generated`, and distinct_local excludes what other ranks already count.
    uint64_t sh_dup = 0;
    std::unique_ptr<mc_result> prefix_res{new mc_result()};
    int shard_begin_replicated(uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out, uint32_t *nlevels) override {
        if (nranks() > 8) { set_error("at most 8 shards"); return MC_EBADCFG; }
        mc_result &res = *prefix_res;  // large (level table): not on the stack, and not shared between engines / threads
        sh_resume = sh_ck_ok = false;
        shard_reset_slots();
        const uint64_t saved_md = cfg.max_distinct, saved_ml = cfg.max_levels;
        cfg.max_distinct = max_distinct;  // the whole job's budgets: the prefix stops where the single-GPU run would
        cfg.max_levels = max_levels;
        stop_frontier = min_frontier ? min_frontier : 1;
        const int rc = run(&res);
        stop_frontier = 0;
        cfg.max_distinct = saved_md;
        cfg.max_levels = saved_ml;
        if (rc) return rc;
        const uint32_t cap = *nlevels;
        *nlevels = res.levels;
        for (uint32_t k = 0; k < res.levels && k < cap; k++) levels_out[k] = res.level_distinct[k];
        if (res.levels > cap) { set_error("shard_begin_replicated: level buffer too small"); return MC_EBADCFG; }
        const unsigned P = nranks(), r = cfg.shard_rank;
        const uint64_t lo = run_lo, hi = run_hi;
        const bool go_on = h_ctr->viol_key == ~0ull && hi > lo;  // otherwise finished or failed inside the prefix
        if (go_on)
            hipLaunchKernelGGL(k_take_owned<S>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, stream, prm, d_arena, lo, hi, r, P, hi,
                               arena_cap, d_parent, d_pslot, d_ctr);
        hipLaunchKernelGGL(k_after_prefix, dim3(1), dim3(1), 0, stream, d_ctr, (unsigned long long)hi, r != 0 ? 1 : 0);
        int rc2 = read_counters();
        if (rc2) return rc2;
        if ((rc2 = check_dev_error())) return rc2;
        sh_lo = hi;
        sh_hi = sh_next = h_ctr->arena_next;
        sh_dup = r == 0 ? sh_hi - hi : sh_hi;
        last_distinct = sh_next;
        return MC_OK;
    }
    int shard_level_size(uint64_t *n) override { *n = sh_hi - sh_lo; return MC_OK; }
    // ------------------------------------------------------------------ one checkpoint file per rank (testout1:10)
    // A rank's share of a sharded run cannot be rebuilt from its arena alone: the states a rank HOLDS are the ones it generated
    // (stay levels) or was sent (move levels), the fingerprints its seen-set slice holds are the ones it OWNS.  So a rank's file is
    // its arena, its parent pointers (index, slot, rank), its seen-set slice as it lies in HBM, its counters, and the level table of
    // the whole job (the same on every rank; the level loop hands it over when a run ends: shard_note_levels).  Restoring needs
    // the same world size, rank, table_capacity and spec; the next mc_shard_run* then continues with the unexpanded frontier.
    struct ShCkHeader {
        char magic[8];
        uint32_t spec_id, nparams;
        int64_t params[16];
        uint32_t words, has_trace, rank, world;
        uint64_t table_cap, lo, hi, next, dup, nlevels;
    };
    std::vector<uint64_t> sh_levels;
    bool sh_ck_ok = false, sh_resume = false;
    int shard_note_levels(const uint64_t *levels, uint32_t n, int32_t verdict) override {
        sh_levels.assign(levels, levels + n);
        sh_ck_ok = verdict == MC_V_OK || verdict == MC_V_BUDGET;
        return MC_OK;
    }
    int shard_resume(uint64_t *levels_out, uint32_t *nlevels) override {
        const uint32_t cap = *nlevels;
        *nlevels = 0;
        if (!sh_resume) return MC_OK;
        if (sh_levels.size() > cap) { set_error("shard_resume: level buffer too small"); return MC_EBADCFG; }
        for (size_t k = 0; k < sh_levels.size(); k++) levels_out[k] = sh_levels[k];
        *nlevels = (uint32_t)sh_levels.size();
        sh_resume = false;
        return MC_OK;
    }
    int shard_checkpoint(const char *path) override {
        if (!sh_ck_ok || sh_levels.empty()) { set_error("mc_shard_checkpoint: needs a sharded run that ended (finished, or stopped on a budget) without an error"); return MC_EBADCFG; }
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipDeviceSynchronize());
        FILE *f = fopen(path, "wb");
        if (!f) { set_error(std::string("mc_shard_checkpoint: cannot write ") + path); return MC_EBADCFG; }
        FileCloser closer{f};
        ShCkHeader h;
        memset(&h, 0, sizeof h);
        memcpy(h.magic, "TLAMCSK1", 8);
        h.spec_id = desc.spec_id;
        h.nparams = ck_params_comparable() ? desc.nparams : 1;
        for (uint32_t i = 0; i < h.nparams && i < 16; i++) h.params[i] = desc.params[i];
        if (!ck_params_comparable()) h.params[0] = (int64_t)program_hash;
        h.words = (uint32_t)W; h.has_trace = d_parent ? 1u : 0u; h.rank = cfg.shard_rank; h.world = nranks();
        h.table_cap = table_cap | (seen_sparse ? SEEN_SPARSE : 0); h.lo = sh_lo; h.hi = sh_hi; h.next = sh_next; h.dup = sh_dup; h.nlevels = sh_levels.size();
        int rc = MC_OK;
        DevCounters c;
        HIP_TRY(hipMemcpy(&c, d_ctr, sizeof c, hipMemcpyDeviceToHost));
        if (fwrite(&h, sizeof h, 1, f) != 1 || fwrite(sh_levels.data(), sizeof(uint64_t), sh_levels.size(), f) != sh_levels.size() || fwrite(&c, sizeof c, 1, f) != 1) {
            set_error("mc_shard_checkpoint: short write");
            rc = MC_EBADCFG;
        }
        const size_t blocks = (size_t)((sh_next + 63) >> 6);
        if (!rc) rc = dev_to_file(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_parent, (size_t)sh_next * sizeof(uint32_t), f);
        if (!rc && d_parent) rc = dev_to_file(d_pslot, (size_t)sh_next * sizeof(uint16_t), f);
        if (!rc && d_prank) rc = dev_to_file(d_prank, (size_t)sh_next, f);
        if (!rc) rc = dev_to_file(d_table, (size_t)table_cap * sizeof(uint64_t), f);
        closer.f = nullptr;
        if (fclose(f) != 0 && !rc) { set_error("mc_shard_checkpoint: close failed"); rc = MC_EBADCFG; }
        return rc;
    }
    int shard_restore(const char *path) override {
        FILE *f = fopen(path, "rb");
        if (!f) { set_error(std::string("mc_shard_restore: cannot read ") + path); return MC_EPARSE; }
        FileCloser closer{f};
        ShCkHeader h;
        int rc = MC_OK;
        auto fail = [&](int code, const char *msg) { set_error(msg); rc = code; };
        if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, "TLAMCSK1", 8) != 0) fail(MC_EPARSE, "mc_shard_restore: not a per-rank checkpoint file");
        else if (h.spec_id != desc.spec_id || h.words != (uint32_t)W) fail(MC_EBADCFG, "mc_shard_restore: the checkpoint belongs to another spec");
        else if (ck_params_comparable() && (h.nparams != desc.nparams || memcmp(h.params, desc.params, sizeof(int64_t) * (h.nparams < 16 ? h.nparams : 16)) != 0))
            fail(MC_EBADCFG, "mc_shard_restore: the checkpoint was written with other constants / invariants");
        else if (!ck_params_comparable() && (h.nparams != 1 || (uint64_t)h.params[0] != program_hash))
            fail(MC_EBADCFG, "mc_shard_restore: the checkpoint was written by another compiled program");
        else if (h.rank != cfg.shard_rank || h.world != nranks()) fail(MC_EBADCFG, "mc_shard_restore: the file is another rank's, or of a run with another number of ranks");
        else if (h.table_cap != (table_cap | (seen_sparse ? SEEN_SPARSE : 0)))
            fail(MC_EBADCFG, "mc_shard_restore: table_capacity differs from the checkpointed run's, or the bucket form it implies (4 slots when the table is >= 3 x the arena): the seen-set slice is stored as it lay in HBM");
        else if (h.next > arena_cap) fail(MC_EARENA, "mc_shard_restore: arena_capacity is smaller than the checkpoint");
        else if (h.lo > h.hi || h.hi != h.next || h.dup > h.next || h.nlevels == 0 || h.nlevels >= MC_MAX_LEVELS) fail(MC_EPARSE, "mc_shard_restore: inconsistent header");
        else if ((d_parent != nullptr) != (h.has_trace != 0)) fail(MC_EBADCFG, "mc_shard_restore: MC_F_TRACE differs from the checkpointed run's");
        DevCounters c;
        if (!rc) {
            sh_levels.assign((size_t)h.nlevels, 0);
            if (fread(sh_levels.data(), sizeof(uint64_t), sh_levels.size(), f) != sh_levels.size() || fread(&c, sizeof c, 1, f) != 1) fail(MC_EPARSE, "mc_shard_restore: the checkpoint file is truncated");
            else if (c.arena_next != h.next || c.viol_key != ~0ull || c.error) fail(MC_EPARSE, "mc_shard_restore: inconsistent counters");
        }
        if (!rc) {
            HIP_TRY(hipSetDevice(cfg.device));
            HIP_TRY(hipDeviceSynchronize());
            const size_t blocks = (size_t)((h.next + 63) >> 6);
            rc = file_to_dev(d_arena, blocks * (size_t)W * 64 * sizeof(uint64_t), f);
            if (!rc && d_parent) rc = file_to_dev(d_parent, (size_t)h.next * sizeof(uint32_t), f);
            if (!rc && d_parent) rc = file_to_dev(d_pslot, (size_t)h.next * sizeof(uint16_t), f);
            if (!rc && d_prank) rc = file_to_dev(d_prank, (size_t)h.next, f);
            if (!rc) rc = file_to_dev(d_table, (size_t)table_cap * sizeof(uint64_t), f);
            if (!rc && fgetc(f) != EOF) fail(MC_EPARSE, "mc_shard_restore: trailing bytes (not the file this engine's configuration wrote)");
        }
        if (rc) { sh_levels.clear(); return rc; }
        for (auto &n : c.n_new) n.v = 0;
        c.max_slots = 0;
        c.atomic_alloc = 0;
        HIP_TRY(hipMemcpy(d_ctr, &c, sizeof c, hipMemcpyHostToDevice));
        memset(kstat, 0, sizeof kstat);
        sh_lo = h.lo; sh_hi = h.hi; sh_next = h.next; sh_dup = h.dup;
        last_distinct = sh_next;
        have_viol = false;
        sl[0].launched = sl[1].launched = false;
        sh_resume = true;
        sh_ck_ok = false;
        return MC_OK;
    }
    int shard_info(void **main_stream, uint64_t *chunk_states, int32_t *traced) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (main_stream) *main_stream = (void *)stream;
        if (chunk_states) *chunk_states = chunk;
        if (traced) *traced = d_parent != nullptr;
        return MC_OK;
    }
    size_t state_bytes() const override { return (size_t)W * 8; }
    // Side stream of the sharded path.  By default the engine's own second stream, and every step call returns
    // with its work finished.  A caller that runs its collectives on a HIP stream hands that stream over with
    // shard_set_stream: compaction / probe / keep / materialise / ingest are then enqueued on it WITHOUT host
    // synchronisation (stream order ties them to the caller's all-to-alls), and only the calls that must return
    // counts to the host wait.
    hipStream_t ext_stream = nullptr;
    bool ext_side = false;
    hipEvent_t ev_slot[2] = {nullptr, nullptr}, ev_keep[2] = {nullptr, nullptr}, ev_ans = nullptr;
    hipStream_t side() const { return ext_side ? ext_stream : stream2; }
    // Three kinds of kernels append states at arena_next (the chunk's locally owned new states, keep, ingest) and they run on
    // different streams: one event chains them, each appender waits for the previous one and records when it is enqueued.
    hipEvent_t ev_append = nullptr;
    void append_begin(hipStream_t s) { if (ev_append) hipStreamWaitEvent(s, ev_append, 0); }
    void append_end(hipStream_t s) {
        if (!ev_append) hipEventCreateWithFlags(&ev_append, hipEventDisableTiming);
        hipEventRecord(ev_append, s);
    }
    int side_done() {
        if (!ext_side) HIP_TRY(hipStreamSynchronize(stream2));
        return MC_OK;
    }
    int shard_set_stream(void *hip_stream, int enable) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        ext_stream = (hipStream_t)hip_stream;
        ext_side = enable != 0;
        return MC_OK;
    }
    int shard_expand_launch(unsigned slot, uint64_t first, uint64_t count, uint64_t send_cap) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) { set_error("shard_expand: slot must be 0 or 1"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        const unsigned P = nranks();
        int rc = shard_alloc(q, send_cap);
        if (rc) return rc;
        q.pend_total = 0;
        q.count = count;
        q.launched = true;
        for (unsigned t = 0; t <= 8; t++) q.pend_off.off[t] = 0;
        if (first + count > sh_hi - sh_lo) { set_error("shard_expand: chunk outside the local frontier"); return MC_EBADCFG; }
        if (count > chunk) { set_error("shard_expand: chunk larger than chunk_states"); return MC_EBADCFG; }
        if (!count) return MC_OK;
        if (!ev_slot[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_slot[slot], hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(stream, ev_slot[slot], 0));  // the slot's previous buckets have been compacted
        if (ev_mat[slot]) HIP_TRY(hipStreamWaitEvent(stream, ev_mat[slot], 0));  // ... and its locally owned new states written
        HIP_TRY(hipMemsetAsync(q.rt_cur, 0, (size_t)(P * NSHARD) * sizeof(PaddedCounter), stream));
        const uint64_t c0 = sh_lo + first, c1 = c0 + count, base = c0 & ~63ull;
        const uint64_t ncols = ((c1 - base) + 63) & ~63ull;
        q.chunk_base = base;
        RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
        rt.my_rank = cfg.shard_rank;
        rt.new_fp = d_newfp;
        q.ncols = ncols;
        timed(0, count, [&] {  // new-list parity = slot: the locally owned new states of this chunk (local-owner shortcut)
            launch_expand<S, true>(!(cfg.flags & MC_F_NOFAMILY), cfg.flags, ncols, stream, 0u, prm,
                                   (const uint64_t *)d_arena, c0, c1, ncols, d_table, seen_arg(), d_newlist, seg_cap, d_ctr, cfg.flags, rt, slot);
        });
        return MC_OK;
    }
    int shard_expand_finish(unsigned slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1 || !sl[slot].launched) { set_error("shard_expand_finish: no expand in flight for this slot"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        q.launched = false;
        const unsigned P = nranks();
        for (unsigned t = 0; t < P; t++) send_counts[t] = 0;
        if (!q.count) return MC_OK;
        // one wait on the expand stream only: the side stream may still be busy with the previous round
        PaddedCounter *cur = h_cur;
        HIP_TRY(hipMemcpyAsync(cur, q.rt_cur, (size_t)(P * NSHARD) * sizeof(PaddedCounter), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof(DevCounters), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int rc;
        if ((rc = check_dev_error())) return rc;
        uint64_t total = 0;
        for (unsigned t = 0; t < P; t++) {
            q.pend_off.off[t] = total;
            for (unsigned x = 0; x < NSHARD; x++) {
                if (cur[t * NSHARD + x].v > q.rt_subcap) { set_error("shard_expand: route bucket overflow (raise send_cap)"); return MC_EROUTE; }
                send_counts[t] += cur[t * NSHARD + x].v;
            }
            total += send_counts[t];
        }
        for (unsigned t = P; t <= 8; t++) q.pend_off.off[t] = total;
        // (the pending list is as long as the largest send capacity asked for so far: more candidates than it holds is "more successors
        //  per state than the fan-out allowance" — MC_EROUTE, the level loop starts over with twice the allowance — not a full arena)
        if (total >= (1ull << 31)) { set_error("shard_expand: more than 2^31 candidates in one round (lower chunk_states)"); return MC_EARENA; }
        if (total > q.pend_cap) { set_error("shard_expand: more candidates in one round than the slot's pending list holds (raise the fan-out allowance)"); return MC_EROUTE; }
        if (total > send_cap) { set_error("shard_expand: send buffer too small (raise the fan-out allowance)"); return MC_EROUTE; }
        q.pend_total = total;
        if (q.keep_pending) {  // the slot's previous keep still reads pend_src
            HIP_TRY(hipStreamWaitEvent(side(), ev_keep[slot], 0));
            q.keep_pending = false;
        }
        {   // the chunk's locally owned new states (local-owner shortcut): materialised on the side stream, in order with the
            // other kernels that append at arena_next (keep, ingest); the expand kernel has finished (stream was synchronised)
            const unsigned bx = (unsigned)((q.ncols + 255) / 256);
            const unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
            append_begin(stream2);
            timed(2, 0, [&] {
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm ? gm : 1, NSHARD), dim3(256), 0, stream2, prm, d_arena, q.chunk_base, d_newlist, seg_cap,
                                   arena_cap, d_parent, d_pslot, d_ctr, slot, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
            }, stream2);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, slot);
            append_end(stream2);
            if (!ev_mat[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_mat[slot], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_mat[slot], stream2));
        }
        if (total) {
            RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
            hipLaunchKernelGGL(k_compact_buckets, dim3(64, P * NSHARD), dim3(256), 0, side(), rt, send_fp, q.pend_src);
        }
        HIP_TRY(hipEventRecord(ev_slot[slot], side()));
        return side_done();
    }
    int shard_probe(const uint64_t *recv_fp, uint64_t n, uint8_t *answers) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n) return MC_OK;
        timed(1, n, [&] {
            hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), recv_fp, n, d_table, seen_arg(), answers, d_ctr);
        }, side());
        return side_done();
    }
    // ---- fixed-capacity rounds: nothing of a round waits for the host (sizes travel in band; errors surface at shard_end_level)
    hipEvent_t ev_exp[2] = {nullptr, nullptr};
    // the slot's new-list segment and its counters (parity = slot) are read by the local materialise + commit on the second
    // stream: the NEXT expand into the same slot waits for them (a host that never blocks enqueues expands back to back)
    hipEvent_t ev_mat[2] = {nullptr, nullptr};
    int shard_expand_pack(unsigned slot, uint64_t *send_fp, uint64_t cap) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1 || !sl[slot].launched) { set_error("shard_expand_pack: no expand in flight for this slot"); return MC_EBADCFG; }
        ShSlot &q = sl[slot];
        q.launched = false;
        const unsigned P = nranks();
        if (cap < 2 || (uint64_t)P * cap > q.pend_cap || (uint64_t)P * cap >= (1ull << 31)) { set_error("shard_expand_pack: capacity does not fit the slot's buffers"); return MC_EBADCFG; }
        q.pend_total = (uint64_t)P * cap;  // the sender scans the whole packed range: answers outside the counts are 0
        for (unsigned t = 0; t <= 8; t++) q.pend_off.off[t] = (t < P ? t : P) * cap;
        if (!ev_exp[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_exp[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_exp[slot], stream));  // behind the slot's expand kernel (or behind nothing: an empty chunk)
        // The buckets are compacted on the EXPAND stream, right behind the kernel that filled them: on the side stream the
        // compaction of round r+1 would queue behind the probes of round r (or those behind it), and the fingerprint exchange of
        // round r+1 could never overlap them.  The caller orders its collective behind this call with an event on that stream
        // (mc_shard_info hands it out).
        if (q.keep_pending) {  // the slot's previous keep still reads pend_src
            HIP_TRY(hipStreamWaitEvent(stream, ev_keep[slot], 0));
            q.keep_pending = false;
        }
        if (q.count) {  // the chunk's locally owned new states (local-owner shortcut), as in shard_expand_finish
            const unsigned bx = (unsigned)((q.ncols + 255) / 256);
            const unsigned gm = bx < 8 * 256 ? (bx + 7) / 8 : 256;
            HIP_TRY(hipStreamWaitEvent(stream2, ev_exp[slot], 0));
            append_begin(stream2);
            timed(2, 0, [&] {
                hipLaunchKernelGGL(k_materialise<S>, dim3(gm ? gm : 1, NSHARD), dim3(256), 0, stream2, prm, d_arena, q.chunk_base, d_newlist, seg_cap,
                                   arena_cap, d_parent, d_pslot, d_ctr, slot, (const LevelCtl *)nullptr, (const uint64_t *)d_newfp);
            }, stream2);
            hipLaunchKernelGGL(k_commit, dim3(1), dim3(1), 0, stream2, d_ctr, slot);
            append_end(stream2);
            if (!ev_mat[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_mat[slot], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_mat[slot], stream2));
            RouteArgs rt{P, q.rt_cur, q.rt_fp, q.rt_src, q.rt_subcap};
            hipLaunchKernelGGL(k_compact_packed, dim3(64, P * NSHARD), dim3(256), 0, stream, rt, cap, send_fp, q.pend_src, d_ctr);
        } else {  // no chunk for this rank in this round: empty buckets
            for (unsigned t = 0; t < P; t++) HIP_TRY(hipMemsetAsync(send_fp + (uint64_t)t * cap, 0, sizeof(uint64_t), stream));
        }
        if (!ev_slot[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_slot[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_slot[slot], stream));
        if (!ext_side) HIP_TRY(hipStreamSynchronize(stream));
        return side_done();
    }
    int shard_probe_pack(const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) override {
        HIP_TRY(hipSetDevice(cfg.device));
        const uint64_t total = (uint64_t)nranks() * cap;
        if (!total) return MC_OK;
        timed(1, total, [&] {
            hipLaunchKernelGGL(k_probe_packed, dim3((unsigned)(nranks() * ((cap + 255) / 256))), dim3(256), 0, side(), recv_fp, cap, nranks(), d_table,
                               seen_arg(), answers, d_ctr);
        }, side());
        return side_done();
    }
    int shard_keep_pack(unsigned slot, const uint8_t *answers_back, uint64_t cap) override {
        if (slot > 1 || sl[slot].pend_total != (uint64_t)nranks() * cap) { set_error("shard_keep_pack: not the capacity the slot was packed with"); return MC_EBADCFG; }
        return shard_keep(slot, answers_back, nullptr);
    }
    // the caller's stream waits (on the device) until the slot's last keep has consumed its answers buffer: a caller that
    // reuses ONE answers buffer per slot calls this before the collective that overwrites it
    int shard_wait_keep(unsigned slot) override {
        if (slot > 1) return MC_EBADCFG;
        HIP_TRY(hipSetDevice(cfg.device));
        if (sl[slot].keep_pending && ev_keep[slot]) HIP_TRY(hipStreamWaitEvent(side(), ev_keep[slot], 0));
        return MC_OK;
    }
    int shard_materialise(unsigned slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        const unsigned P = nranks();
        for (unsigned t = 0; t < P; t++) send_counts[t] = 0;
        q.moved = 0;
        if (!q.pend_total) return MC_OK;
        const unsigned bx = (unsigned)((q.pend_total + 255) / 256);
        int rc = scan_answers(answers_back, q.pend_total, side());
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_range_ends, dim3(1), dim3(64), 0, side(), d_incl, q.pend_off, P, d_ends);
        unsigned long long *ends = h_ends;  // the per-owner counts go back to the host: the caller sizes its all-to-all with them
        HIP_TRY(hipMemcpyAsync(ends, d_ends, P * sizeof(unsigned long long), hipMemcpyDeviceToHost, side()));
        HIP_TRY(hipStreamSynchronize(side()));
        OwnerOffsets start;
        std::vector<PaddedCounter> cnt(P);
        for (unsigned t = 0; t <= 8; t++) start.off[t] = 0;
        for (unsigned t = 0; t < P; t++) {
            start.off[t] = t ? ends[t - 1] : 0;
            cnt[t].v = ends[t] - start.off[t];
        }
        hipLaunchKernelGGL(k_compact_new, dim3(bx), dim3(256), 0, side(), answers_back, d_incl, q.pend_src, q.pend_total, q.pend_off, start, P,
                           d_new_src);
        BlockPlan plan;
        uint64_t blocks = 0;
        for (unsigned t = 0; t < 8; t++) {
            plan.blk_off[t] = blocks;
            plan.cnt[t] = t < P ? cnt[t].v : 0;
            if (t < P) { send_counts[t] = cnt[t].v; blocks += (cnt[t].v + 63) / 64; }
        }
        plan.blk_off[8] = blocks;
        for (unsigned t = P; t < 8; t++) plan.blk_off[t] = blocks;
        q.plan = plan;
        q.moved = 0;
        for (unsigned t = 0; t < P; t++) q.moved += plan.cnt[t];
        if (blocks * 64 > send_cap) { set_error("shard_materialise: state send buffer too small"); return MC_EARENA; }  // (the loop sizes it for the upper bound and repeats)
        if (blocks) {
            timed(2, blocks * 64, [&] {
                hipLaunchKernelGGL(k_send_materialise<S>, dim3((unsigned)((blocks * 64 + 255) / 256)), dim3(256), 0, side(), prm, d_arena,
                                   q.chunk_base, d_new_src, q.pend_off, plan, P, (uint64_t *)send_states);
            }, side());
        }
        return side_done();
    }
    // "stay" mode: materialise the positively answered candidates of slot's expand into the LOCAL arena.
    // The count of new states stays on the device (arena_next); the host learns it at shard_end_level.
    int shard_keep(unsigned slot, const uint8_t *answers_back, uint64_t *n_new) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (n_new) *n_new = 0;
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        if (!q.pend_total) return MC_OK;
        const unsigned bx = (unsigned)((q.pend_total + 255) / 256);
        // keep runs on the engine's own second stream, behind the caller's stream at this point (the answers
        // are ready there), so the caller's next exchange does not queue behind the materialisation
        hipStream_t ks = stream2;
        if (ext_side) {
            if (!ev_ans) HIP_TRY(hipEventCreateWithFlags(&ev_ans, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(ev_ans, ext_stream));
            HIP_TRY(hipStreamWaitEvent(ks, ev_ans, 0));
        }
        int rc = scan_answers(answers_back, q.pend_total, ks);
        if (rc) return rc;
        append_begin(ks);
        const uint32_t *d_total = d_incl + (q.pend_total - 1);
        OwnerOffsets one, zero;  // a single range covering every pending candidate
        for (unsigned t = 0; t <= 8; t++) { one.off[t] = t ? q.pend_total : 0; zero.off[t] = 0; }
        hipLaunchKernelGGL(k_compact_new, dim3(bx), dim3(256), 0, ks, answers_back, d_incl, q.pend_src, q.pend_total, one, zero, 1u, d_new_src);
        // grid sized for the upper bound (every candidate new); surplus threads leave at once
        timed(2, 0, [&] {
            hipLaunchKernelGGL(k_materialise_list<S>, dim3(bx), dim3(256), 0, ks, prm, d_arena, q.chunk_base, d_new_src, d_total, arena_cap,
                               d_parent, d_pslot, d_ctr);
        }, ks);
        hipLaunchKernelGGL(k_bump_arena_next, dim3(1), dim3(1), 0, ks, d_ctr, d_total, 0ull, (unsigned long long)arena_cap);
        append_end(ks);
        if (!ev_keep[slot]) HIP_TRY(hipEventCreateWithFlags(&ev_keep[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ev_keep[slot], ks));
        q.keep_pending = true;
        if (!ext_side) {
            uint32_t total32 = 0;
            HIP_TRY(hipMemcpyAsync(&total32, d_total, sizeof total32, hipMemcpyDeviceToHost, ks));
            HIP_TRY(hipStreamSynchronize(ks));
            if (n_new) *n_new = total32;
        }
        return MC_OK;
    }
    // recv_states: one bucket per source rank, back to back, each a whole number of 64-state blocks;
    // n = valid states of ONE bucket starting at recv_states (call once per source)
    int shard_ingest(const uint8_t *recv_states, uint64_t n) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n) return MC_OK;
        append_begin(side());
        hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), d_arena, W, (const uint64_t *)recv_states, n,
                           arena_cap, d_parent, d_ctr);
        hipLaunchKernelGGL(k_bump_arena_next, dim3(1), dim3(1), 0, side(), d_ctr, (const uint32_t *)nullptr, (unsigned long long)n,
                           (unsigned long long)arena_cap);
        append_end(side());
        return side_done();
    }
    // (parent index on this rank << 16 | slot) of the states the slot's last shard_materialise put into its send buffer, same
    // order (owner by owner, without the block padding); call it before the next shard_materialise / shard_keep of any slot
    int shard_materialise_parents(unsigned slot, uint64_t *send_parents) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (slot > 1) return MC_EBADCFG;
        ShSlot &q = sl[slot];
        if (!q.moved) return MC_OK;
        hipLaunchKernelGGL(k_send_parents, dim3((unsigned)((q.moved + 255) / 256)), dim3(256), 0, side(), (const uint32_t *)d_new_src, q.chunk_base,
                           q.pend_off, q.plan, nranks(), send_parents);
        return side_done();
    }
    // right after the shard_ingest of the same bucket: the n states just appended came from `src_rank`
    int shard_ingest_parents(const uint64_t *recv_parents, uint64_t n, unsigned src_rank) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (!n || !d_prank) return MC_OK;  // without MC_F_TRACE nothing is recorded
        hipLaunchKernelGGL(k_ingest_parents, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, side(), recv_parents, n, src_rank, d_parent, d_pslot,
                           d_prank, (const DevCounters *)d_ctr);
        return side_done();
    }
    int shard_violation(int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        int rc = read_counters();
        if (rc) return rc;
        *found = h_ctr->viol_key != ~0ull;
        *idx = 0; *slot = 0; *verdict = MC_V_OK; *invariant = -1;
        if (*found) {
            const unsigned long long k = h_ctr->viol_key;
            const unsigned kind = (unsigned)(k & 7u);
            *idx = viol_idx(k);
            *slot = (uint32_t)(k >> 8 & 0xffffu);
            *verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
            if (kind == VK_INVARIANT) *invariant = (int32_t)(k >> 3 & 31u);
        }
        return MC_OK;
    }
    // one step of a counterexample walk: the state at arena index idx of THIS rank and where its parent lives
    // (parent_rank == shard_rank: here; parent_idx == 0xffffffff: an initial state; parent_slot 0xfffc: the entry is a copy
    // of state parent_idx made by the replicated prefix, not a step)
    int shard_fetch(uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (idx & (1ull << 63)) {  // an initial state, rebuilt from its ordinal in Init's enumeration (a violating initial state is not looked up in the arena)
            const uint64_t ord = idx & ~(1ull << 63);
            if (ord >= S::num_init(prm)) return MC_EBADCFG;
            S::init(prm, ord, WordRef{(uint64_t *)state_out, 1});
            *parent_rank = cfg.shard_rank;
            *parent_idx = 0xffffffffull;
            *parent_slot = SLOT_INIT;
            return MC_OK;
        }
        if (!d_parent) { set_error("engine created without MC_F_TRACE"); return MC_ESTATE; }
        if (idx >= arena_cap) return MC_EBADCFG;
        HIP_TRY(hipDeviceSynchronize());
        int rc = fetch_state(idx, (uint64_t *)state_out);
        if (rc) return rc;
        uint32_t p = 0;
        uint16_t ps = 0;
        uint8_t pr = 0xff;
        HIP_TRY(hipMemcpy(&p, d_parent + idx, sizeof p, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&ps, d_pslot + idx, sizeof ps, hipMemcpyDeviceToHost));
        if (d_prank) HIP_TRY(hipMemcpy(&pr, d_prank + idx, sizeof pr, hipMemcpyDeviceToHost));
        *parent_rank = pr == 0xff ? cfg.shard_rank : pr;
        *parent_idx = p;
        *parent_slot = ps;
        return MC_OK;
    }
    int shard_end_level(uint64_t *new_local) override {
        HIP_TRY(hipSetDevice(cfg.device));
        if (sl[0].launched || sl[1].launched) { set_error("shard_end_level: an expand is still in flight"); return MC_EBADCFG; }
        HIP_TRY(hipStreamSynchronize(side()));
        HIP_TRY(hipStreamSynchronize(stream2));
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        sh_next = h_ctr->arena_next;
        last_distinct = sh_next;
        sh_lo = sh_hi;
        sh_hi = sh_next;
        *new_local = sh_hi - sh_lo;
        return MC_OK;
    }
    // invariants of the unexpanded frontier [lo, hi) for specs that check on expansion (see k_check_frontier); no-op otherwise
    int check_frontier(uint64_t lo, uint64_t hi) {
        if constexpr (ChecksOnExpand<S>::value) {
            if (hi > lo) {
                hipLaunchKernelGGL(k_check_frontier<S>, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, stream, prm,
                                   (const uint64_t *)d_arena, lo, hi, d_ctr);
                return read_counters();
            }
        }
        return MC_OK;
    }
    int shard_check_frontier() override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        HIP_TRY(hipStreamSynchronize(stream2));
        return check_frontier(sh_lo, sh_hi);
    }
    int shard_counters(uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) override {
        HIP_TRY(hipSetDevice(cfg.device));
        HIP_TRY(hipStreamSynchronize(side()));
        int rc = read_counters();
        if (rc) return rc;
        if ((rc = check_dev_error())) return rc;
        *generated = 0;
        for (int t = 0; t < NSHARD; t++) *generated += h_ctr->generated[t].v;
        *distinct_local = h_ctr->arena_next - sh_dup;
        *verdict = MC_V_OK;
        if (h_ctr->viol_key != ~0ull) {
            const unsigned kind = (unsigned)(h_ctr->viol_key & 7u);
            *verdict = kind == VK_INVARIANT ? MC_V_INVARIANT : kind == VK_ASSERT ? MC_V_ASSERT : kind == VK_DEADLOCK ? MC_V_DEADLOCK : MC_V_SPECERR;
        }
        return MC_OK;
    }
    int kernel_stats(mc_kernel_stats *o) override {
        o->expand = kstat[0];
        o->insert = kstat[1];
        o->materialise = kstat[2];
        o->state_bytes = (uint64_t)W * 8;
        o->cand_cells = kstat_cells;
        o->inwave_states = kstat_inwave;
        return MC_OK;
    }
};

}  // namespace mc

// --------------------------------------------------------------------------------------- engine factories
// group 1: atomic_add + pcal_intro, 2: raft (2 servers), 3: raft (3), 4: raft (5), 5: serializableSnapshotIsolation,
// 6: compiled PlusCal, 7: Voting / Paxos
namespace mc {
static int spec_group(const mc_spec_desc *d) {
    if (!d) return 0;
    switch (d->spec_id) {
    case MC_SPEC_ATOMIC_ADD: case MC_SPEC_PCAL_INTRO: return 1;
    case MC_SPEC_RAFT: return d->nparams < 1 ? 0 : d->params[0] == 2 ? 2 : d->params[0] == 3 ? 3 : d->params[0] == 5 ? 4 : 0;
    case MC_SPEC_SSI: return 5;
    case MC_SPEC_PCAL: return 6;
    case MC_SPEC_PAXOS: return 7;
    default: return 0;
    }
}
template <class S>
static int make_engine(const typename S::Params &prm, const mc_spec_desc *spec, const mc_config *cfg, EngineBase **out) {
    auto *e = new Engine<S>();
    e->prm = prm;
    e->desc = *spec;
    e->cfg = *cfg;
    const int r = e->alloc();
    if (r) { delete e; return r; }
    *out = e;
    return MC_OK;
}
}  // namespace mc
extern "C" {
int mc_make_engine_1(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_2(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_3(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_4(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_5(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_6(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
int mc_make_engine_7(const mc_spec_desc *, const mc_config *, mc::EngineBase **);
#if MC_TU == 1 || MC_TU == -1
int mc_make_engine_1(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    if (d->spec_id == MC_SPEC_ATOMIC_ADD) {
        mc::SpecAtomicAdd::Params p;
        if (mc::SpecAtomicAdd::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
        return mc::make_engine<mc::SpecAtomicAdd>(p, d, c, out);
    }
    mc::SpecPcalIntro::Params p;
    if (mc::SpecPcalIntro::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecPcalIntro>(p, d, c, out);
}
#endif
#define MC_RAFT_FACTORY(K, SPEC)                                                                   \
    int mc_make_engine_##K(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {      \
        mc::RaftParams p;                                                                          \
        if (mc::SPEC::make_params(d->params, d->nparams, p)) return MC_EBADCFG;                    \
        return mc::make_engine<mc::SPEC>(p, d, c, out);                                            \
    }
#if MC_TU == 2 || MC_TU == -1
MC_RAFT_FACTORY(2, SpecRaft2)
#endif
#if MC_TU == 3 || MC_TU == -1
MC_RAFT_FACTORY(3, SpecRaft3)
#endif
#if MC_TU == 4 || MC_TU == -1
MC_RAFT_FACTORY(4, SpecRaft5)
#endif
#if MC_TU == 5 || MC_TU == -1
int mc_make_engine_5(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::SsiParams p;
    if (mc::SpecSsi::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecSsi>(p, d, c, out);
}
#endif
#if MC_TU == 7 || MC_TU == -1
// Voting / Paxos (spec_paxos.h)
int mc_make_engine_7(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::PaxosParams p;
    if (mc::SpecPaxos::make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    return mc::make_engine<mc::SpecPaxos>(p, d, c, out);
}
#endif
#if MC_TU == 6 || MC_TU == -1
// compiled PlusCal (spec_vm.h): the program image is copied to the device, the kernels read it through prm.code
int mc_make_engine_6(const mc_spec_desc *d, const mc_config *c, mc::EngineBase **out) {
    mc::VmParams p;
    if (mc::vm_make_params(d->params, d->nparams, p)) return MC_EBADCFG;
    if (hipSetDevice(c->device) != hipSuccess) { mc::set_error("hipSetDevice failed"); return MC_EHIP; }
    int32_t *d_code = nullptr;
    if (hipMalloc(&d_code, (size_t)p.code_len * sizeof(int32_t)) != hipSuccess ||
        hipMemcpy(d_code, p.code, (size_t)p.code_len * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) {
        mc::set_error("cannot upload the PlusCal program image");
        return MC_EHIP;
    }
    // identity of the compiled program for checkpoints: the whole image (header, tables, code) and the scalars derived from it
    uint64_t ph = 0xcbf29ce484222325ull;
    auto mixin = [&](uint64_t v) { ph = (ph ^ v) * 0x100000001b3ull; ph ^= ph >> 29; };
    for (int i = 0; i < p.code_len; i++) mixin((uint32_t)p.code[i]);
    for (int v : {p.nv, p.words, p.ninst, p.maxch, p.pc_base, p.done, p.init_entry, p.ninv, p.ncon, p.label_tab, p.self_tab, p.code_len}) mixin((uint32_t)v);
    for (int i = 0; i < 8; i++) mixin((uint32_t)p.inv_entry[i]);
    mixin(p.num_init);
    p.code = d_code;  // host-side helpers (format, action_of) use p.host only
    const int rc = p.nv <= 16 ? mc::make_engine<mc::SpecVm16>(p, d, c, out)
                 : p.nv <= 32 ? mc::make_engine<mc::SpecVm32>(p, d, c, out)
                 : p.nv <= 64 ? mc::make_engine<mc::SpecVm64>(p, d, c, out) : mc::make_engine<mc::SpecVm>(p, d, c, out);
    if (rc) hipFree(d_code);
    else { (*out)->owned_device_blob = d_code; (*out)->program_hash = ph ? ph : 1; }
    return rc;
}
#endif
}  // extern "C"

#if MC_TU == 0 || MC_TU == -1
static thread_local std::string g_last_error;
// --------------------------------------------------------------------------------------- C ABI
using namespace mc;

struct mc_engine {
    EngineBase *impl;
};

extern "C" {

int mc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mc_engine_create(const mc_spec_desc *spec, const mc_config *cfg, mc_engine **out) {
    if (!spec || !cfg || !out) return MC_EBADCFG;
    *out = nullptr;
    if (mc_device_count() <= 0) { set_error("no HIP device: libtlamc has no CPU fallback"); return MC_EHIP; }
    g_last_error.clear();
    EngineBase *impl = nullptr;
    const int group = spec_group(spec);
    int rc = MC_EBADCFG;
    switch (group) {
    case 1: rc = mc_make_engine_1(spec, cfg, &impl); break;
    case 2: rc = mc_make_engine_2(spec, cfg, &impl); break;
    case 3: rc = mc_make_engine_3(spec, cfg, &impl); break;
    case 4: rc = mc_make_engine_4(spec, cfg, &impl); break;
    case 5: rc = mc_make_engine_5(spec, cfg, &impl); break;
    case 6: rc = mc_make_engine_6(spec, cfg, &impl); break;
    case 7: rc = mc_make_engine_7(spec, cfg, &impl); break;
    default: break;
    }
    if (rc == MC_OK) *out = new mc_engine{impl};
    else if (rc == MC_EBADCFG && g_last_error.empty()) set_error("unknown spec id or constants out of range");
    return rc;
}
int mc_engine_run(mc_engine *e, mc_result *out) { return e && out ? e->impl->run(out) : MC_EBADCFG; }
int mc_engine_step(mc_engine *e, uint32_t levels, mc_result *out) { return e && out ? e->impl->step(levels, out) : MC_EBADCFG; }
int mc_engine_set_progress(mc_engine *e, mc_progress_fn fn, void *user, double min_interval_seconds) {
    if (!e) return MC_EBADCFG;
    e->impl->progress_fn = fn;
    e->impl->progress_user = user;
    e->impl->progress_interval = min_interval_seconds > 0 ? min_interval_seconds : 0.0;
    return MC_OK;
}
int mc_engine_trace(mc_engine *e, uint8_t *states_out, int32_t *actions_out, size_t *n_inout) {
    return e && n_inout ? e->impl->trace(states_out, actions_out, n_inout) : MC_EBADCFG;
}
int mc_engine_kernel_stats(mc_engine *e, mc_kernel_stats *out) { return e && out ? e->impl->kernel_stats(out) : MC_EBADCFG; }
int mc_engine_read_states(mc_engine *e, uint64_t first, uint64_t count, uint8_t *out) {
    return e && (out || !count) ? e->impl->read_states(first, count, out) : MC_EBADCFG;
}
int mc_engine_checkpoint(mc_engine *e, const char *path) { return e && path ? e->impl->checkpoint(path) : MC_EBADCFG; }
int mc_engine_restore(mc_engine *e, const char *path) { return e && path ? e->impl->restore(path) : MC_EBADCFG; }
void mc_engine_destroy(mc_engine *e) {
    if (!e) return;
    delete e->impl;
    delete e;
}

size_t mc_state_bytes(const mc_spec_desc *spec) {
    size_t n = 0;
    dispatch_spec(spec, [&](auto s, const auto &prm) { n = sizeof(uint64_t) * decltype(s)::words(prm); return 0; });
    return n;
}
uint32_t mc_fp_owner(uint64_t fp, uint32_t shard_count) { return fp_owner(fp, shard_count); }
int mc_state_format(const mc_spec_desc *spec, const uint8_t *state, char *buf, size_t cap) {
    if (!state || !buf || !cap) return MC_EBADCFG;
    int n = MC_EBADCFG;
    int rc = dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        n = S::format(prm, w, buf, cap);
        if ((size_t)n < cap) buf[n] = 0; else buf[cap - 1] = 0;
        return 0;
    });
    return rc ? rc : n;
}
// host-side evaluation of one (state, slot) pair: the action it is (mc_action_name's argument) and its successor
int mc_state_action(const mc_spec_desc *spec, const uint8_t *state, int32_t slot) {
    if (!state) return MC_EBADCFG;
    int a = -1;
    int rc = dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        a = S::action_of(prm, w, (int)slot);
        return 0;
    });
    return rc ? rc : a;
}
int mc_state_apply(const mc_spec_desc *spec, const uint8_t *state, int32_t slot, uint8_t *successor_out) {
    if (!state || !successor_out) return MC_EBADCFG;
    return dispatch_spec(spec, [&](auto s, const auto &prm) {
        using S = decltype(s);
        uint64_t w[S::MAX_WORDS], o[S::MAX_WORDS];
        memcpy(w, state, sizeof(uint64_t) * S::words(prm));
        S::apply(prm, CWordRef{w, 1}, (int)slot, WordRef{o, 1});
        memcpy(successor_out, o, sizeof(uint64_t) * S::words(prm));
        return 0;
    });
}
const char *mc_action_name(const mc_spec_desc *spec, int32_t action) {
    const char *nm = "?";
    dispatch_spec(spec, [&](auto s, const auto &prm) {
        if constexpr (std::is_same_v<std::decay_t<decltype(prm)>, VmParams>) nm = vm_action_name(prm.host, action);  // label names live in the program
        else nm = decltype(s)::action_name(action);
        return 0;
    });
    return nm;
}
const char *mc_strerror(int code) {
    switch (code) {
    case MC_OK: return "ok";
    case MC_EBADCFG: return "bad spec or configuration";
    case MC_EHIP: return "HIP error / no device";
    case MC_EOVERFLOW: return "packed-state slot array overflow";
    case MC_ETABLEFULL: return "seen-set full";
    case MC_EARENA: return "state arena full";
    case MC_ERCCL: return "exchange failure";
    case MC_ESTATE: return "call sequence error";
    case MC_EPARSE: return "parse error";
    case MC_ENOSPEC: return "module is not a lowered spec";
    case MC_EROUTE: return "exchange bucket full";
    default: return "unknown error";
    }
}
const char *mc_last_error(void) { return g_last_error.c_str(); }
void mc_set_error_internal(const char *msg) { g_last_error = msg ? msg : ""; }


int mc_engine_debug_reexpand(mc_engine *e, unsigned extra_flags, double *ms) { return e && ms ? e->impl->debug_reexpand(extra_flags, ms) : MC_EBADCFG; }
// profiling builds only (-DMC_PHASE_PROF): cycles per phase of k_expand_family summed over all wavefronts since the last reset
int mc_engine_debug_phases(mc_engine *e, uint64_t *out48, int reset) { return e && out48 ? e->impl->debug_phases(out48, reset) : MC_EBADCFG; }
// ---- sharded (multi-GPU) step API
int mc_shard_begin(mc_engine *e) { return e ? e->impl->shard_begin() : MC_EBADCFG; }
int mc_shard_begin_replicated(mc_engine *e, uint64_t min_frontier, uint64_t max_distinct, uint64_t max_levels, uint64_t *levels_out,
                              uint32_t *nlevels) {
    return e && levels_out && nlevels ? e->impl->shard_begin_replicated(min_frontier, max_distinct, max_levels, levels_out, nlevels) : MC_EBADCFG;
}
int mc_shard_level_size(mc_engine *e, uint64_t *n) { return e && n ? e->impl->shard_level_size(n) : MC_EBADCFG; }
int mc_shard_info(mc_engine *e, void **main_stream_out, uint64_t *chunk_states_out, int32_t *traced_out) {
    return e ? e->impl->shard_info(main_stream_out, chunk_states_out, traced_out) : MC_EBADCFG;
}
size_t mc_engine_state_bytes_internal(mc_engine *e) { return e ? e->impl->state_bytes() : 0; }
int mc_shard_expand(mc_engine *e, uint64_t first, uint64_t count, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) {
    if (!e || !send_counts) return MC_EBADCFG;
    const int rc = e->impl->shard_expand_launch(0, first, count, send_cap);
    return rc ? rc : e->impl->shard_expand_finish(0, send_fp, send_cap, send_counts);
}
int mc_shard_set_stream(mc_engine *e, void *hip_stream, int enable) { return e ? e->impl->shard_set_stream(hip_stream, enable) : MC_EBADCFG; }
int mc_shard_expand_launch(mc_engine *e, uint32_t slot, uint64_t first, uint64_t count, uint64_t send_cap) {
    return e ? e->impl->shard_expand_launch(slot, first, count, send_cap) : MC_EBADCFG;
}
int mc_shard_expand_finish(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t send_cap, uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_expand_finish(slot, send_fp, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_probe(mc_engine *e, const uint64_t *recv_fp, uint64_t n, uint8_t *answers) {
    return e ? e->impl->shard_probe(recv_fp, n, answers) : MC_EBADCFG;
}
int mc_shard_materialise(mc_engine *e, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap, uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_materialise(0, answers_back, send_states, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_materialise_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint8_t *send_states, uint64_t send_cap,
                              uint64_t *send_counts) {
    return e && send_counts ? e->impl->shard_materialise(slot, answers_back, send_states, send_cap, send_counts) : MC_EBADCFG;
}
int mc_shard_ingest(mc_engine *e, const uint8_t *recv_states, uint64_t n) { return e ? e->impl->shard_ingest(recv_states, n) : MC_EBADCFG; }
int mc_shard_expand_pack(mc_engine *e, uint32_t slot, uint64_t *send_fp, uint64_t cap) { return e && send_fp ? e->impl->shard_expand_pack(slot, send_fp, cap) : MC_EBADCFG; }
int mc_shard_probe_pack(mc_engine *e, const uint64_t *recv_fp, uint64_t cap, uint8_t *answers) {
    return e && recv_fp && answers ? e->impl->shard_probe_pack(recv_fp, cap, answers) : MC_EBADCFG;
}
int mc_shard_keep_pack(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t cap) {
    return e && answers_back ? e->impl->shard_keep_pack(slot, answers_back, cap) : MC_EBADCFG;
}
int mc_shard_wait_keep(mc_engine *e, uint32_t slot) { return e ? e->impl->shard_wait_keep(slot) : MC_EBADCFG; }
int mc_shard_keep(mc_engine *e, const uint8_t *answers_back, uint64_t *n_new) { return e && n_new ? e->impl->shard_keep(0, answers_back, n_new) : MC_EBADCFG; }
int mc_shard_keep_slot(mc_engine *e, uint32_t slot, const uint8_t *answers_back, uint64_t *n_new) {
    return e ? e->impl->shard_keep(slot, answers_back, n_new) : MC_EBADCFG;
}
int mc_shard_end_level(mc_engine *e, uint64_t *new_local) { return e && new_local ? e->impl->shard_end_level(new_local) : MC_EBADCFG; }
int mc_shard_counters(mc_engine *e, uint64_t *generated, uint64_t *distinct_local, int32_t *verdict) {
    return e && generated && distinct_local && verdict ? e->impl->shard_counters(generated, distinct_local, verdict) : MC_EBADCFG;
}
int mc_shard_check_frontier(mc_engine *e) { return e ? e->impl->shard_check_frontier() : MC_EBADCFG; }
int mc_shard_materialise_parents(mc_engine *e, uint32_t slot, uint64_t *send_parents) { return e ? e->impl->shard_materialise_parents(slot, send_parents) : MC_EBADCFG; }
int mc_shard_ingest_parents(mc_engine *e, const uint64_t *recv_parents, uint64_t n, uint32_t src_rank) {
    return e ? e->impl->shard_ingest_parents(recv_parents, n, src_rank) : MC_EBADCFG;
}
int mc_shard_violation(mc_engine *e, int32_t *found, uint64_t *idx, uint32_t *slot, int32_t *verdict, int32_t *invariant) {
    return e && found && idx && slot && verdict && invariant ? e->impl->shard_violation(found, idx, slot, verdict, invariant) : MC_EBADCFG;
}
int mc_shard_note_levels(mc_engine *e, const uint64_t *levels, uint32_t n, int32_t verdict) { return e && (levels || !n) ? e->impl->shard_note_levels(levels, n, verdict) : MC_EBADCFG; }
int mc_shard_resume(mc_engine *e, uint64_t *levels_out, uint32_t *nlevels) { return e && levels_out && nlevels ? e->impl->shard_resume(levels_out, nlevels) : MC_EBADCFG; }
int mc_shard_checkpoint(mc_engine *e, const char *path) { return e && path ? e->impl->shard_checkpoint(path) : MC_EBADCFG; }
int mc_shard_restore(mc_engine *e, const char *path) { return e && path ? e->impl->shard_restore(path) : MC_EBADCFG; }
int mc_shard_fetch(mc_engine *e, uint64_t idx, uint8_t *state_out, uint32_t *parent_rank, uint64_t *parent_idx, uint32_t *parent_slot) {
    return e && state_out && parent_rank && parent_idx && parent_slot ? e->impl->shard_fetch(idx, state_out, parent_rank, parent_idx, parent_slot) : MC_EBADCFG;
}

}  // extern "C"
#endif  // MC_TU == 0 || MC_TU == -1
