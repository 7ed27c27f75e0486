// engine_kernels.h — the DEVICE half of the BFS engine: every HIP kernel of tla_rust_amd/csrc/engine.hip, the device-resident counter
// block and the small templates that pick a kernel's form from what a spec defines.  Included by engine.hip only (inside namespace mc,
// after spec_registry.h).  It is its own file so that the stamp of a counter collection (bench.py kernel_source_hash,
// profiles/summarize_pmc.py) covers the kernels and nothing else: a fix in the host half — Engine, the C ABI, checkpoints — no longer
// invalidates rocprofv3 counters that were collected on unchanged kernels.
#ifndef TLAMC_ENGINE_KERNELS_H
#define TLAMC_ENGINE_KERNELS_H

namespace mc {

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
    // round 6: where the in-wave writers park the overflow of their survivor lists (65 new-list entries per chunk: 64 survivors + a
    // link to the wavefront's chunk before; k_expand_family's tail).  Same segments and parities as n_new, which stays 0 in such a level
    PaddedCounter n_side[2 * NSHARD];
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
template <class T>
__device__ __forceinline__ T *uniform_generic(T *p) {  // a wave-uniform pointer that arrived in vector registers (argument of an out-of-line function)
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (T *)(((uint64_t)hi << 32) | lo);
}
// ... and one that is known to point into LDS, as an address-space-3 pointer (ds_ instructions; a flat access to LDS is slower and the
// compiler's address-space inference does not see through a function argument)
template <class T>
__device__ __forceinline__ __attribute__((address_space(3))) T *uniform_lds(T *p) {
    typedef __attribute__((address_space(3))) T *LdsP;
    const uint32_t a = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LdsP)p);
    return (LdsP)(uintptr_t)a;
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

// Invariants of a STORED state (specs that check when a state is expanded): S::parent_status_step where a spec has one — the verdict of
// S::parent_status computed from what the state's last step can have changed (spec_ssi.h) — else S::parent_status
template <class S, class = void>
struct HasStepStatus : std::false_type {};
template <class S>
struct HasStepStatus<S, decltype((void)S::STEP_STATUS)> : std::true_type {};
template <class S, class Ref>
__device__ __forceinline__ unsigned stored_state_status(const typename S::Params &prm, const typename S::Local &loc, Ref s) {
    if constexpr (HasStepStatus<S>::value) return S::parent_status_step(prm, loc, s);
    else return S::parent_status(prm, loc, s);
}

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
// Round 6, MC_SEEN_ROTATE (default on; 0 = rounds 2-5's "first empty slot of the bucket"): a fingerprint's slots inside a bucket are
// tried in ROTATED order, starting at slot j0 = bits 32.. of the fingerprint (the bucket comes from the low 32 bits, the owner rank of a
// sharded run from the top 24).  Every prober follows the same sequence, so the linear-probing argument is unchanged — a fingerprint
// lives in the first slot of ITS sequence that was empty when it arrived; a reader checks the whole bucket it fetched for a match
// first — but now the slot a NEW fingerprint will take is known without reading the bucket, and it is empty with probability
// 1 - load (with "first empty slot" it is slot 0, empty only while the whole bucket is: e^(-4 load)).  BLIND = true uses that: one
// compare-and-swap at the home slot, unread — one trip to memory instead of two for a new state whose home slot is free, and for a
// known one that sits there; only when the slot holds another fingerprint does the bucket get read.  For a search whose candidates
// are mostly NEW (the SI models: 85 %) that halves the dependent round trips of an insert; where most candidates are duplicates
// (raft: 75 %) a blind compare-and-swap turns an L2-cacheable read into a memory-side atomic, so those kernels keep reading first.
#ifndef MC_SEEN_ROTATE
#define MC_SEEN_ROTATE 1
#endif
// PRE = true: the home bucket was read EARLIER (seen_load_home: the reader had other work between asking and needing it — the by-pairs
// kernel evaluates its next batch meanwhile); `pre` holds its SLOTS words and the first round of the loop uses them instead of loading.
template <int SLOTS>
__device__ __forceinline__ void seen_load_home(const uint64_t *table, uint64_t nbuckets, uint64_t fp, unsigned long long (&pre)[SLOTS]) {
    const uint64_t bk = ((fp & 0xffffffffull) * nbuckets) >> 32;
    const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + bk * SLOTS);
#pragma unroll
    for (int i = 0; i < SLOTS / 2; ++i) { const ulonglong2 v = line[i]; pre[2 * i] = v.x; pre[2 * i + 1] = v.y; }
}
template <int SLOTS, bool BLIND = false, bool PRE = false>
__device__ __forceinline__ bool seen_insert_t(uint64_t *table, uint64_t nbuckets, uint64_t fp, unsigned &err, const unsigned long long *pre = nullptr) {
    uint64_t bk = ((fp & 0xffffffffull) * nbuckets) >> 32;
#if MC_SEEN_ROTATE
    const unsigned j0 = (unsigned)(fp >> 32) & (unsigned)(SLOTS - 1);
    if constexpr (BLIND) {
        const unsigned long long cur = atomicCAS((unsigned long long *)&table[bk * SLOTS + j0], 0ull, (unsigned long long)fp);
        if (cur == 0) return true;
        if (cur == fp) return false;
    }
#else
    static_assert(!BLIND, "a blind first compare-and-swap needs the rotated slot order");
#endif
    for (int probe = 0; probe < 2048; ++probe) {
        const uint64_t b = bk * SLOTS;
        unsigned long long slot[SLOTS];
        if (PRE && probe == 0) {
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) slot[i] = pre[i];
        } else {
#if MC_NT_PROBE & 1
        const mc_ull2 *line = reinterpret_cast<const mc_ull2 *>(table + b);
#pragma unroll
        for (int i = 0; i < SLOTS / 2; ++i) { const mc_ull2 v = __builtin_nontemporal_load(line + i); slot[2 * i] = v.x; slot[2 * i + 1] = v.y; }
#else
        const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + b);
#pragma unroll
        for (int i = 0; i < SLOTS / 2; ++i) { const ulonglong2 v = line[i]; slot[2 * i] = v.x; slot[2 * i + 1] = v.y; }
#endif
        }
#if MC_SEEN_ROTATE
        unsigned zm = 0;   // bit i: slot i was empty when the bucket was read
        bool hit = false;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            hit |= slot[i] == fp;
            zm |= (slot[i] == 0 ? 1u : 0u) << i;
        }
        if (hit) return false;
        // the empty slots in this fingerprint's order: j0, j0 + 1, ... (mod SLOTS)
        unsigned rot = ((zm >> j0) | (zm << (SLOTS - j0))) & ((1u << SLOTS) - 1u);
        while (rot) {
            const unsigned i = ((unsigned)__builtin_ctz(rot) + j0) & (unsigned)(SLOTS - 1);
            const unsigned long long cur = atomicCAS((unsigned long long *)&table[b + i], 0ull, (unsigned long long)fp);
            if (cur == 0) return true;
            if (cur == fp) return false;
            rot &= rot - 1;   // another fingerprint took it meanwhile: on to the next slot of the sequence
        }
#else
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            unsigned long long cur = slot[i];
            if (cur == 0) cur = atomicCAS((unsigned long long *)&table[b + i], 0ull, (unsigned long long)fp);
            if (cur == 0) return true;
            if (cur == fp) return false;
        }
#endif
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
template <bool BLIND = false>
__device__ __forceinline__ bool seen_insert(uint64_t *table, uint64_t nbuckets, uint64_t fp, unsigned &err) {
    if (nbuckets & SEEN_SPARSE) return seen_insert_t<MC_SPARSE_SLOTS, BLIND>(table, nbuckets & ~SEEN_SPARSE, fp, err);
    return seen_insert_t<8, BLIND>(table, nbuckets, fp, err);
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
constexpr unsigned MC_FI_PARK = 1u << 19;  // internal flag bit of launch_expand (not in include/tlamc.h): the PARK instantiation of k_expand_family
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

// MC_EXPAND_INSERT_MINW (the generated-code translation unit sets it, pcal_codegen.cpp): wavefronts per SIMD the register allocation must leave
// room for — a generated lowering keeps a whole state in registers twice (parent and successor) and would otherwise take the 512 a lone
// wavefront may have
#ifndef MC_EXPAND_INSERT_MINW
#define MC_EXPAND_INSERT_MINW 1
#endif
template <class S, bool ROUTE>
__global__ void __launch_bounds__(256, MC_EXPAND_INSERT_MINW)
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
            const unsigned ps = stored_state_status<S>(prm, loc, s);  // specs that check invariants per expanded state
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
#ifndef MC_WFILT
#define MC_WFILT 128   // (round 5: 256 -> 128 entries; the kilobyte went to the survivor list, MC_OCAP below)
#endif
constexpr int WFILT = MC_WFILT;
static_assert(WFILT >= 64 && (WFILT & (WFILT - 1)) == 0, "the filter is direct-mapped by fingerprint bits and cleared 64 entries at a time");
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
// The parent's Summary travels in REGISTERS, word by word: an aggregate passed by value goes through the stack, i.e. through scratch
// memory — 48 bytes stored and loaded again per lane and call, 27 GB per step of the contract workload that the L2 had to hold beside
// the seen-set lines and the parent rows (round 5; the kernel has no private segment left but the callee's one saved register).
template <class S, size_t... I>
__device__ __noinline__ void wave_write_survivors_r(typename S::Params prm_v, const uint64_t *arena_v, uint64_t pidx, bool mine, unsigned slot, uint64_t fp,
                                                    uint64_t *arena_w_v, uint64_t oidx, std::index_sequence<I...>, decltype((void)I, uint64_t{})... qw) {
    const typename S::Params prm = wave_uniform_copy(prm_v);
    const uint64_t *arena = (const uint64_t *)uniform_ptr(arena_v);
    uint64_t *arena_w = (uint64_t *)uniform_ptr(arena_w_v);
    if (!mine) return;
    const uint64_t words[] = {qw...};
    typename S::Summary q;
    static_assert(sizeof(q) == sizeof(words), "the Summary is passed as whole 64-bit words");
    __builtin_memcpy(&q, words, sizeof q);
    const int W = S::words(prm);
    const CWordRef sp = arena_cref(arena, pidx, W);
    if constexpr (HasSummaryWriter<S>::value) S::apply_summary_patch(prm, q, sp, (int)slot, fp, arena_ref(arena_w, oidx, W));
    else if constexpr (HasKnownFp<S>::value) S::apply_known_fp(prm, sp, (int)slot, fp, arena_ref(arena_w, oidx, W));
    else S::apply(prm, sp, (int)slot, arena_ref(arena_w, oidx, W));
}
template <class S, size_t... I>
__device__ __forceinline__ void wave_write_survivors_split(typename S::Params prm, const uint64_t *arena, uint64_t pidx, bool mine, unsigned slot, uint64_t fp,
                                                          uint64_t *arena_w, uint64_t oidx, const typename S::Summary &q, std::index_sequence<I...> seq) {
    uint64_t words[sizeof...(I)];
    __builtin_memcpy(words, &q, sizeof words);
    wave_write_survivors_r<S>(prm, arena, pidx, mine, slot, fp, arena_w, oidx, seq, words[I]...);
}
#if defined(MC_SUMMARY_BY_STACK) && MC_SUMMARY_BY_STACK   // A/B: rounds 1-4's form, the Summary by value (= through scratch memory)
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
#else
template <class S>
__device__ __forceinline__ void wave_write_survivors(typename S::Params prm, const uint64_t *arena, uint64_t pidx, bool mine, unsigned slot, uint64_t fp,
                                                    uint64_t *arena_w, uint64_t oidx, const typename S::Summary &q) {
    static_assert(sizeof(typename S::Summary) % 8 == 0, "the Summary is passed as whole 64-bit words");
    wave_write_survivors_split<S>(prm, arena, pidx, mine, slot, fp, arena_w, oidx, q, std::make_index_sequence<sizeof(typename S::Summary) / 8>{});
}
#endif
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
// Round 5: 448 entries (a ring that need not be a power of two).  With 256 a wavefront with more than 192 survivors — three per parent:
// every level that grows by 2 x and more has many — pushed 64 at a time through the global new-list, and the level could not end before
// a k_materialise had written them, alone on the device (16 level ends x 230 us on the contract workload, profiles/r05l_levels_*.txt,
// and 27 M of its 526 M states re-read from HBM by that kernel); with 384 (320 before the first batch leaves) the list is the rare
// exception (t3 138.3 -> 132.0 ms, profiles/r05m), with 448 rarer still (-1.2 %; the 5-server model 171.5 -> 169.0, profiles/r05t): 448
// is what fits — the tail's sort order (2 x 448 x 2 bytes) fills the dead family queues + filter exactly, and eight workgroups of
// 20.3 KB fill the CU's 160 KB of LDS.
#ifndef MC_OCAP
#define MC_OCAP 448
#endif
constexpr int OCAP = MC_OCAP;
static_assert(OCAP >= 128 && OCAP % 64 == 0 && OCAP <= 512, "survivor list: whole batches; positions have 9 bits in the tail's sort order");
// position in the survivor ring: x < 2 * OCAP
__device__ __forceinline__ unsigned owrap(unsigned x) {
    if constexpr ((OCAP & (OCAP - 1)) == 0) return x & (unsigned)(OCAP - 1);
    else return x >= (unsigned)OCAP ? x - (unsigned)OCAP : x;
}
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
#if MC_ASYNC_PROBE && MC_SEEN_ROTATE
#error "the split-phase probes (measured and not adopted in round 5) follow the first-empty-slot order: build them with -DMC_SEEN_ROTATE=0"
#endif
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
    uint64_t filt[WFILT];          // (directly behind fq: both are dead when the tail begins, its sort order lies over the two)
    typename S::Summary sum[NB * 64];
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

// ONE ROUND of the workgroup's tail (k_expand_family, in-wave writes): the wavefronts pool what their survivor lists hold, sort it by action
// class, take the arena indices with one atomicAdd and write the rows.  Inlined into the kernel for the first round — the only one of a
// workgroup whose wavefronts parked nothing — and into tail_more_rounds below for the others.  Returns whether another round is due
// (some wavefront still has parked chunks; false also when the arena is full: err).  prof(phase): MC_PROF of a profiling build.
// (the LDS pointers are template types: generic pointers in the kernel, where the compiler infers LDS, address-space-3 pointers in
//  tail_more_rounds, where it cannot — a flat access to LDS is slower than a ds_ one)
template <class S, int WAVES, bool PARK, class WQ, class FL, class HP, class OP, class RP, class PP, class Prof>
__device__ __forceinline__ bool tail_round(const typename S::Params &prm, const uint64_t *arena, uint64_t *arena_w, uint64_t arena_cap, uint32_t *parent, uint16_t *pslot,
                                           DevCounters *ctr, unsigned flags, WQ wq, FL fls, HP hist_base, unsigned hist_stride,
                                           OP wg_out0, RP order, PP wg_park, unsigned w, unsigned lane, uint64_t wg_idx0,
                                           unsigned ohead, unsigned on, unsigned &err, Prof &&prof) {
        constexpr int NCLS = SlotClasses<S>::value;
        auto &Q = wq[w];
        auto hist_at = [&](unsigned c, unsigned ww) -> auto & { return hist_base[ww * hist_stride + c]; };
        (void)flags;
        // a wavefront counts its own survivors per class as soon as IT has finished — in the shadow of the wait for its siblings
        prof(16);      // (profiling builds: 16 = the counting sort, 4 = waiting at barrier (1), 17 = the writes)
        unsigned ccnt[NCLS];
#pragma unroll
        for (int c = 0; c < NCLS; ++c) ccnt[c] = 0;
        for (unsigned t = 0; t < on; t += 64) {
            const bool valid = t + lane < on;
            const unsigned e_ = valid ? Q.o_ent[owrap(ohead + t + lane)] : O_DEAD;
            const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) ccnt[c] += (unsigned)__popcll(__ballot(cls == c));
        }
#pragma unroll
        for (int c = 0; c < NCLS; ++c) if (lane == 0) hist_at((unsigned)c, w) = ccnt[c];
        prof(4);
        __syncthreads();  // (1) no wavefront of the workgroup generates any more: the filters are free, the lists final, the counts there
        prof(16);
        const unsigned h = lane < (unsigned)(NCLS * WAVES) ? hist_at(lane / WAVES, lane % WAVES) : 0u;
        unsigned incl = h;
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if ((int)lane >= o) incl += u; }
        const unsigned excl = incl - h, total = __shfl(incl, 63);
        bool wg_more = false;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) if constexpr (PARK) wg_more |= wg_park[ww] != 0u;   // (written by its wavefront before barrier (1) or before barrier (3) of the round before)
#if defined(MC_TAIL_ABLATE) && MC_TAIL_ABLATE
        // ABLATION BUILD ONLY (profiles/tail_ablate.py: ONE level is timed, its output is garbage and is never expanded).  flags bit 21: every
        // workgroup's survivors start on a 64-state boundary — every column store of the writer is one whole 512-byte row of a block: the
        // most that sector-aligned allocation could ever give (the holes are left as they are)
        if (w == 0 && lane == 0) {
            if (flags & (1u << 21)) *wg_out0 = total ? ((atomicAdd(&ctr->arena_next, (unsigned long long)(((total + 63u) & ~63u) + 64u)) + 63ull) & ~63ull) : 0ull;
            else *wg_out0 = total ? atomicAdd(&ctr->arena_next, (unsigned long long)total) : 0ull;
        }
#else
        if (w == 0 && lane == 0) *wg_out0 = total ? atomicAdd(&ctr->arena_next, (unsigned long long)total) : 0ull;
#endif
#pragma unroll
        for (int c = 0; c < NCLS; ++c) ccnt[c] = __shfl(excl, c * WAVES + (int)w);  // where this wavefront's class-c survivors go
        for (unsigned t = 0; t < on; t += 64) {
            const bool valid = t + lane < on;
            const unsigned k = owrap(ohead + t + lane);
            const unsigned e_ = valid ? Q.o_ent[k] : O_DEAD;
            const int cls = e_ != O_DEAD ? SlotClasses<S>::of((int)(e_ >> 6)) : -1;
#pragma unroll
            for (int c = 0; c < NCLS; ++c) {
                const unsigned long long b = __ballot(cls == c);
                if (cls == c) order[ccnt[c] + (unsigned)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)((w << 9) | k);
                ccnt[c] += (unsigned)__popcll(b);
            }
        }
        __syncthreads();  // (2) the order and the workgroup's first arena index are visible
        prof(17);
        const unsigned long long out0 = *wg_out0;
        if (total && out0 + total > arena_cap) {
            err |= DEV_EARENA;
            return false;  // (workgroup-uniform: out0 and total are the workgroup's)
        } else {
            for (unsigned bt = w * 64u; bt < total; bt += 64u * WAVES) {  // batch of 64 sorted survivors; the wavefronts take turns
                const bool mine = bt + lane < total;
                const unsigned ref = mine ? order[bt + lane] : 0u;
                const unsigned e = mine ? wq[ref >> 9].o_ent[ref & 511u] : 0u;
                const uint64_t sfp = mine ? wq[ref >> 9].o_fp[ref & 511u] : 0ull;
                const uint64_t pidx = wg_idx0 + (ref >> 9) * 64u + (e & 63u), oidx = out0 + bt + lane;
                typename S::Summary qsum;   // (the parent's Summary: copied word by word whatever address space `fls` points into)
                __builtin_memcpy(&qsum, &fls[ref >> 9].sum[e & 63u], sizeof qsum);
#if defined(MC_TAIL_ABLATE) && MC_TAIL_ABLATE
                // flags bit 20: the writer reads "its parent" from the arena's first two blocks (same lanes, same instructions, every gather an
                // L1 / L2 hit): the most that keeping the parent rows on the CU could give.  bit 22: every row is stored into the arena's LAST
                // 64 states (same stores, no HBM write traffic to speak of).  bit 23: no writer at all (allocation only).
                {
                    const uint64_t rpidx = (flags & (1u << 20)) ? (uint64_t)((ref >> 9) * 64u + (e & 63u)) : pidx;
                    const uint64_t woidx = (flags & (1u << 22)) ? (arena_cap - 64u + (oidx & 63u)) : oidx;
                    wave_write_survivors<S>(prm, arena, rpidx, mine && !(flags & (1u << 23)), e >> 6, sfp, arena_w, woidx, qsum);
                }
#else
                wave_write_survivors<S>(prm, arena, pidx, mine, e >> 6, sfp, arena_w, oidx, qsum);
#endif
                if (mine && parent) { parent[oidx] = (uint32_t)pidx; pslot[oidx] = (uint16_t)(e >> 6); }
            }
        }
        return wg_more;
}
// ROUNDS 2 .. of the tail: every wavefront refills its list from its chain of parked chunks (up to OCAP survivors, newest chunk first) and
// the round runs again, until the chains are empty.  Out of line — the exception (11 % of the five-server model's states, 0.15 % of the
// contract workload's), and as a loop around the round in the kernel itself it cost the 3-server kernel 8 - 12 spilled VGPRs inside the
// writer's batch loop (profiles/r06ze: 4 % of the contract line).  Returns DEV_E* bits.
template <class S, int WAVES, class FamLdsT>
__device__ __noinline__ unsigned tail_more_rounds(typename S::Params prm_v, const uint64_t *arena_v, uint64_t *arena_w_v, uint64_t arena_cap_v, uint32_t *parent_v,
                                                  uint16_t *pslot_v, DevCounters *ctr_v, unsigned flags_v, FamQueues *wq_v, FamLdsT *fls_v, unsigned *hist_v,
                                                  unsigned hist_stride_v, unsigned long long *wg_out0_v, uint16_t *order_v, unsigned *wg_park_v, uint64_t wg_idx0_v,
                                                  const uint32_t *seg_v, const uint64_t *fps_v) {
    const typename S::Params prm = wave_uniform_copy(prm_v);
    const uint64_t *arena = uniform_generic(arena_v);
    uint64_t *arena_w = uniform_generic(arena_w_v);
    const uint64_t arena_cap = wave_uniform_copy(arena_cap_v), wg_idx0 = wave_uniform_copy(wg_idx0_v);
    uint32_t *parent = uniform_generic(parent_v);
    uint16_t *pslot = uniform_generic(pslot_v);
    DevCounters *ctr = uniform_generic(ctr_v);
    const unsigned flags = __builtin_amdgcn_readfirstlane(flags_v), hist_stride = __builtin_amdgcn_readfirstlane(hist_stride_v);
    auto wq = uniform_lds(wq_v);               // (LDS pointers: address-space-3 pointers rebuilt from their 32-bit LDS addresses)
    auto fls = uniform_lds(fls_v);
    auto hist_base = uniform_lds(hist_v);
    auto wg_out0 = uniform_lds(wg_out0_v);
    auto order = uniform_lds(order_v);
    auto wg_park = uniform_lds(wg_park_v);
    const uint32_t *seg = uniform_generic(seg_v);
    const uint64_t *fps = uniform_generic(fps_v);
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned err = 0;
    for (;;) {
        __syncthreads();  // (3) nobody reads a list, the order or the counts of the round before any more
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // (the chunks were stored by this very wavefront)
        unsigned on = 0;
        unsigned park = (unsigned)__builtin_amdgcn_readfirstlane((int)wg_park[w]);
        while (park && on + 64u <= (unsigned)OCAP) {
            const uint64_t p0 = (uint64_t)park - 1u;
            const uint32_t se = seg[p0 + lane];
            wq[w].o_ent[on + lane] = (uint16_t)(((se >> 24) << 6) | (se & 63u));
            wq[w].o_fp[on + lane] = fps[p0 + lane];
            park = (unsigned)__builtin_amdgcn_readfirstlane((int)seg[p0 + 64]);
            on += 64;
        }
        if (lane == 0) wg_park[w] = park;   // (read by every wavefront after barrier (1) of the round that begins here)
        wave_lds_fence();
        if (!tail_round<S, WAVES, true>(prm, arena, arena_w, arena_cap, parent, pslot, ctr, flags, wq, fls, hist_base, hist_stride, wg_out0, order, wg_park, w, lane,
                                  wg_idx0, 0u, on, err, [](int) {})) break;
    }
    return err;
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
#define MC_EXPAND_WAVES 2
#endif
// PARK (round 6): the instantiation whose in-wave writers park the overflow of their survivor lists and write it in later rounds of the
// tail (see `spill` below).  A template parameter, not a flag: the parking code costs the kernel a dozen scalar registers and two spilled
// VGPRs it should not pay on a model whose lists hardly ever fill up (the contract workload: 0.15 % of the states; profiles/r06zf, r06zg) —
// and for the five-server model, whose lists do (11 %), the step is no shorter with every state written in-wave (r06zh): MC_F_PARK, not the default.
template <class S, bool ROUTE, int NB, int MINW = MC_EXPAND_MINW, int WAVES = MC_EXPAND_WAVES, bool PARK = false>
__global__ void __launch_bounds__(64 * WAVES, MINW)
k_expand_family(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi, uint64_t ncols,
                uint64_t *table, uint64_t mask, uint32_t *__restrict__ newlist, uint64_t seg_cap, DevCounters *ctr, unsigned flags,
                RouteArgs rt, unsigned parity) {
    static_assert(NB >= 1 && NB <= 4, "the queue entry has two bits for the block");
    static_assert(WAVES == 1 || WAVES == 2 || WAVES == 4, "wavefronts per workgroup");
    using FamLdsT = FamLds<S, NB>;
    __shared__ FamQueues wq[WAVES];
    __shared__ FamLdsT fls[WAVES];
    if (rt.lc) {
        if (rt.lc->stop) return;
        lo = rt.lc->lo;
        hi = rt.lc->hi;
        ncols = ((hi - (lo & ~63ull)) + 63) & ~63ull;
    }
    const unsigned lane = threadIdx.x & 63;
    FamQueues &Q = wq[threadIdx.x >> 6];
    FamLdsT &FL = fls[threadIdx.x >> 6];
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
    // IN-WAVE OVERFLOW, PARKED (round 6; the PARK instantiation, MC_F_PARK).  A wavefront whose survivor list is about to fill up used
    // to push its 64 oldest survivors through the global new-list to k_materialise — a second kernel beside the expand, unsorted batches
    // (every branch of the writer walked), the parents read again from HBM: 0.15 % of the contract workload's states but 11 % of the
    // five-server model's (fan-out ~27 per parent, W = 192 B: 103 M states, a 93 - 109 ms kernel beside a 156 ms one).  Now the 64
    // survivors are PARKED in the new-list's memory — 10 bytes per survivor instead of a row, as a chunk linked to the wavefront's chunk
    // before (the wavefront keeps one position, no table) — and the workgroup's tail takes them back: after the round that writes
    // what the lists hold, every wavefront refills its list from its chain (up to OCAP entries) and the tail runs again, until the
    // chains are empty.  Same sort by action class, same writer, same allocation; nothing reaches k_materialise.
    // (its state lives in LDS, not in a register that would be live through the whole kernel: wg_park[w] = position + 1 of wavefront w's
    //  newest parked chunk in its new-list segment, 0 = none — first try: one more scalar register and the refill loop cost the 3-server
    //  kernel 8 spilled VGPRs and 4 % of the contract line)
    static_assert(!PARK || (!ROUTE && !ASYNC_BUILD), "parking: fused runs, synchronous probes");
    __shared__ unsigned wg_park[WAVES];
    if constexpr (PARK) { if (lane == 0) wg_park[threadIdx.x >> 6] = 0u; }
    bool tail_more = false;   // workgroup-uniform: the tail's first round left parked chunks (the later rounds run at the kernel's very end)
    // (the tail's class counts and first arena index: at function scope, the later rounds name them again)
    __shared__ unsigned wg_hist_s[ASYNC_BUILD ? 1 : WAVES * SlotClasses<S>::value];
    __shared__ unsigned long long wg_out0_s;
    auto spill_on = [&]() __attribute__((always_inline)) -> bool {   // wave-uniform
        if constexpr (PARK) return inwave && !(flags & MC_F_WAVETAIL) && rt.new_fp != nullptr;
        else return false;
    };
    MC_PROF_DECL

    auto flush_out = [&](unsigned take) __attribute__((always_inline)) {
        MC_PROF(4);
        // (the `take` oldest entries: all confirmed — tentative ones are the newest — but some may be tombstones)
        const unsigned e = lane < take ? Q.o_ent[owrap(ohead + lane)] : O_DEAD;
        const unsigned long long bl = __ballot(e != O_DEAD);
        unsigned long long pos = 0;
        // (spill: a chunk is 64 entries + the link; same segment, a cursor of its own — k_materialise never sees these entries)
        const bool spill = spill_on();
        if (lane == 0 && bl) pos = spill ? atomicAdd(&ctr->n_side[pshard].v, 65ull) : atomicAdd(&ctr->n_new[pshard].v, (unsigned long long)__popcll(bl));
        unsigned long long pos0;  // (lane 0's)
        if constexpr (PARK) pos0 = wave_uniform_copy(pos);
        else pos0 = __shfl(pos, 0);
        pos = pos0 + (unsigned)__popcll(bl & ((1ull << lane) - 1ull));
        if (e != O_DEAD) {
            seg[pos] = (uint32_t)(wave_col0 + (e & 63u)) | ((uint32_t)(e >> 6) << 24);
            if (rt.new_fp) rt.new_fp[(uint64_t)pshard * seg_cap + pos] = Q.o_fp[owrap(ohead + lane)];
        }
        if (spill) {
            if (lane == 0) {
                seg[pos0 + 64] = wg_park[threadIdx.x >> 6];
                wg_park[threadIdx.x >> 6] = (uint32_t)pos0 + 1u;
            }
            wave_lds_fence();
        }
        ohead = owrap(ohead + take);
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
                    const unsigned k = owrap(ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull)));
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
                const unsigned k = owrap(ohead + on + (unsigned)__popcll(b & ((1ull << lane) - 1ull)));
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
                        const unsigned pos = owrap(cas_obase + (unsigned)__popcll(casmask & ((1ull << lane) - 1ull)));
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
                        const unsigned k = owrap(ohead + on + (unsigned)__popcll(bn & ((1ull << lane) - 1ull)));
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
                        const unsigned k = owrap(ohead + on + (unsigned)__popcll(bc & ((1ull << lane) - 1ull)));
                        Q.o_ent[k] = (uint16_t)(ent & ((1u << Q_DSP_SHIFT) - 1u));
                        Q.o_fp[k] = fp;
                        casret = cur;
                    }
                    casmask = bc;
                    cas_obase = owrap(ohead + on);
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
            }
        }
        MC_PROF(5);
        const int lane_inflight = active ? loc.inflight : 0;
        // (round 5, measured and NOT adopted: the drain of the family queues folded into this loop as its last step, so that run_full —
        //  family batches and the probe code behind enqueue — is inlined once: 14 KB less code, no spills, and 147.3 -> 153.6 ms per step
        //  on the t3 graph, profiles/r05b_ab.jsonl `base` against `old`; the two message loops above folded into one: 14 spilled VGPRs)
        for (int step = 0; step < ((flags & 64u) ? 0 : S::FIX); ++step) {
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
            static_assert(OCAP * sizeof(uint16_t) <= sizeof(FL.fq) + sizeof(FL.filt), "a wavefront's sorted order fits its own dead queues + filter");
            uint16_t *order = reinterpret_cast<uint16_t *>(&FL.fq[0][0]);  // this wavefront's own family queues + filter: dead, its generation is over
            unsigned ccnt[NCLS];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) ccnt[c] = 0;
            for (unsigned t = 0; t < on; t += 64) {
                const bool valid = t + lane < on;
                const unsigned e_ = valid ? Q.o_ent[owrap(ohead + t + lane)] : O_DEAD;
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
                const unsigned k = owrap(ohead + t + lane);
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
        static_assert(sizeof(fls[0].fq) % 8 == 0 && offsetof(FamLdsT, filt) == sizeof(fls[0].fq), "the filter lies directly behind the family queues");
        static_assert(WAVES * OCAP * sizeof(uint16_t) <= sizeof(fls[0].fq) + sizeof(fls[0].filt), "the sorted order aliases the family queues + the duplicate filter of the first wavefront");
        uint16_t *order = reinterpret_cast<uint16_t *>(&fls[0].fq[0][0]);   // [WAVES * OCAP]: (wavefront << 9) | position in its list
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
            hist_base = wg_hist_s;
            hist_stride = NCLS;
            wg_out0 = &wg_out0_s;
        }
        const unsigned w = threadIdx.x >> 6;
        {
            const uint64_t wg_idx0 = base + (uint64_t)blockIdx.x * (64u * WAVES);
            const bool more = tail_round<S, WAVES, PARK>(prm, arena, rt.arena_w, rt.arena_cap, rt.parent, rt.pslot, ctr, flags, &wq[0], &fls[0], hist_base, hist_stride, wg_out0,
                                                   order, &wg_park[0], w, lane, wg_idx0, ohead, on, err, [&](int ph) { MC_PROF(ph); (void)ph; });
            tail_more = more;   // (rounds 2 ..: at the kernel's very end, when nothing of the wavefront's own state is live any more)
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
    if constexpr (PARK) {
        // ROUNDS 2 .. of the tail (a workgroup whose wavefronts parked survivors): out of line (tail_more_rounds), and HERE — after the
        // wavefront has handed in its counters — so that no value of the kernel is live across the call; every argument is a kernel
        // argument or the address of a shared variable
        if (tail_more) {
            const unsigned e2 = tail_more_rounds<S, WAVES, FamLdsT>(prm, arena, rt.arena_w, rt.arena_cap, rt.parent, rt.pslot, ctr, flags, &wq[0], &fls[0], &wg_hist_s[0],
                                                                    (unsigned)SlotClasses<S>::value, &wg_out0_s, reinterpret_cast<uint16_t *>(&fls[0].fq[0][0]), &wg_park[0],
                                                                    base + (uint64_t)blockIdx.x * (64u * WAVES), seg, rt.new_fp + (uint64_t)pshard * seg_cap);
            if (e2 && lane == 0) atomicOr(&ctr->error, e2);
        }
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
// (engine_pairs.h, included after this file)
template <class S>
static bool launch_expand_pairs(hipStream_t stream, typename S::Params prm, const uint64_t *arena, uint64_t lo, uint64_t hi, uint64_t ncols, uint64_t *table,
                                uint64_t mask, uint32_t *newlist, uint64_t seg_cap, DevCounters *ctr, unsigned flags, RouteArgs rt, unsigned parity);
template <class S, bool ROUTE, class... A>
static void launch_expand(bool by_family, unsigned flags, uint64_t ncols, hipStream_t stream, unsigned slices, A... args) {
    if constexpr (!ROUTE) {
        // specs with the by-pairs protocol: the fused expand + insert + write kernel whenever the run writes in-wave (rt.arena_w)
        if (by_family && launch_expand_pairs<S>(stream, args...)) return;
    }
    if constexpr (UsesFamilies<S>::value) {
        if (by_family) {
            // MC_F_OCC3 (A/B): the register budget of 3 wavefronts per SIMD (no spills) instead of 4 (a dozen spilled VGPRs)
            constexpr unsigned WG = 64u * MC_EXPAND_WAVES;   // columns (parents) per workgroup
            if constexpr (!ROUTE) {
                if (flags & MC_FI_PARK) {   // (set by the host's level loop, never by a caller: see k_expand_family's PARK)
                    hipLaunchKernelGGL((k_expand_family<S, false, 1, MC_EXPAND_MINW, MC_EXPAND_WAVES, true>), dim3((unsigned)((ncols + WG - 1) / WG)), dim3(WG), 0, stream, args...);
                    return;
                }
            }
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
    for (int t = 0; t < NSHARD; t++) { n += ctr->n_new[parity * NSHARD + t].v; ctr->n_new[parity * NSHARD + t].v = 0; ctr->n_side[parity * NSHARD + t].v = 0; }
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
        const unsigned ps = stored_state_status<S>(prm, loc, g);
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
        for (int t = 0; t < NSHARD; t++) { n += ctr->n_new[t].v; ctr->n_new[t].v = 0; ctr->n_side[t].v = 0; }
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


}  // namespace mc

#endif  // TLAMC_ENGINE_KERNELS_H
