// engine_pairs.h — k_expand_pairs: the fused expand + insert + WRITE kernel for specs whose successor is "parent + a small delta" and
// whose action slots all have compile-time indices (S::PAIR_FAMILIES; round 6: examples/serializableSnapshotIsolation.tla — BASELINE
// config 5 — whose 77 slots per state walked, one lane per state, every slot body for the whole wavefront whenever ONE lane was enabled:
// 24.6 VALU + 11.9 SALU wave-instructions per generated successor, and a second kernel re-derived every new state from scratch).
// Included by engine.hip only (inside namespace mc, after engine_kernels.h).  Its own file so that the stamp of a counter collection
// (bench.py kernel_source_hash) of the raft kernels does not move when this kernel changes, and the other way round.
//
// One WAVEFRONT = one arena block of 64 parents; wavefronts are independent (no workgroup barrier, no shared state but the device
// counters):
//   load      lane = parent: the row in one coalesced round trip, staged in LDS; S::load builds the parent's tables, S::parent_status
//             checks the invariants of the expanded state, S::summarize leaves the tables the actions read in LDS, S::guards yields
//             the mask of slots whose guard holds (128 bits per lane)
//   layout    per-family counts -> one 64-bit wave scan (16 bits per family) -> every lane scatters its enabled slots into the
//             wavefront's pair list, FAMILY-MAJOR: entry = (slot << 6) | parent lane
//   pass 1    batches of 64 pairs of ONE family: S::eval_pair<F> from the parent's Summary + row (LDS) -> status, fingerprint ->
//             seen-set probe / insert, all 64 lanes at once; pairs the seen-set knew are struck from the list
//   alloc     ONE atomicAdd(arena_next) per wavefront for the survivors of all its parents (~8 per parent on config 5: a word takes
//             ~88 returning atomics per microsecond, MI355X_MICROARCH "dequeue", so one per batch would bound the kernel)
//   pass 2    the same batches again, the survivors only: S::eval_pair<F> once more (the delta is a few dozen instructions; keeping
//             16 bytes per pair through the probes instead would cost 20 KB of LDS per wavefront, i.e. the occupancy that hides the
//             probes' latency) and S::write_pair: lanes = consecutive arena indices, whole rows of the word-major blocks; the
//             parent's words come from LDS — no second read of the parent, no new-list, no k_materialise.
// A wavefront whose 64 parents enable more pairs than the list holds (never seen on config 5) works in S::PAIR_ROUNDS rounds of at
// most S::PAIR_ROUND_SLOTS slots per parent.
#ifndef TLAMC_ENGINE_PAIRS_H
#define TLAMC_ENGINE_PAIRS_H

namespace mc {

template <class S, class = void>
struct UsesPairs : std::false_type {};
template <class S>
struct UsesPairs<S, decltype((void)S::PAIR_ROUNDS)> : std::true_type {};

// DYNAMIC KEYS (S::PAIR_KEYS): the code a pair runs is not a function of its slot alone but of the parent too — a generated PlusCal lowering
// (spec_gen.h): slot = (process instance, choice), code = the LABEL that instance stands at in this parent.  The pair list is then laid
// out by S::pair_key(parent row, slot) in [0, PAIR_KEYS), PAIR_KEYS <= 64, with a counting sort through an LDS histogram (two walks over
// a lane's enabled slots: count, then claim positions), and evaluated as ONE family whose eval_pair switches on the key: inside a batch
// of 64 pairs (almost) every lane takes the same case.  Slot-by-slot, a wavefront walked every label some parent stood at for every slot:
// 14 000 lane-instructions per successor on the Michael-Scott queue model.
template <class S, class = void>
struct DynamicKeys : std::integral_constant<int, 0> {};
template <class S>
struct DynamicKeys<S, decltype((void)S::PAIR_KEYS)> : std::integral_constant<int, S::PAIR_KEYS> {};
// specs without a pair base (the successor's fingerprint is computed from the whole successor): S::W_PAIR_BASE < 0
template <class S>
constexpr bool has_pair_base() { return S::W_PAIR_BASE >= 0; }

constexpr int PAIR_CAP = 1280;             // pair-list entries per wavefront
constexpr unsigned PAIR_DEAD = 0xffffu;    // a pair the seen-set already knew (or that was not enabled / not in the model)
#ifndef MC_PAIR_WAVES
#define MC_PAIR_WAVES 4
#endif
#ifndef MC_PAIR_COMPACT    // 0 = A/B: pass 2 walks the list with its holes
#define MC_PAIR_COMPACT 1
#endif
#ifndef MC_PAIR_MINW       // wavefronts per SIMD the register allocation leaves room for (the generated-code unit: 2 — a state twice in registers)
#define MC_PAIR_MINW 4
#endif

template <class S>
struct PairLds {
    uint64_t row[S::MAX_WORDS][64];   // the 64 parents' rows, word-major like the arena block they came from; word S::W_PAIR_BASE holds
                                      // S::pair_base instead (the part of a successor's fingerprint its parent determines)
    typename S::Summary sum[64];
    uint16_t list[PAIR_CAP];
    unsigned succ[2];                 // deadlock check: bit p = parent p has a successor
    unsigned kcnt[DynamicKeys<S>::value > 0 ? DynamicKeys<S>::value : 1];   // dynamic keys: histogram, then the keys' cursors
};
// a parent's row in LDS (address-space-qualified: ds_read_b64, not flat_load)
struct LdsRow {
    const __attribute__((address_space(3))) uint64_t *col;  // &row[0][parent]
    __device__ __forceinline__ uint64_t get(int w) const { return col[w * 64]; }
};
// a row held in registers by the lane that loaded it (indices are compile-time constants after unrolling)
template <int N>
struct RegRow {
    const uint64_t *r;
    __device__ __forceinline__ uint64_t get(int w) const { return r[w]; }
};

// specs whose candidates are mostly NEW states probe with a blind first compare-and-swap (S::BLIND_INSERT; seen_insert_t in engine_kernels.h)
template <class S, class = void>
struct BlindInsert : std::false_type {};
template <class S>
struct BlindInsert<S, decltype((void)S::BLIND_INSERT)> : std::integral_constant<bool, S::BLIND_INSERT && MC_SEEN_ROTATE> {};

template <class S, int WAVES = MC_PAIR_WAVES>
__global__ void __launch_bounds__(64 * WAVES, MC_PAIR_MINW)
k_expand_pairs(typename S::Params prm, const uint64_t *__restrict__ arena, uint64_t lo, uint64_t hi, uint64_t ncols,
               uint64_t *table, uint64_t mask, DevCounters *ctr, unsigned flags, RouteArgs rt) {
    constexpr int NF = S::PAIR_FAMILIES, MW = S::MAX_WORDS;
    static_assert(NF >= 1 && NF <= 4, "per-family counts travel as 16-bit fields of one 64-bit scan");
    static_assert(S::PAIR_ROUND_SLOTS * 64 <= PAIR_CAP, "a round's pairs fit the list");
    static_assert(S::TOTAL_SLOTS <= 128 && S::TOTAL_SLOTS < 1024, "guard mask: 128 bits; list entry: slot << 6 | lane in 16 bits");
    constexpr int KEYS = DynamicKeys<S>::value;
    static_assert(KEYS <= 64 && (KEYS == 0 || NF == 1), "dynamic keys: one lane per key in the prefix sum, one family");
    __shared__ PairLds<S> lds[WAVES];
    if (rt.lc) {
        if (rt.lc->stop) return;
        lo = rt.lc->lo;
        hi = rt.lc->hi;
        ncols = ((hi - (lo & ~63ull)) + 63) & ~63ull;
    }
    const unsigned lane = threadIdx.x & 63;
    PairLds<S> &L = lds[threadIdx.x >> 6];
    const uint64_t base = lo & ~63ull;
    const uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;  // whole wavefronts leave together (ncols % 64 == 0); no barrier below
    const uint64_t idx = base + col;
    const bool active = idx >= lo && idx < hi;
    const int W = S::words(prm);
    const unsigned shard = blockIdx.x & (NSHARD - 1);
    unsigned long long viol = ~0ull;
    unsigned gen = 0, err = 0, cands = 0;
    // (profiling builds, -DMC_PHASE_PROF: cycles per phase, exclusive — 0 row load + S::load | 1 parent_status | 2 summarize, base, guards |
    //  3 layout: scan + scatter | 8 pass 1: eval_pair | 9 seen-set probe / insert | 10 allocation | 11 pass 2: eval_pair | 12 write_pair |
    //  7 epilogue; 24 / 25: pairs evaluated in pass 1 / 2; 40: wavefronts)
    MC_PROF_DECL

    // ---- load: the row (coalesced: lane = state of one arena block), tables, invariants, guards
    uint64_t glo = 0, ghi = 0;
    {
        const GlobalWords blk = uniform_ptr(arena + ((idx - lane) >> 6) * (uint64_t)W * 64);
        uint64_t r[MW];
#pragma unroll
        for (int w = 0; w < MW; w++) r[w] = (active && w < W) ? blk[(unsigned)w * 64u + lane] : 0ull;
#pragma unroll
        for (int w = 0; w < MW; w++) L.row[w][lane] = r[w];
        if (active) {
            typename S::Local loc;
            const RegRow<MW> rr{r};
            S::load(prm, rr, loc);
            MC_PROF(1);
            const unsigned ps = stored_state_status<S>(prm, loc, rr);
            if (ps & ST_INVARIANT) viol = viol_key(idx, SLOT_PARENT, VK_INVARIANT, ps >> 8);
            MC_PROF(2);
            typename S::Summary q;
            S::summarize(loc, q);
            L.sum[lane] = q;
            if constexpr (has_pair_base<S>()) L.row[S::W_PAIR_BASE][lane] = S::pair_base(prm, loc, rr);
            if (!(flags & 64u)) S::guards(prm, loc, glo, ghi);  // (64 = ablation: load the parents only)
        }
    }
    if (lane < 2) L.succ[lane] = 0;
    const bool track_succ = (flags & MC_F_DEADLOCK) != 0;

    // ---- rounds: all pairs at once when they fit the list, else S::PAIR_ROUNDS rounds
    const unsigned total_all = wave_sum_u32((unsigned)__popcll(glo) + (unsigned)__popcll(ghi));
    const bool single = total_all <= (unsigned)PAIR_CAP;
    const int nrounds = single ? 1 : S::PAIR_ROUNDS;
    for (int round = 0; round < nrounds; ++round) {
        MC_PROF(3);
        uint64_t mlo = glo, mhi = ghi;
        if (!single) {
            uint64_t rlo = 0, rhi = 0;
            static_for<0, S::PAIR_ROUNDS>([&](auto rc) {
                if (round == decltype(rc)::value) {
                    constexpr auto rm = S::round_mask(decltype(rc)::value);
                    rlo = rm.lo; rhi = rm.hi;
                }
            });
            mlo &= rlo;
            mhi &= rhi;
        }
        unsigned fs[NF + 1];  // family f's pairs: list[fs[f] .. fs[f + 1])
        fs[0] = 0;
        if constexpr (KEYS > 0) {
            // counting sort by S::pair_key: histogram in LDS, prefix over the keys (lane = key), positions claimed from the keys' cursors
            const LdsRow myrow{(const __attribute__((address_space(3))) uint64_t *)&L.row[0][lane]};
            wave_lds_fence();  // (the previous round's list has been read)
            if (lane < (unsigned)KEYS) L.kcnt[lane] = 0;
            wave_lds_fence();
            for (uint64_t a = mlo; a; a &= a - 1) atomicAdd(&L.kcnt[S::pair_key(prm, myrow, (int)__builtin_ctzll(a))], 1u);
            for (uint64_t b = mhi; b; b &= b - 1) atomicAdd(&L.kcnt[S::pair_key(prm, myrow, 64 + (int)__builtin_ctzll(b))], 1u);
            wave_lds_fence();
            const unsigned c = lane < (unsigned)KEYS ? L.kcnt[lane] : 0u;
            unsigned incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(incl, o);
                if ((int)lane >= o) incl += u;
            }
            fs[1] = __builtin_amdgcn_readfirstlane(__shfl(incl, 63));
            if (fs[1] == 0) continue;
            wave_lds_fence();
            if (lane < (unsigned)KEYS) L.kcnt[lane] = incl - c;
            wave_lds_fence();
            for (uint64_t a = mlo; a; a &= a - 1) {
                const unsigned sl = (unsigned)__builtin_ctzll(a);
                L.list[atomicAdd(&L.kcnt[S::pair_key(prm, myrow, (int)sl)], 1u)] = (uint16_t)((sl << 6) | lane);
            }
            for (uint64_t b = mhi; b; b &= b - 1) {
                const unsigned sl = 64u + (unsigned)__builtin_ctzll(b);
                L.list[atomicAdd(&L.kcnt[S::pair_key(prm, myrow, (int)sl)], 1u)] = (uint16_t)((sl << 6) | lane);
            }
            wave_lds_fence();
        } else {
        // per-family counts of this lane, 16 bits each; inclusive scan over the lanes
        uint64_t cnt = 0;
        static_for<0, NF>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            constexpr auto fm = S::family_mask(F);
            cnt |= (uint64_t)((unsigned)__popcll(mlo & fm.lo) + (unsigned)__popcll(mhi & fm.hi)) << (16 * F);
        });
        uint64_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t u = __shfl_up(incl, o);
            if ((int)lane >= o) incl += u;
        }
        const uint64_t excl = incl - cnt;
        const uint64_t tot = __shfl(incl, 63);
#pragma unroll
        for (int f = 0; f < NF; f++) fs[f + 1] = __builtin_amdgcn_readfirstlane(fs[f] + (unsigned)(tot >> (16 * f) & 0xffffu));
        if (fs[NF] == 0) continue;
        wave_lds_fence();  // (the previous round's list has been read)
        static_for<0, NF>([&](auto fc) {
            constexpr int F = decltype(fc)::value;
            constexpr auto fm = S::family_mask(F);
            unsigned pos = fs[F] + (unsigned)(excl >> (16 * F) & 0xffffu);
            uint64_t a = mlo & fm.lo, b = mhi & fm.hi;
            while (a) {
                const unsigned s = (unsigned)__builtin_ctzll(a);
                a &= a - 1;
                L.list[pos++] = (uint16_t)((s << 6) | lane);
            }
            while (b) {
                const unsigned s = 64u + (unsigned)__builtin_ctzll(b);
                b &= b - 1;
                L.list[pos++] = (uint16_t)((s << 6) | lane);
            }
        });
        wave_lds_fence();
        }

        unsigned nsurv = 0;
        unsigned long long out0 = 0;
        bool write_ok = true;
        // PIPELINED PROBES (sparse tables: 32-byte buckets, rotated slot order).  An insert is two dependent trips to memory — read the
        // bucket, compare-and-swap into the fingerprint's first empty slot — and the second one, a memory-side atomic, is the long one
        // (call g: hiding the READ alone behind the next batch's evaluation changed nothing).  So a batch goes through three stations,
        // one per loop iteration: its home buckets are asked for right after its evaluation (P1); an iteration later the buckets are
        // looked at and the compare-and-swaps ISSUED, their return values left in flight (P2); another iteration later the returns are
        // read: 0 = new, the fingerprint itself = somebody else's insert came first, anything else (another fingerprint took the slot,
        // or the bucket was full) = the synchronous prober decides.  12 VGPRs of pipeline state; MC_PAIR_PIPE = 0 / 1: A/B (probe where
        // evaluated / hide the read only).
#ifndef MC_PAIR_PIPE
#define MC_PAIR_PIPE 2
#endif
        const bool pipelined = MC_PAIR_PIPE && MC_SEEN_ROTATE && MC_SPARSE_SLOTS == 4 && (mask & SEEN_SPARSE) != 0;  // wave-uniform
        const uint64_t nbk = mask & ~SEEN_SPARSE;
        // (no "is the station occupied" flags: an empty station holds key 0 = no candidate, and every iteration walks both stations — a
        //  path on which the buckets asked for earlier were never looked at would make the next load a write to registers with a load
        //  still in flight, and the compiler would have to wait for everything right there)
        unsigned long long p1_b[4] = {0ull, 0ull, 0ull, 0ull};
        uint64_t p1_key = 0, p2_key = 0;
        unsigned long long p2_ret = 0;
        unsigned p1_i = 0, p2_i = 0;
        bool p1_live = false, p2_live = false;
        int p2_state = 0;  // 0: known (or no candidate) | 1: compare-and-swap in flight | 2: new (decided synchronously) | 3: ask the synchronous prober
        auto finish_p2 = [&]() __attribute__((always_inline)) {  // station 3: read the compare-and-swaps' returns
            bool is_new = p2_state == 2;
            if (p2_state == 1) {
                if (p2_ret == 0ull) is_new = true;
                else if (p2_ret != p2_key) p2_state = 3;
            }
            if (p2_state == 3) is_new = seen_insert_t<4>(table, nbk, p2_key, err);
            if (p2_live && !is_new) L.list[p2_i] = (uint16_t)PAIR_DEAD;
            nsurv += (unsigned)__popcll(__ballot(is_new));
            p2_state = 0; p2_live = false;
        };
        auto advance_p1 = [&]() __attribute__((always_inline)) {  // station 2: look at the buckets, issue the compare-and-swaps
            p2_key = p1_key; p2_i = p1_i; p2_live = p1_live; p2_state = 0; p2_ret = 0;
#if MC_SEEN_ROTATE
            const unsigned j0 = (unsigned)(p1_key >> 32) & 3u;
            unsigned zm = 0;
            bool hit = false;
#pragma unroll
            for (int t = 0; t < 4; ++t) { hit |= p1_b[t] == p1_key; zm |= (p1_b[t] == 0ull ? 1u : 0u) << t; }
            if (p1_key && !hit && !(flags & 16u)) {  // (16 = ablation: no probes)
                const unsigned rot = ((zm >> j0) | (zm << (4u - j0))) & 15u;
                if (rot) {
                    const unsigned t = ((unsigned)__builtin_ctz(rot) + j0) & 3u;
                    const uint64_t bk = ((p1_key & 0xffffffffull) * nbk) >> 32;
                    p2_ret = atomicCAS((unsigned long long *)&table[bk * 4 + t], 0ull, (unsigned long long)p1_key);
                    p2_state = 1;
                } else {
                    p2_state = 3;  // the home bucket is full: the probe goes on in the next one
                }
            }
#endif
            p1_key = 0; p1_live = false;
            if (MC_PAIR_PIPE < 2) {  // A/B: wait for the compare-and-swap where it is issued
                asm volatile("" : "+v"(p2_ret));
                finish_p2();
            }
        };
        unsigned fe[NF];  // pass 2: family f's surviving pairs are list[fs[f] .. fe[f])
#pragma unroll
        for (int f = 0; f < NF; f++) fe[f] = fs[f + 1];
        for (int pass = 0; pass < 2; ++pass) {
            unsigned run = 0;  // pass 2: survivors written so far
            static_for<0, NF>([&](auto fc) {
                constexpr int F = decltype(fc)::value;
                const unsigned fend = pass == 0 ? fs[F + 1] : fe[F];
                for (unsigned b = fs[F]; b < fend; b += 64) {
                    const unsigned i = b + lane;
                    const bool mine = i < fend;
                    const unsigned e = mine ? (unsigned)L.list[i] : PAIR_DEAD;
                    const bool live = e != PAIR_DEAD;
                    const unsigned long long bl = __ballot(live);
                    if (!bl) continue;
                    const unsigned p = e & 63u;
                    const int slot = (int)(e >> 6);
                    unsigned st = 0;
                    uint64_t fp = 0;
                    typename S::PairOut po;
                    const LdsRow row{(const __attribute__((address_space(3))) uint64_t *)&L.row[0][p]};
                    MC_PROF(pass == 0 ? 8 : 11);
                    MC_PROF_PAIRS(pass, (unsigned)__popcll(bl));
                    if (live) st = S::template eval_pair<F>(prm, L.sum[p], row, slot, fp, po);
                    if (pass == 0) {
                        MC_PROF(9);
                        const uint64_t pidx = idx - lane + p;
                        uint64_t key = 0;
                        if (st & ST_ENABLED) {
                            ++gen;
                            if (track_succ) atomicOr(&L.succ[p >> 5], 1u << (p & 31u));
                            if (st & ST_OVERFLOW) err |= DEV_EOVERFLOW;
                            else if (st & ST_ASSERT) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_ASSERT, 0));
                            else if (st & ST_SPECERR) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_SPECERR, 0));
                            else {
                                if (st & ST_INVARIANT) viol = min(viol, viol_key(pidx, (unsigned)slot, VK_INVARIANT, st >> 8));
                                if (!(st & (ST_OUT_OF_MODEL | ST_SELFLOOP))) key = fp;
                            }
                        }
                        if (key) ++cands;
                        if (pipelined) {
                            finish_p2();    // the batch two evaluations back: its compare-and-swaps have had a whole evaluation to return
                            advance_p1();   // the batch before this one: its buckets have had a whole evaluation to arrive
                            // station 1: this batch's buckets are asked for now, STRAIGHT into the pipeline registers (a copy would be a
                            // use, and a use is a wait) and by every lane (a lane without a candidate reads bucket 0: a conditional
                            // load becomes a select, i.e. a use)
                            seen_load_home<4>(table, nbk, key, p1_b);
                            p1_key = key; p1_i = i; p1_live = live;
                        } else {
                            bool is_new = false;
                            if (key) is_new = (flags & 16u) ? false : seen_insert<BlindInsert<S>::value>(table, mask, key, err);
                            if (live && !is_new) L.list[i] = (uint16_t)PAIR_DEAD;
                            nsurv += (unsigned)__popcll(__ballot(is_new));
                        }
                    } else if (write_ok) {
                        MC_PROF(12);
                        const uint64_t oidx = out0 + run + (unsigned)__popcll(bl & ((1ull << lane) - 1ull));
                        if (live) {
                            S::write_pair(prm, row, po, arena_ref(rt.arena_w, oidx, W));
                            if (rt.parent) { rt.parent[oidx] = (uint32_t)(idx - lane + p); rt.pslot[oidx] = (uint16_t)slot; }
                        }
                        run += (unsigned)__popcll(bl);
                    }
                }
            });
            if (pass == 0) { finish_p2(); advance_p1(); finish_p2(); }  // drain the pipeline
            if (pass == 0) {
                MC_PROF(10);
                if (!nsurv) break;
                wave_lds_fence();  // the struck entries are visible to the lanes that read them in pass 2
                // COMPACTION (round 6, last third).  Pass 2 walked the list as pass 1 left it — the struck pairs as holes: where a third of the
                // candidates are new (the generated PlusCal models: 64 survivors of ~210 pairs per wavefront, profiles/r06zy) it evaluated 3.4
                // batches with 19 live lanes each, as expensive as pass 1 for a third of the pairs.  When fewer than 3/4 of the pairs survive,
                // every family's survivors are moved to the front of its range first (stable: the batches stay sorted by key; in place: a batch
                // is read before anything is written, and writes land at or below the positions read), and pass 2 runs full batches.
                if (MC_PAIR_COMPACT && nsurv * 4u < fs[NF] * 3u) {  // (wave-uniform)
                    static_for<0, NF>([&](auto fc) {
                        constexpr int F = decltype(fc)::value;
                        unsigned wpos = fs[F];
                        for (unsigned b = fs[F]; b < fs[F + 1]; b += 64) {
                            const unsigned i = b + lane;
                            const unsigned e = i < fs[F + 1] ? (unsigned)L.list[i] : PAIR_DEAD;
                            const unsigned long long bl = __ballot(e != PAIR_DEAD);
                            if (e != PAIR_DEAD) L.list[wpos + (unsigned)__popcll(bl & ((1ull << lane) - 1ull))] = (uint16_t)e;
                            wpos += (unsigned)__popcll(bl);
                        }
                        fe[F] = wpos;
                    });
                    wave_lds_fence();
                }
                if (lane == 0) out0 = atomicAdd(&ctr->arena_next, (unsigned long long)nsurv);
                out0 = __shfl(out0, 0);
                if (out0 + nsurv > rt.arena_cap) { err |= DEV_EARENA; write_ok = false; }
            }
        }
    }

    MC_PROF(7);
    if (track_succ) {
        wave_lds_fence();
        if (active && !(L.succ[lane >> 5] >> (lane & 31u) & 1u)) viol = min(viol, viol_key(idx, SLOT_NONE, VK_DEADLOCK, 0));
    }
    const unsigned gsum = wave_sum_u32(gen), csum = wave_sum_u32(cands);
    const unsigned long long vmin = wave_min_u64(viol);
    const unsigned eor = wave_or_u32(err);
    if (lane == 0) {
        if (gsum) atomicAdd(&ctr->generated[shard].v, (unsigned long long)gsum);
        if (csum) atomicAdd(&ctr->cells[shard].v, (unsigned long long)csum);
        if (vmin != ~0ull) atomicMin(&ctr->viol_key, vmin);
        if (eor) atomicOr(&ctr->error, eor);
    }
    MC_PROF_END;
}

// launch_expand's door (declared in engine_kernels.h): true = launched
template <class S>
static bool launch_expand_pairs(hipStream_t stream, typename S::Params prm, const uint64_t *arena, uint64_t lo, uint64_t hi, uint64_t ncols, uint64_t *table,
                                uint64_t mask, uint32_t *, uint64_t, DevCounters *ctr, unsigned flags, RouteArgs rt, unsigned) {
    if constexpr (UsesPairs<S>::value) {
        if (!rt.arena_w) return false;  // (route mode, the re-expansion ablation: the slot-by-slot kernel)
        constexpr unsigned WG = 64u * MC_PAIR_WAVES;
        hipLaunchKernelGGL((k_expand_pairs<S>), dim3((unsigned)((ncols + WG - 1) / WG)), dim3(WG), 0, stream, prm, arena, lo, hi, ncols, table, mask, ctr, flags, rt);
        return true;
    } else {
        return false;
    }
}

}  // namespace mc

#endif  // TLAMC_ENGINE_PAIRS_H
