// spec_gen.h — the Spec concept over GENERATED code: a PlusCal program that pcal_compile.cpp compiled to a bytecode image (spec_vm.h) is
// translated once more, by pcal_codegen.cpp, into straight-line C++ — one function per label of the algorithm, per initial-state enumeration
// and per invariant, its variables named cells of a local array whose indices are all compile-time constants (so the array lives in
// registers: the interpreter indexes it dynamically and keeps it in scratch memory) — and compiled for gfx950 when the model is loaded
// (hipcc at load time, the object cached by the hash of the program: engine.hip MC_TU == 9, pcal_codegen.cpp mc_jit_engine).
// This is north_star's "lowering each spec's next-state relation to a fixed-width packed state vector so that successor generation runs
// as a hand-written HIP kernel": the kernels are engine_kernels.h's, the lowering is what a hand would have written for THIS program.
//
// Host-side helpers (state text, action names) are the interpreter's (SpecVmT), and so is the row that LEAVES an engine (traces,
// mc_engine_read_states: export_row below): the two back-ends are interchangeable state by state, which is how tests/ compare them.  The row
// an engine of generated code STORES — and fingerprints — is packed to the cells' inferred ranges when that saves words (G::PACKED).
// G (generated, namespace-free struct) provides:
//   constexpr int NV, NINST, MAXCH, PC_BASE, DONE, NINV, NCON; constexpr unsigned long long NUM_INIT
//   struct Cells { int32_t c0, c1, ... };   // the variable cells as NAMED members: nothing can index them with a variable, so nothing
//                                           // can push them out of the registers (see pcal_codegen.cpp CELL)
//   constexpr int NW (words of a stored row), VMW (words of the interpreter's row); constexpr bool PACKED
//   static void zero(Cells &), from_words(const uint64_t *, Cells &); static bool to_words(const Cells &, uint64_t *) — the STORED row: the
//   interpreter's layout, or (round 6, PACKED) every cell in the bits its inferred range needs (pcal_codegen.cpp "cell ranges"); false = a
//   value outside its cell's range (an evaluation error: never stored)
//   static void cells_to_vm(const Cells &, uint64_t *), cells_from_vm(const uint64_t *, Cells &) — the interpreter's row: what leaves the engine
//   template <class Ref> static int32_t pc_from_row(Ref row, int inst)
//   template <int INST> static int32_t pc_of(const Cells &)
//   static int run_init(uint64_t &ch, Cells &v)                                     // R_* of SpecVmT::Run
//   template <int INST> static int run_inst(int32_t label, uint64_t &ch, Cells &v, const Cells &old, int &aux)
//   static int run_inv(int k, Cells &v, int32_t &result)                            // INVARIANTs 0 .. NINV-1, then CONSTRAINTs
#pragma once
#include "spec_vm.h"

#ifndef MC_GEN_FP_SUM
#define MC_GEN_FP_SUM 0
#endif
#ifndef MC_GEN_KEY_BY_INST   // 0 = A/B: the by-pairs kernel's pairs sorted by label alone (round 6 until its last third)
#define MC_GEN_KEY_BY_INST 1
#endif

namespace mc {

// the by-pairs protocol's constants (engine_pairs.h), present only for programs the kernel's fixed sizes take: at most 128 slots (guard
// mask), at most 64 labels (one lane per key in the prefix sum)
template <class G, bool OK>
struct SpecGenPairs {};
template <class G>
struct SpecGenPairs<G, true> {
    // (a wavefront whose parents enable more pairs than its list holds works in rounds of 20 consecutive slots: 20 x 64 = the list)
    static constexpr int PAIR_FAMILIES = 1, TOTAL_SLOTS = G::NINST * G::MAXCH + 1, PAIR_ROUND_SLOTS = 20, PAIR_ROUNDS = (TOTAL_SLOTS + 19) / 20,
                         PAIR_KEYS = (MC_GEN_KEY_BY_INST && G::NKEYS <= 64) ? G::NKEYS : G::NLABELS, W_PAIR_BASE = -1;
};
template <class G>
constexpr bool spec_gen_pairs_ok() { return G::NINST * G::MAXCH + 1 <= 128 && G::NLABELS <= 64; }

template <class G>
struct SpecGenT : SpecGenPairs<G, spec_gen_pairs_ok<G>()> {
    using Params = VmParams;
    using VM = SpecVmT<128>;
    static constexpr int NV = G::NV, MAX_VARS = NV, MAX_WORDS = G::NW, FIX_SLOTS = 0, STAGE_WORDS = 0;
    static constexpr bool KEY_BY_INST = MC_GEN_KEY_BY_INST && G::NKEYS <= 64;   // pairs sorted by (instance, label): one copy of a label's code per batch
    static constexpr int EXPORT_WORDS = G::VMW;   // (engine.hip: rows leave the engine through export_row)
    static constexpr bool PACKED_ROWS = G::PACKED;
    static constexpr bool SLICE_SLOTS = true;
    MC_HD static int words(const Params &) { return MAX_WORDS; }
    MC_HD static int max_slots(const Params &) { return G::NINST * G::MAXCH + 1; }
    using Cells = typename G::Cells;
    struct Local { Cells v; };
    enum Run { R_DISABLED = 0, R_OK = 1, R_ASSERT = 2, R_ERROR = 3, R_OVERFLOW = 4 };

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        if (vm_make_params(p, np, o)) return -1;
        // the generated code IS a program: the one it was generated from
        if (o.nv != G::NV || o.ninst != G::NINST || o.maxch != G::MAXCH || o.pc_base != G::PC_BASE || o.done != G::DONE || o.ninv != G::NINV ||
            o.ncon != G::NCON || o.num_init != G::NUM_INIT)
            return -1;
        return 0;
    }

    MC_HD static uint64_t fp_words(const uint64_t *w) {  // (SpecVmT::fp_vals over the stored row: the interpreter's fingerprints when the layout is the interpreter's)
#if MC_GEN_FP_SUM
        // packed rows: the fingerprint is the sum of one hmum term per stored word (salted by the word's index; independent terms, two
        // 32 x 32 -> 64 multiply-accumulates each: mc_common.h, the raft / SSI lowerings' H) and ONE fmix64 over the sum, instead of a chain
        // of MAX_WORDS dependent fmix64 rounds (six quarter-rate multiplies each).  The interpreter's layout keeps the interpreter's fingerprints.
        if constexpr (G::PACKED) {
            uint64_t h = 0;
#pragma unroll
            for (int k = 0; k < MAX_WORDS; ++k) h += hmum(w[k], 0x632be59bd9b4e019ull * (uint64_t)(k + 1));
            return fp_nonzero(fmix64(h ^ 0x9e3779b97f4a7c15ull));
        }
#endif
        uint64_t h = 0x9e3779b97f4a7c15ull;
#pragma unroll
        for (int k = 0; k < MAX_WORDS; ++k) h = fmix64(h ^ (w[k] + 0x632be59bd9b4e019ull * (uint64_t)(k + 1)));
        return fp_nonzero(h);
    }
    template <class Ref>
    MC_HD static void unpack(Ref s, Cells &v) {
        uint64_t w[MAX_WORDS];
#pragma unroll
        for (int k = 0; k < MAX_WORDS; ++k) w[k] = s.get(k);
        G::from_words(w, v);
    }
    MC_HD static unsigned inv_status(Cells &v) {
        unsigned st = 0;
#pragma unroll
        for (int k = 0; k < G::NINV; ++k) {
            if (st) break;
            int32_t res = 0;
            const int r = G::run_inv(k, v, res);
            if (r != R_OK) return ST_SPECERR;
            if (!res) st = ST_INVARIANT | (unsigned)k << 8;
        }
#pragma unroll
        for (int k = G::NINV; k < G::NINV + G::NCON; ++k) {
            int32_t res = 0;
            const int r = G::run_inv(k, v, res);
            if (r != R_OK) return ST_SPECERR;
            if (!res) return st | ST_OUT_OF_MODEL;
        }
        return st;
    }

    MC_HD static uint64_t num_init(const Params &) { return G::NUM_INIT; }
    MC_HD static void init(const Params &, uint64_t k, WordRef out) {
        Cells v;
        G::zero(v);
        uint64_t ch = k;
        G::run_init(ch, v);
        uint64_t w[MAX_WORDS];
        (void)G::to_words(v, w);   // (init_status below re-checks the stored row; a range miss would show as a state that is not the interpreter's)
#pragma unroll
        for (int i = 0; i < MAX_WORDS; ++i) out.set(i, w[i]);
    }
    MC_HD static uint64_t fp_of(const Params &, CWordRef s) {
        uint64_t w[MAX_WORDS];
#pragma unroll
        for (int k = 0; k < MAX_WORDS; ++k) w[k] = s.get(k);
        return fp_words(w);
    }
    MC_HD static unsigned init_status(const Params &, CWordRef s) {
        Cells v;
        unpack(s, v);
        return ST_ENABLED | inv_status(v);
    }
    template <class Ref>
    MC_HD static void load(const Params &, Ref s, Local &l) { unpack(s, l.v); }
    MC_HD static int nslots(const Params &, const Local &) { return G::NINST * G::MAXCH + 1; }
    template <class Ref>
    MC_HD static unsigned parent_status(const Params &, const Local &, Ref) { return 0; }

    // the instance's label and its code, by a chain over the (few) instances: every cell access inside is a named member
    template <int I>
    MC_HD static int dispatch_inst(int inst, uint64_t &ch, Cells &v, const Cells &cur, int &aux) {
        if constexpr (I < G::NINST) {
            if (inst == I) {
                const int32_t label = G::template pc_of<I>(cur);
                if (label == G::DONE) return R_DISABLED;
                return G::template run_inst<I>(label, ch, v, cur, aux);
            }
            return dispatch_inst<I + 1>(inst, ch, v, cur, aux);
        } else {
            return R_DISABLED;
        }
    }
    template <int I>
    MC_HD static bool all_done(const Cells &cur) {
        if constexpr (I < G::NINST) return G::template pc_of<I>(cur) == G::DONE && all_done<I + 1>(cur);
        else return true;
    }
    // successor of the state `cur` through `slot`, left in v; returns the ST_* status (SpecVmT::step)
    MC_HD static unsigned step(const Cells &cur, int slot, Cells &v) {
        v = cur;
        constexpr int last = G::NINST * G::MAXCH;
        if (slot == last) return all_done<0>(cur) ? (unsigned)ST_ENABLED : 0u;  // (\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars
        if (slot < 0 || slot > last) return 0;
        const int inst = slot / G::MAXCH;
        uint64_t ch = (uint64_t)(slot % G::MAXCH);
        int aux = 0;
        int r = dispatch_inst<0>(inst, ch, v, cur, aux);
        // a path that ends early (failed assert, evaluation error) has not consumed all of its choice index: only the index whose
        // unconsumed remainder is 0 reports it (SpecVmT::run)
        if ((r == R_ASSERT || r == R_ERROR || r == R_OVERFLOW) && ch != 0) r = R_DISABLED;
        if (r == R_DISABLED) return 0;
        if (r == R_ASSERT) return ST_ENABLED | ST_ASSERT;
        if (r == R_ERROR) return ST_ENABLED | ST_SPECERR;
        if (r == R_OVERFLOW) return ST_ENABLED | ST_OVERFLOW;
        return ST_ENABLED | inv_status(v);
    }
    template <class Ref>
    MC_HD static unsigned eval(const Params &, const Local &l, Ref, int slot, uint64_t &fp) {
        Cells v;
        const unsigned st = step(l.v, slot, v);
        if (st & ST_ENABLED) {
            uint64_t w[MAX_WORDS];
            if (!G::to_words(v, w)) return ST_ENABLED | ST_SPECERR;   // outside an inferred range: reported, never stored
            fp = fp_words(w);
        }
        return st;
    }
    template <class Ref>
    MC_HD static unsigned apply(const Params &, Ref s, int slot, WordRef out) {
        Cells cur, v;
        unpack(s, cur);
        const unsigned st = step(cur, slot, v);
        uint64_t w[MAX_WORDS];
        (void)G::to_words(v, w);   // (eval decided that this successor is stored)
#pragma unroll
        for (int k = 0; k < MAX_WORDS; ++k) out.set(k, w[k]);
        return st;
    }

    // ---- by-pairs protocol (engine_pairs.h, dynamic keys): the wavefront's enabled (parent, slot) pairs are sorted by the LABEL the slot's
    // process instance stands at, so that a batch of 64 pairs runs one label's code
    struct Summary {};
    MC_HD static void summarize(const Local &, Summary &) {}
    struct SlotMask { uint64_t lo, hi; };
    MC_HD static constexpr SlotMask round_mask(int r) {  // round r: slots 20 r .. 20 r + 19
        SlotMask m{0, 0};
        for (int s = 20 * r; s < 20 * r + 20 && s < G::NINST * G::MAXCH + 1; s++) { if (s < 64) m.lo |= 1ull << s; else m.hi |= 1ull << (s - 64); }
        return m;
    }
    template <int I>
    MC_HD static void guards_inst(const Cells &cur, uint64_t &lo, uint64_t &hi) {
        if constexpr (I < G::NINST) {
            const int32_t label = G::template pc_of<I>(cur);
            if (label != G::DONE) {
                const int n = G::nch(label);   // choices the label's longest path consumes: an index beyond it cannot be fully consumed
#pragma unroll
                for (int c = 0; c < G::MAXCH; ++c)
                    if (c < n) { const int s = I * G::MAXCH + c; if (s < 64) lo |= 1ull << s; else hi |= 1ull << (s - 64); }
            }
            guards_inst<I + 1>(cur, lo, hi);
        }
    }
    // bit s set <= slot s may be enabled (an `await` or a shorter path can still disable it: the evaluation says so)
    MC_HD static void guards(const Params &, const Local &l, uint64_t &lo, uint64_t &hi) {
        lo = hi = 0;
        guards_inst<0>(l.v, lo, hi);
        constexpr int last = G::NINST * G::MAXCH;
        if (all_done<0>(l.v)) { if (last < 64) lo |= 1ull << last; else hi |= 1ull << (last - 64); }
    }
    MC_HD static bool guard_is_exact(int) { return false; }
    // the label slot's instance stands at, read from the parent's packed row (cell PC_BASE + inst: half of word (PC_BASE + inst) / 2)
    template <class Ref>
    MC_HD static int pair_key(const Params &, Ref row, int slot) {
        if constexpr (KEY_BY_INST) {
            if (slot >= G::NINST * G::MAXCH) return G::NKEYS - 1;
            const int inst = slot / G::MAXCH;
            return G::key_of(inst, G::pc_from_row(row, inst));
        } else {
            if (slot >= G::NINST * G::MAXCH) return G::DONE;
            const int32_t label = G::pc_from_row(row, slot / G::MAXCH);
            return label >= 0 && label < G::NLABELS ? label : G::DONE;
        }
    }
    struct PairOut { uint64_t w[MAX_WORDS]; };   // the successor's packed row
    template <int F, class Ref>
    MC_HD static unsigned eval_pair(const Params &, const Summary &, Ref row, int slot, uint64_t &fp, PairOut &o) {
        Cells cur, v;
        unpack(row, cur);
        const unsigned st = step(cur, slot, v);
        if (st & ST_ENABLED) {
            if (!G::to_words(v, o.w)) return ST_ENABLED | ST_SPECERR;
            fp = fp_words(o.w);
        }
        return st;
    }
    template <class Ref>
    MC_HD static void write_pair(const Params &, Ref, const PairOut &o, WordRef out) {
#pragma unroll
        for (int k = 0; k < MAX_WORDS; ++k) out.set(k, o.w[k]);
    }

    // host side: the interpreter's helpers, on the interpreter's row
    MC_HD static void export_row(const Params &, const uint64_t *stored, uint64_t *vm_row) {
        Cells v;
        G::from_words(stored, v);
        G::cells_to_vm(v, vm_row);
    }
    static int action_of(const Params &p, const uint64_t *parent, int slot) {
        uint64_t vm[G::VMW];
        export_row(p, parent, vm);
        return VM::action_of(p, vm, slot);
    }
    static const char *action_name(int a) { return VM::action_name(a); }
    static int format(const Params &p, const uint64_t *w, char *buf, size_t cap) {
        uint64_t vm[G::VMW];
        export_row(p, w, vm);
        return VM::format(p, vm, buf, cap);
    }
};

}  // namespace mc
