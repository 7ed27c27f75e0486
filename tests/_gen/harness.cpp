// tests/_gen/harness.cpp — TEST-ONLY: the GENERATED lowering of a compiled PlusCal program (tla_rust_amd/csrc/pcal_codegen.cpp -> spec_gen.h)
// against the bytecode interpreter (spec_vm.h) on the host, state by state: a breadth-first search driven by the interpreter, and for
// EVERY reachable state and EVERY slot the two back-ends must agree on the status, the fingerprint and the successor's row; the
// initial states likewise.  The generated code may STORE its rows packed to the cells' inferred ranges (GenProg::PACKED): the search is driven
// by the interpreter on the interpreter's rows, every row is handed to the generated code through pack() — which also proves the range
// analysis on every reachable state: pack() followed by SpecGen::export_row must give the interpreter's row back (kind 9) — and every row
// the generated code produces is compared in both layouts (stored row == pack(interpreter's row), and its fingerprint == the fingerprint
// of that stored row).  Built per program by tests/helpers.py gen_harness (g++, the generated header given with -DGEN_HEADER), loaded
// beside libshim.so (which holds the PlusCal front-end the interpreter's host helpers live in).
#include GEN_HEADER
#include <stdint.h>
#include <string.h>
#include <unordered_set>
#include <vector>

using namespace mc;

struct GenCheck {
    uint64_t distinct, generated, mismatches, states_checked, pairs_checked, first_bad_state, first_bad_slot;
    uint32_t depth;
    int32_t first_bad_kind;   // 1 status, 2 fingerprint, 3 row, 4 init, 5 guards miss an enabled slot, 6 key out of range, 7 eval_pair, 8 write_pair,
                              // 9 a reachable state does not survive pack + export (a cell outside its inferred range),
                              // 10 two distinct states share a fingerprint of the generated code (or one state has two)
    uint32_t stored_words, vm_words;
};

template <class T>
static int nkeys() {
    if constexpr (spec_gen_pairs_ok<GenProg>()) return T::PAIR_KEYS;
    else return GenProg::NLABELS;
}
// the interpreter's row -> the row the generated code stores; false: a cell is outside the range the analysis inferred
static bool pack(const uint64_t *vm_row, uint64_t *stored) {
    GenProg::Cells v;
    GenProg::cells_from_vm(vm_row, v);
    return GenProg::to_words(v, stored);
}

extern "C" int gen_check(const void *program, uint64_t max_states, GenCheck *out) {
    memset(out, 0, sizeof *out);
    const int64_t handle = (int64_t)(intptr_t)program;
    VmParams p;
    if (SpecGen::make_params(&handle, 1, p)) return -1;   // (also: the generated constants are this program's)
    using VM = SpecVm;
    using GS = SpecGen;
    const int W = p.words, GW = SpecGen::MAX_WORDS;
    out->stored_words = (uint32_t)GW;
    out->vm_words = (uint32_t)W;
    auto bad = [&](int kind, uint64_t state, uint64_t slot) {
        if (!out->mismatches++) { out->first_bad_kind = kind; out->first_bad_state = state; out->first_bad_slot = slot; }
    };
    std::unordered_set<uint64_t> seen, gseen;   // the interpreter's fingerprints (they drive the search) / the generated code's
    std::vector<uint64_t> cur, next;
    uint64_t a[VM::MAX_WORDS], b[VM::MAX_WORDS], pa[VM::MAX_WORDS], back[VM::MAX_WORDS];
    // a == b in the layout the generated code stores (and, exported again, in the interpreter's)
    auto same_row = [&](const uint64_t *vm_row, const uint64_t *stored) {
        memset(pa, 0, sizeof pa);
        if (!pack(vm_row, pa) || memcmp(pa, stored, (size_t)GW * 8)) return false;
        memset(back, 0xee, sizeof back);
        GS::export_row(p, stored, back);
        return memcmp(back, vm_row, (size_t)W * 8) == 0;
    };
    for (uint64_t k = 0; k < VM::num_init(p); k++) {
        memset(a, 0, sizeof a);
        memset(b, 0xff, sizeof b);
        VM::init(p, k, WordRef{a, 1});
        GS::init(p, k, WordRef{b, 1});
        if (!same_row(a, b) || VM::init_status(p, CWordRef{a, 1}) != GS::init_status(p, CWordRef{b, 1}) || (!GenProg::PACKED && VM::fp_of(p, CWordRef{a, 1}) != GS::fp_of(p, CWordRef{b, 1})))
            bad(4, k, 0);
        out->generated++;
        if (VM::init_status(p, CWordRef{a, 1}) & ST_OUT_OF_MODEL) continue;
        if (seen.insert(VM::fp_of(p, CWordRef{a, 1})).second) {
            next.insert(next.end(), a, a + W);
            out->distinct++;
            if (!gseen.insert(GS::fp_of(p, CWordRef{b, 1})).second) bad(10, k, 0);
        } else if (gseen.insert(GS::fp_of(p, CWordRef{b, 1})).second) bad(10, k, 0);
    }
    cur.swap(next);
    uint32_t level = 1;
    const int ns = p.ninst * p.maxch + 1;
    while (!cur.empty() && (!max_states || out->distinct < max_states)) {
        const uint64_t n = cur.size() / (size_t)W;
        for (uint64_t i = 0; i < n; i++) {
            const CWordRef s{&cur[i * (size_t)W], 1};
            uint64_t ps[VM::MAX_WORDS];   // the same state as the generated code stores it
            memset(ps, 0, sizeof ps);
            out->states_checked++;
            if (!pack(&cur[i * (size_t)W], ps)) { bad(9, out->states_checked - 1, 0); continue; }
            memset(back, 0xee, sizeof back);
            GS::export_row(p, ps, back);
            if (memcmp(back, &cur[i * (size_t)W], (size_t)W * 8)) { bad(9, out->states_checked - 1, 0); continue; }
            const CWordRef gs{ps, 1};
            VM::Local lv;
            GS::Local lg;
            VM::load(p, s, lv);
            GS::load(p, gs, lg);
            uint64_t glo = 0, ghi = 0;   // the by-pairs protocol of the generated spec (engine_pairs.h): guards, key, eval_pair, write_pair
            constexpr bool PAIRS = spec_gen_pairs_ok<GenProg>();   // (programs beyond the kernel's fixed sizes keep the slot-by-slot kernel)
            if (PAIRS) GS::guards(p, lg, glo, ghi);
            for (int slot = 0; slot < ns; slot++) {
                uint64_t f0 = 0, f1 = 0;
                const unsigned s0 = VM::eval(p, lv, s, slot, f0), s1 = GS::eval(p, lg, gs, slot, f1);
                out->pairs_checked++;
                if (PAIRS) {
                    const bool g = slot < 64 ? (glo >> slot & 1u) : (ghi >> (slot - 64) & 1u);
                    if ((s0 & ST_ENABLED) && !g) bad(5, out->states_checked - 1, (uint64_t)slot);   // an enabled slot the guards miss: a lost successor
                    const int key = GS::pair_key(p, gs, slot);
                    if (key < 0 || key >= nkeys<GS>()) bad(6, out->states_checked - 1, (uint64_t)slot);
                    uint64_t f2 = 0;
                    GS::PairOut po;
                    const unsigned s2 = GS::eval_pair<0>(p, GS::Summary{}, gs, slot, f2, po);
                    if (s2 != s0 || ((s0 & ST_ENABLED) && f2 != f1)) bad(7, out->states_checked - 1, (uint64_t)slot);
                    if ((s0 & ST_ENABLED) && !(s0 & (ST_ASSERT | ST_SPECERR | ST_OVERFLOW))) {
                        memset(a, 0, sizeof a);
                        memset(b, 0xff, sizeof b);
                        VM::apply(p, s, slot, WordRef{a, 1});
                        GS::write_pair(p, gs, po, WordRef{b, 1});
                        if (!same_row(a, b) || GS::fp_of(p, CWordRef{b, 1}) != f2) bad(8, out->states_checked - 1, (uint64_t)slot);
                    }
                }
                if (s0 != s1) { bad(1, out->states_checked - 1, (uint64_t)slot); continue; }
                if (!(s0 & ST_ENABLED)) continue;
                out->generated++;
                if (!GenProg::PACKED && f0 != f1) bad(2, out->states_checked - 1, (uint64_t)slot);   // (the interpreter's layout: the interpreter's fingerprints)
                if (s0 & (ST_ASSERT | ST_SPECERR | ST_OVERFLOW)) continue;
                memset(a, 0, sizeof a);
                memset(b, 0xff, sizeof b);
                VM::apply(p, s, slot, WordRef{a, 1});
                GS::apply(p, gs, slot, WordRef{b, 1});
                // (out-of-model successors are compared too: never stored, but evaluated — their rows must pack)
                if (!same_row(a, b) || GS::fp_of(p, CWordRef{b, 1}) != f1) bad(f0 != f1 && !GenProg::PACKED ? 2 : 3, out->states_checked - 1, (uint64_t)slot);
                if (s0 & ST_OUT_OF_MODEL) continue;
                if (seen.insert(f0).second) {
                    next.insert(next.end(), a, a + W);
                    out->distinct++;
                    if (!gseen.insert(f1).second) bad(10, out->states_checked - 1, (uint64_t)slot);
                } else if (gseen.insert(f1).second) bad(10, out->states_checked - 1, (uint64_t)slot);
            }
        }
        cur.clear();
        cur.swap(next);
        if (!cur.empty()) level++;
    }
    out->depth = level;
    return 0;
}
