// spec_pluscal.h — device lowerings of the two root PlusCal specs of the reference.
//
//   atomic_add  : reference atomic_add.tla:4-23 (Increment :11-15, Check :17-21), generalised
//                 to N adders + 1 checker awaiting N.
//   pcal_intro  : reference pcal_intro.tla:4-23 (Transfer :11-15, C :16, MoneyInvariant :23);
//                 variant 1 = the README's version with labels A:/B: (README.md:232-236) whose
//                 failing TLC run is README.md:267-321.
// Translation shape per examples/p-manual.pdf App. B pp.60-64: one action per label guarded by
// pc[self], `await` as an enabling conjunct, `assert` as Assert(...), plus the
// "(\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars" disjunct (p.63).
// The translated modules these lowerings were written against are specs/atomic_add.tla and
// specs/pcal_intro.tla.
#pragma once
#include "mc_common.h"
#include <stdio.h>

namespace mc {

// ------------------------------------------------------------------------------------------
// atomic_add: one 64-bit word per state.
//   bits 0..N-1  pc[adder i] = "Done"      bit 56  pc[checker] = "Done"
//   bits 57..62  global_counter
// Slots, in the order of Next (Checker \/ \E self: AdderProc(self) \/ termination):
//   0 = Check, 1..N = Increment(self), N+1 = terminating stutter.
struct SpecAtomicAdd {
    struct Params { int n; };
    static constexpr int MAX_WORDS = 1, FIX_SLOTS = 0, STAGE_WORDS = 0;
    MC_HD static int words(const Params &) { return 1; }
    MC_HD static int max_slots(const Params &p) { return p.n + 2; }
    static constexpr uint64_t SALT = 0x5bd1e9955bd1e995ull;
    struct Local { uint64_t w; };

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        if (np < 1 || p[0] < 1 || p[0] > 56) return -1;
        o.n = (int)p[0];
        return 0;
    }
    MC_HD static uint64_t num_init(const Params &) { return 1; }
    MC_HD static void init(const Params &, uint64_t, WordRef out) { out.set(0, 0); }
    MC_HD static uint64_t fp_words(const Params &, uint64_t w) { return fp_nonzero(fmix64(w ^ SALT)); }
    MC_HD static uint64_t fp_of(const Params &p, CWordRef s) { return fp_words(p, s.get(0)); }
    MC_HD static unsigned init_status(const Params &, CWordRef) { return ST_ENABLED; }
    MC_HD static void load(const Params &, CWordRef s, Local &l) { l.w = s.get(0); }
    MC_HD static int nslots(const Params &p, const Local &) { return p.n + 2; }
    MC_HD static unsigned parent_status(const Params &, const Local &, CWordRef) { return 0; }

    MC_HD static bool step(const Params &p, uint64_t w, int slot, uint64_t &nw) {
        const int n = p.n;
        const uint64_t counter = (w >> 57) & 63;
        if (slot == 0) {  // Check: pc[N+1] = "Check" /\ global_counter = N   (atomic_add.tla:17-21)
            if ((w >> 56 & 1) || counter != (uint64_t)n) return false;
            nw = w | (1ull << 56);
            return true;
        }
        if (slot <= n) {  // Increment(self)   (atomic_add.tla:11-15)
            const uint64_t bit = 1ull << (slot - 1);
            if (w & bit) return false;
            nw = (w | bit) + (1ull << 57);
            return true;
        }
        // (\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars   (p-manual p.63)
        const uint64_t all = ((n == 64 ? ~0ull : (1ull << n) - 1ull)) | (1ull << 56);
        if ((w & all) != all) return false;
        nw = w;
        return true;
    }
    MC_HD static unsigned eval(const Params &p, const Local &l, CWordRef, int slot, uint64_t &fp) {
        uint64_t nw;
        if (!step(p, l.w, slot, nw)) return 0;
        fp = fp_words(p, nw);
        return ST_ENABLED;
    }
    MC_HD static unsigned apply(const Params &p, CWordRef s, int slot, WordRef out) {
        uint64_t nw = 0;
        const bool ok = step(p, s.get(0), slot, nw);
        out.set(0, nw);
        return ok ? (unsigned)ST_ENABLED : 0u;
    }
    static int action_of(const Params &p, const uint64_t *, int slot) { return slot == 0 ? 1 : slot <= p.n ? 0 : 2; }
    static const char *action_name(int a) {
        static const char *nm[] = {"Increment", "Check", "Terminating"};
        return a >= 0 && a < 3 ? nm[a] : a < 0 ? "Initial predicate" : "?";
    }
    static int format(const Params &p, const uint64_t *w, char *buf, size_t cap) {
        size_t k = 0;
        k += snprintf(buf + k, cap - k, "/\\ global_counter = %d\n/\\ pc = <<", (int)((w[0] >> 57) & 63));
        for (int i = 0; i < p.n && k < cap; i++)
            k += snprintf(buf + k, cap - k, "%s\"%s\"", i ? ", " : "", (w[0] >> i & 1) ? "Done" : "Increment");
        if (k < cap) k += snprintf(buf + k, cap - k, ", \"%s\">>", (w[0] >> 56 & 1) ? "Done" : "Check");
        return (int)(k < cap ? k : cap);
    }
};

// ------------------------------------------------------------------------------------------
// pcal_intro: one 64-bit word per state.
//   bits 0..7 alice_account + 64   8..15 bob_account + 64   16..23 account_total
//   bits 24+3i..26+3i pc[i] (0 Transfer, 1 A, 2 B, 3 C, 4 Done)   36+6i..41+6i money[i]
// Slots: self = TransProc(self) for self < P (at most one label is enabled), P = termination.
struct SpecPcalIntro {
    struct Params { int variant, check_inv, max_money, nproc; };
    static constexpr int MAX_WORDS = 1, FIX_SLOTS = 0, STAGE_WORDS = 0;
    MC_HD static int words(const Params &) { return 1; }
    MC_HD static int max_slots(const Params &p) { return p.nproc + 1; }
    static constexpr uint64_t SALT = 0x27d4eb2f165667c5ull;
    enum { PC_TRANSFER = 0, PC_A = 1, PC_B = 2, PC_C = 3, PC_DONE = 4 };
    struct Local { uint64_t w; };

    static int make_params(const int64_t *p, unsigned np, Params &o) {
        o.variant = np > 0 ? (int)p[0] : 0;
        o.check_inv = np > 1 ? (int)p[1] : 1;
        o.max_money = np > 2 ? (int)p[2] : 20;
        o.nproc = np > 3 ? (int)p[3] : 2;
        if (o.variant < 0 || o.variant > 1 || o.nproc < 1 || o.nproc > 4 || o.max_money < 1 || o.max_money > 63 ||
            o.nproc * o.max_money > 70) return -1;
        return 0;
    }
    MC_HD static int alice(uint64_t w) { return (int)(w & 255) - 64; }
    MC_HD static int bob(uint64_t w) { return (int)(w >> 8 & 255) - 64; }
    MC_HD static int total(uint64_t w) { return (int)(w >> 16 & 255); }
    MC_HD static int pc(uint64_t w, int i) { return (int)(w >> (24 + 3 * i) & 7); }
    MC_HD static int money(uint64_t w, int i) { return (int)(w >> (36 + 6 * i) & 63); }
    MC_HD static uint64_t with_pc(uint64_t w, int i, int v) { return bits_set(w, 24 + 3 * i, 3, (uint64_t)v); }

    MC_HD static uint64_t num_init(const Params &p) {
        uint64_t n = 1;
        for (int i = 0; i < p.nproc; i++) n *= (uint64_t)p.max_money;
        return n;
    }
    // Init (pcal_intro.tla:5-9): alice = bob = 10, total = alice + bob, money \in [1..P -> 1..MaxMoney]
    MC_HD static void init(const Params &p, uint64_t k, WordRef out) {
        uint64_t w = (10 + 64) | ((uint64_t)(10 + 64) << 8) | (20ull << 16);
        for (int i = p.nproc - 1; i >= 0; i--) {
            w |= (uint64_t)(1 + k % (uint64_t)p.max_money) << (36 + 6 * i);
            k /= (uint64_t)p.max_money;
        }
        out.set(0, w);  // pc = [self \in ProcSet |-> "Transfer"] is all-zero
    }
    MC_HD static uint64_t fp_words(const Params &, uint64_t w) { return fp_nonzero(fmix64(w ^ SALT)); }
    MC_HD static uint64_t fp_of(const Params &p, CWordRef s) { return fp_words(p, s.get(0)); }
    // MoneyInvariant == alice_account + bob_account = account_total   (pcal_intro.tla:23)
    MC_HD static bool inv_ok(const Params &p, uint64_t w) { return !p.check_inv || alice(w) + bob(w) == total(w); }
    MC_HD static unsigned init_status(const Params &p, CWordRef s) {
        return ST_ENABLED | (inv_ok(p, s.get(0)) ? 0u : ST_INVARIANT);
    }
    MC_HD static void load(const Params &, CWordRef s, Local &l) { l.w = s.get(0); }
    MC_HD static int nslots(const Params &p, const Local &) { return p.nproc + 1; }
    MC_HD static unsigned parent_status(const Params &, const Local &, CWordRef) { return 0; }

    MC_HD static unsigned step(const Params &p, uint64_t w, int slot, uint64_t &nw) {
        if (slot == p.nproc) {  // termination disjunct (p-manual p.63)
            for (int i = 0; i < p.nproc; i++)
                if (pc(w, i) != PC_DONE) return 0;
            nw = w;
            return ST_ENABLED;
        }
        const int self = slot, a = alice(w), b = bob(w), m = money(w, self);
        switch (pc(w, self)) {
        case PC_TRANSFER:
            if (p.variant == 0) {  // pcal_intro.tla:11-15, one atomic step
                nw = w;
                if (a >= m) nw = (w & ~0xffffull) | (uint64_t)(a - m + 64) | ((uint64_t)(b + m + 64) << 8);
                nw = with_pc(nw, self, PC_C);
            } else {  // README.md:232-236: only the test; A and B are separate labels
                nw = with_pc(w, self, a >= m ? PC_A : PC_C);
            }
            return ST_ENABLED;
        case PC_A:
            nw = with_pc((w & ~0xffull) | (uint64_t)(a - m + 64), self, PC_B);
            return ST_ENABLED;
        case PC_B:
            nw = with_pc((w & ~0xff00ull) | ((uint64_t)(b + m + 64) << 8), self, PC_C);
            return ST_ENABLED;
        case PC_C:  // Assert(alice_account >= 0, "Failure of assertion at line 16, column 4.")
            nw = with_pc(w, self, PC_DONE);
            return ST_ENABLED | (a >= 0 ? 0u : ST_ASSERT);
        default:
            return 0;
        }
    }
    MC_HD static unsigned eval(const Params &p, const Local &l, CWordRef, int slot, uint64_t &fp) {
        uint64_t nw;
        unsigned st = step(p, l.w, slot, nw);
        if (!st) return 0;
        if (!(st & ST_ASSERT) && !inv_ok(p, nw)) st |= ST_INVARIANT;
        fp = fp_words(p, nw);
        return st;
    }
    MC_HD static unsigned apply(const Params &p, CWordRef s, int slot, WordRef out) {
        uint64_t nw = 0;
        const unsigned st = step(p, s.get(0), slot, nw);
        out.set(0, nw);
        return st;
    }
    // action family of the successor produced from `parent` by `slot`
    static int action_of(const Params &p, const uint64_t *parent, int slot) {
        if (slot == p.nproc) return 4;
        const int c = pc(parent[0], slot);
        return c == PC_TRANSFER ? 0 : c == PC_A ? 1 : c == PC_B ? 2 : 3;
    }
    static const char *action_name(int a) {
        static const char *nm[] = {"Transfer", "A", "B", "C", "Terminating"};
        return a >= 0 && a < 5 ? nm[a] : a < 0 ? "Initial predicate" : "?";
    }
    static int format(const Params &p, const uint64_t *ws, char *buf, size_t cap) {
        static const char *pcn[] = {"Transfer", "A", "B", "C", "Done", "?", "?", "?"};
        const uint64_t w = ws[0];
        size_t k = 0;
        k += snprintf(buf + k, cap - k, "/\\ alice_account = %d\n/\\ bob_account = %d\n/\\ account_total = %d\n/\\ pc = <<",
                      alice(w), bob(w), total(w));
        for (int i = 0; i < p.nproc && k < cap; i++) k += snprintf(buf + k, cap - k, "%s\"%s\"", i ? ", " : "", pcn[pc(w, i)]);
        if (k < cap) k += snprintf(buf + k, cap - k, ">>\n/\\ money = <<");
        for (int i = 0; i < p.nproc && k < cap; i++) k += snprintf(buf + k, cap - k, "%s%d", i ? ", " : "", money(w, i));
        if (k < cap) k += snprintf(buf + k, cap - k, ">>");
        return (int)(k < cap ? k : cap);
    }
};

}  // namespace mc
