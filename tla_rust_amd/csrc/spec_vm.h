// spec_vm.h — generic lowering for PlusCal algorithms compiled by pcal_compile.cpp: a small stack machine that
// every lane runs on its own state.  This is what lets `mc X.tla` check a PlusCal spec the hand-lowered
// registry does not know (the reference's README.md:26-42 roadmap is new PlusCal specs of lock-free
// algorithms); the two root specs of the reference (pcal_intro.tla, atomic_add.tla) compile through it too
// and are parity-tested against their hand lowerings (spec_pluscal.h) and against the TLA+ evaluator of the test suite.
//
// State: nv 32-bit variables (globals, pc per process instance, process locals per instance, arrays
// flattened), two per 64-bit word.  Strings (labels, string constants) are interned integers.
// Slots (the order of Next in the translation, p-manual p.63): one block of `maxch` slots per process
// instance — slot = instance * maxch + choice, where `choice` enumerates the either/with alternatives of the
// instance's current label — and a last slot for the terminating disjunct
// `(\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars`.
// Program image (one int32 array, device resident): header, label-entry table, instance tables, code.
#pragma once
#include "mc_common.h"
#include <stdio.h>

namespace mc {

enum VmOp : int32_t {
    VM_HALT = 0, VM_PUSH, VM_SELF, VM_LOAD, VM_LOADX, VM_STORE, VM_STOREX, VM_LOADT, VM_STORET,
    VM_ADD, VM_SUB, VM_MUL, VM_DIV, VM_MOD, VM_NEG, VM_EQ, VM_NE, VM_LT, VM_LE, VM_GT, VM_GE, VM_NOT,
    VM_JMP, VM_JZ, VM_JNZ, VM_CHOOSE, VM_AWAIT, VM_ASSERT, VM_SETPC, VM_FAIL, VM_POP, VM_NOP,
    // bounded sequences: cell `base` holds Len, cells base+1 .. base+cap the elements (unused cells are 0); operands: base, cap, indexed
    VM_LOADSEQ, VM_STORESEQ, VM_APPEND, VM_TAIL, VM_SEQCLR, VM_SEQCOPY,
    // sets of small naturals (0..31) as 32-bit masks
    VM_BIT, VM_OR, VM_AND, VM_ANDN, VM_POPCNT,
    // inside a defined operator the variables are the UNPRIMED ones: reads come from the state before the step
    VM_OLD_ON, VM_OLD_OFF,
    // ARRAYS of bounded sequences (`box = [p \in 1..N |-> <<>>]`: the channels of a message-passing algorithm): element k of the array is
    // the sequence at base + k * (cap + 1).  VM_SEQSEL replaces the array index on the stack by that offset; the sequence instruction that
    // follows pops it when its third operand is 1 (every sequence instruction has the operands base, cap, indexed)
    VM_SEQSEL, VM_SEQLEN,
    // SETS of RECORDS (a message soup: `msgs := msgs \cup {[type |-> "1a", bal |-> b]}`): cell `base` holds the number of elements, field f of
    // element i is cell base + 1 + f * cap + i; the elements are kept in ascending lexicographic order of their cells, without duplicates, so
    // that one set has one representation.  Operands: base, cap, k (fields); the k field values are on the stack, the last field on top
    VM_RSADD, VM_RSDEL, VM_RSHAS
};

// header words of the program image
enum VmHdr : int32_t {
    VMH_MAGIC = 0, VMH_NV, VMH_NINST, VMH_MAXCH, VMH_PC_BASE, VMH_DONE, VMH_INIT_ENTRY, VMH_NINV, VMH_INV0 /* .. +8 */,
    VMH_LABEL_TAB = VMH_INV0 + 8, VMH_SELF_TAB, VMH_NLABELS, VMH_NUM_INIT_LO, VMH_NUM_INIT_HI, VMH_CODE_LEN,
    VMH_NCON /* CONSTRAINTs: entries VMH_INV0 + ninv .. of the same table */, VMH_SIZE
};
constexpr int32_t VM_MAGIC = 0x70634c31;  // "pcL1"
constexpr int32_t VM_DEFAULT_INIT = -2147483641;  // the cell of a variable declared without an initial value (defaultInitValue)

// what the host keeps about a compiled program (names, types, source positions); defined in pcal_compile.cpp
int vm_format(const void *host, const int32_t *vals, char *buf, size_t cap);
int vm_action_of(const void *host, const int32_t *parent_vals, int slot);
const char *vm_action_name(const void *host, int action);
int vm_failed_assert(const void *host, const int32_t *vals, int *label);

struct VmParams {
    const int32_t *code;   // program image: host memory in host builds, device memory inside Engine<SpecVm>
    const void *host;      // pcal::Program (host only)
    int nv, words, ninst, maxch, pc_base, done, init_entry, ninv, ncon, label_tab, self_tab, code_len;
    int inv_entry[8];      // ninv INVARIANTs, then ncon CONSTRAINTs
    uint64_t num_init;
};

int vm_make_params(const int64_t *p, unsigned np, VmParams &o);  // pcal_compile.cpp (host)

// MAXV = capacity of the per-lane variable arrays.  The interpreter indexes them dynamically, so they live in
// scratch (private) memory; four instantiations (16 / 32 / 64 / 128 cells) keep small programs under the scratch size
// above which the runtime allocates scratch per dispatch (528 B/lane at 64 cells made every launch ~150 us).
template <int MAXV>
struct SpecVmT {
    using Params = VmParams;
    static constexpr bool IS_VM = true;
    static constexpr int MAX_VARS = MAXV, MAX_WORDS = MAX_VARS / 2, FIX_SLOTS = 0, STAGE_WORDS = 0;
    static constexpr bool SLICE_SLOTS = true;  // a slot = one interpreted action instance: small frontiers are sliced by slot (engine.hip)
    static constexpr int STACK = 16, TEMPS = 8;  // pcal_compile.cpp checks both bounds when it emits code
    MC_HD static int words(const Params &p) { return p.words; }
    MC_HD static int max_slots(const Params &p) { return p.ninst * p.maxch + 1; }
    struct Local { int32_t v[MAX_VARS]; };

    static int make_params(const int64_t *p, unsigned np, Params &o) { return vm_make_params(p, np, o); }

    enum Run { R_DISABLED = 0, R_OK = 1, R_ASSERT = 2, R_ERROR = 3, R_OVERFLOW = 4 };

    // run the code at `entry` on the variables v[]; `ch` = choice index consumed by VM_CHOOSE; top of stack at HALT
    // is returned through `result` (invariants); `aux` = assertion id on R_ASSERT
    // A path that ends early (failed assert, evaluation error) has not consumed all of its choice index: only the
    // index whose unconsumed remainder is 0 reports it, the others would enumerate the same path again.
    MC_HD static int run(const Params &p, int entry, int32_t self, int inst, uint64_t ch, int32_t *v, int32_t &result, int &aux,
                         const int32_t *old = nullptr) {
        uint64_t rest = ch;
        const int r = run_raw(p, entry, self, inst, rest, v, result, aux, old ? old : v);
        return (r == R_ASSERT || r == R_ERROR || r == R_OVERFLOW) && rest != 0 ? (int)R_DISABLED : r;
    }
    MC_HD static int run_raw(const Params &p, int entry, int32_t self, int inst, uint64_t &ch, int32_t *v, int32_t &result, int &aux,
                             const int32_t *old) {
        const int32_t *__restrict__ c = p.code;
        int32_t st[STACK], t[TEMPS];
        int sp = 0, pc = entry, old_depth = 0;
        const int32_t *rd = v;  // where variable READS come from: v, or `old` inside a defined operator
        result = 0;
        aux = 0;
        for (int steps = 0; steps < 100000; ++steps) {
            const int32_t op = c[pc++];
            switch (op) {
            case VM_HALT:
                result = sp > 0 ? st[sp - 1] : 0;
                return ch == 0 ? R_OK : R_DISABLED;  // an unconsumed choice index would enumerate a path twice
            case VM_PUSH: if (sp >= STACK) return R_ERROR; st[sp++] = c[pc++]; break;
            case VM_SELF: if (sp >= STACK) return R_ERROR; st[sp++] = self; break;
            case VM_LOAD: if (sp >= STACK) return R_ERROR; st[sp++] = rd[c[pc++]]; break;
            case VM_OLD_ON: if (++old_depth == 1) rd = old; break;
            case VM_OLD_OFF: if (--old_depth == 0) rd = v; break;
            case VM_LOADX: {  // base, lo, n : index on the stack
                const int32_t base = c[pc], lo = c[pc + 1], n = c[pc + 2];
                pc += 3;
                const int32_t i = st[sp - 1] - lo;
                if (i < 0 || i >= n) return R_ERROR;  // TLC: function applied outside its domain
                st[sp - 1] = rd[base + i];
                break;
            }
            case VM_STORE: v[c[pc++]] = st[--sp]; break;
            case VM_STOREX: {  // base, lo, n : value on top, index below
                const int32_t base = c[pc], lo = c[pc + 1], n = c[pc + 2];
                pc += 3;
                const int32_t val = st[--sp], i = st[--sp] - lo;
                if (i < 0 || i >= n) return R_ERROR;
                v[base + i] = val;
                break;
            }
            case VM_LOADT: if (sp >= STACK) return R_ERROR; st[sp++] = t[c[pc++]]; break;
            case VM_STORET: t[c[pc++]] = st[--sp]; break;
            // TLC's integers are 32-bit and it reports an overflow instead of wrapping: so does the program (R_ERROR)
            case VM_ADD: { --sp; int32_t r; if (__builtin_add_overflow(st[sp - 1], st[sp], &r)) return R_ERROR; st[sp - 1] = r; break; }
            case VM_SUB: { --sp; int32_t r; if (__builtin_sub_overflow(st[sp - 1], st[sp], &r)) return R_ERROR; st[sp - 1] = r; break; }
            case VM_MUL: { --sp; int32_t r; if (__builtin_mul_overflow(st[sp - 1], st[sp], &r)) return R_ERROR; st[sp - 1] = r; break; }
            case VM_DIV: {  // TLA+ \div rounds towards minus infinity, for a divisor of either sign; a \div 0 is an error
                --sp;
                const int32_t a = st[sp - 1], b = st[sp];
                if (b == 0 || (a == INT32_MIN && b == -1)) return R_ERROR;
                int32_t q = a / b;
                if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
                st[sp - 1] = q;
                break;
            }
            case VM_MOD: {
                --sp;
                const int32_t a = st[sp - 1], b = st[sp];
                if (b <= 0) return R_ERROR;
                int32_t r = a % b;
                if (r < 0) r += b;
                st[sp - 1] = r;
                break;
            }
            case VM_NEG: if (st[sp - 1] == INT32_MIN) return R_ERROR; st[sp - 1] = -st[sp - 1]; break;
            case VM_EQ: --sp; st[sp - 1] = st[sp - 1] == st[sp]; break;
            case VM_NE: --sp; st[sp - 1] = st[sp - 1] != st[sp]; break;
            case VM_LT: --sp; st[sp - 1] = st[sp - 1] < st[sp]; break;
            case VM_LE: --sp; st[sp - 1] = st[sp - 1] <= st[sp]; break;
            case VM_GT: --sp; st[sp - 1] = st[sp - 1] > st[sp]; break;
            case VM_GE: --sp; st[sp - 1] = st[sp - 1] >= st[sp]; break;
            case VM_NOT: st[sp - 1] = !st[sp - 1]; break;
            case VM_JMP: pc = c[pc]; break;
            case VM_JZ: { const int32_t a = c[pc++]; if (!st[--sp]) pc = a; break; }
            case VM_JNZ: { const int32_t a = c[pc++]; if (st[--sp]) pc = a; break; }
            case VM_CHOOSE: {  // n alternatives: push ch % n
                const uint64_t n = (uint64_t)c[pc++];
                if (sp >= STACK) return R_ERROR;
                st[sp++] = (int32_t)(ch % n);
                ch /= n;
                break;
            }
            case VM_AWAIT: if (!st[--sp]) return R_DISABLED; break;
            case VM_ASSERT: { const int32_t id = c[pc++]; if (!st[--sp]) { aux = id; return R_ASSERT; } break; }
            case VM_SETPC: v[p.pc_base + inst] = c[pc++]; break;
            case VM_POP: --sp; break;
            case VM_SEQSEL: {  // lo, n, stride : index on the stack
                const int32_t lo = c[pc], n = c[pc + 1], stride = c[pc + 2];
                pc += 3;
                const int32_t i = st[--sp] - lo;
                if (i < 0 || i >= n) return R_ERROR;  // TLC: function applied outside its domain
                st[sp++] = i * stride;   // ... consumed by the sequence instruction that follows (its third operand says so)
                break;
            }
            case VM_SEQLEN: {  // base, cap : Len of the selected sequence
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0);
                pc += 3;
                if (sp >= STACK) return R_ERROR;
                st[sp++] = rd[base];
                break;
            }
            case VM_LOADSEQ: {  // base, cap : 1-based index on the stack; Head(q) = q[1]
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0), cap = c[pc + 1];
                pc += 3;
                const int32_t i = st[sp - 1];
                if (i < 1 || i > rd[base] || i > cap) return R_ERROR;  // TLC: index outside 1..Len(q), Head(<<>>)
                st[sp - 1] = rd[base + i];
                break;
            }
            case VM_STORESEQ: {  // base, cap : value on top, index below
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0), cap = c[pc + 1];
                pc += 3;
                const int32_t val = st[--sp], i = st[--sp];
                if (i < 1 || i > v[base] || i > cap) return R_ERROR;
                v[base + i] = val;
                break;
            }
            case VM_APPEND: {  // base, cap : value on the stack
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0), cap = c[pc + 1];
                pc += 3;
                const int32_t val = st[--sp], n = v[base];
                if (n >= cap) return R_OVERFLOW;  // longer than the cells this program reserves: reported, never truncated
                v[base + 1 + n] = val;
                v[base] = n + 1;
                break;
            }
            case VM_TAIL: {  // base, cap
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0), cap = c[pc + 1];
                pc += 3;
                const int32_t n = v[base];
                if (n < 1) return R_ERROR;  // Tail(<<>>)
                for (int32_t k = 1; k < cap; ++k) v[base + k] = k < n ? v[base + k + 1] : 0;
                v[base + cap] = 0;
                v[base] = n - 1;
                break;
            }
            case VM_SEQCLR: {
                const int32_t base = c[pc] + (c[pc + 2] ? st[--sp] : 0), cap = c[pc + 1];
                pc += 3;
                for (int32_t k = 0; k <= cap; ++k) v[base + k] = 0;
                break;
            }
            case VM_BIT: {  // element -> singleton mask; only 0..31 can be a member of a set variable
                const int32_t x = st[sp - 1];
                if (x < 0 || x > 31) return R_OVERFLOW;
                st[sp - 1] = (int32_t)(1u << x);
                break;
            }
            case VM_OR: --sp; st[sp - 1] |= st[sp]; break;
            case VM_AND: --sp; st[sp - 1] &= st[sp]; break;
            case VM_ANDN: --sp; st[sp - 1] &= ~st[sp]; break;
            case VM_POPCNT: {
                uint32_t x = (uint32_t)st[sp - 1];
                int32_t n = 0;
                for (; x; x &= x - 1) ++n;
                st[sp - 1] = n;
                break;
            }
            case VM_SEQCOPY: {  // dst, src, cap
                const int32_t dst = c[pc], src = c[pc + 1], cap = c[pc + 2];
                pc += 3;
                for (int32_t k = 0; k <= cap; ++k) v[dst + k] = v[src + k];
                break;
            }
            case VM_RSADD: case VM_RSDEL: case VM_RSHAS: {
                const int32_t base = c[pc], cap = c[pc + 1], k = c[pc + 2];
                pc += 3;
                sp -= k;   // the element: st[sp .. sp + k)
                if (sp < 0) return R_ERROR;
                const int32_t *src = op == VM_RSHAS ? rd : v;
                const int32_t n = src[base];
                int32_t pos = 0, cmp = 1;   // first element that is not smaller; cmp == 0: it is equal
                for (; pos < n; ++pos) {
                    cmp = 0;
                    for (int32_t f = 0; f < k && cmp == 0; ++f) {
                        const int32_t a = src[base + 1 + f * cap + pos], b = st[sp + f];
                        cmp = a < b ? -1 : a > b ? 1 : 0;
                    }
                    if (cmp >= 0) break;
                }
                const bool found = pos < n && cmp == 0;
                if (op == VM_RSHAS) { st[sp++] = found; break; }
                if (op == VM_RSADD) {
                    if (found) break;
                    if (n >= cap) return R_OVERFLOW;  // more elements than the cells this program reserves: reported, never dropped
                    for (int32_t f = 0; f < k; ++f) {
                        for (int32_t i = n; i > pos; --i) v[base + 1 + f * cap + i] = v[base + 1 + f * cap + i - 1];
                        v[base + 1 + f * cap + pos] = st[sp + f];
                    }
                    v[base] = n + 1;
                } else if (found) {
                    for (int32_t f = 0; f < k; ++f) {
                        for (int32_t i = pos; i + 1 < n; ++i) v[base + 1 + f * cap + i] = v[base + 1 + f * cap + i + 1];
                        v[base + 1 + f * cap + n - 1] = 0;   // unused cells are 0: one set, one representation
                    }
                    v[base] = n - 1;
                }
                break;
            }
            case VM_NOP: break;
            case VM_FAIL:
            default: return R_ERROR;
            }
            if (sp < 0) return R_ERROR;
        }
        return R_ERROR;
    }

    MC_HD static uint64_t pack(const int32_t *v, int w, int nv) {
        const uint32_t lo = (uint32_t)v[2 * w], hi = 2 * w + 1 < nv ? (uint32_t)v[2 * w + 1] : 0u;
        return (uint64_t)lo | (uint64_t)hi << 32;
    }
    MC_HD static uint64_t fp_vals(const Params &p, const int32_t *v) {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (int w = 0; w < p.words; ++w) h = fmix64(h ^ (pack(v, w, p.nv) + 0x632be59bd9b4e019ull * (uint64_t)(w + 1)));
        return fp_nonzero(h);
    }
    template <class Ref>
    MC_HD static void unpack(const Params &p, Ref s, int32_t *v) {
        for (int w = 0; w < p.words; ++w) {
            const uint64_t x = s.get(w);
            v[2 * w] = (int32_t)(uint32_t)x;
            if (2 * w + 1 < p.nv) v[2 * w + 1] = (int32_t)(uint32_t)(x >> 32);
        }
    }
    // first violated invariant of the state v[] (or -1); R_ERROR inside an invariant counts as a spec error
    MC_HD static unsigned inv_status(const Params &p, int32_t *v) {
        unsigned st = 0;
        for (int k = 0; k < p.ninv && !st; ++k) {
            int32_t res;
            int aux;
            const int r = run(p, p.inv_entry[k], 0, 0, 0, v, res, aux);
            if (r != R_OK) return ST_SPECERR;
            if (!res) st = ST_INVARIANT | (unsigned)k << 8;
        }
        // cfg CONSTRAINT (FIFO/MCInnerFIFO.cfg:23-26, p-manual section 4.3 p.36): a state outside the constraint is generated and
        // invariant-checked like any other (a violation is reported), but neither stored nor expanded
        for (int k = p.ninv; k < p.ninv + p.ncon; ++k) {
            int32_t res;
            int aux;
            const int r = run(p, p.inv_entry[k], 0, 0, 0, v, res, aux);
            if (r != R_OK) return ST_SPECERR;
            if (!res) return st | ST_OUT_OF_MODEL;
        }
        return st;
    }

    MC_HD static uint64_t num_init(const Params &p) { return p.num_init; }
    MC_HD static void init(const Params &p, uint64_t k, WordRef out) {
        int32_t v[MAX_VARS];
        for (int i = 0; i < MAX_VARS; ++i) v[i] = 0;
        int32_t res;
        int aux;
        run(p, p.init_entry, 0, 0, k, v, res, aux);
        for (int w = 0; w < p.words; ++w) out.set(w, pack(v, w, p.nv));
    }
    MC_HD static uint64_t fp_of(const Params &p, CWordRef s) {
        int32_t v[MAX_VARS];
        unpack(p, s, v);
        return fp_vals(p, v);
    }
    MC_HD static unsigned init_status(const Params &p, CWordRef s) {
        int32_t v[MAX_VARS];
        unpack(p, s, v);
        return ST_ENABLED | inv_status(p, v);
    }
    template <class Ref>
    MC_HD static void load(const Params &p, Ref s, Local &l) { unpack(p, s, l.v); }
    MC_HD static int nslots(const Params &p, const Local &) { return p.ninst * p.maxch + 1; }
    template <class Ref>
    MC_HD static unsigned parent_status(const Params &, const Local &, Ref) { return 0; }

    // successor of the state `cur` through `slot`, left in v[]; returns the ST_* status
    MC_HD static unsigned step(const Params &p, const int32_t *cur, int slot, int32_t *v) {
        for (int i = 0; i < p.nv; ++i) v[i] = cur[i];
        const int last = p.ninst * p.maxch;
        if (slot == last) {  // (\A self \in ProcSet: pc[self] = "Done") /\ UNCHANGED vars
            for (int i = 0; i < p.ninst; ++i) if (cur[p.pc_base + i] != p.done) return 0;
            return ST_ENABLED;
        }
        if (slot < 0 || slot > last) return 0;
        const int inst = slot / p.maxch;
        const uint64_t ch = (uint64_t)(slot % p.maxch);
        const int32_t label = cur[p.pc_base + inst];
        if (label == p.done) return 0;
        int32_t res;
        int aux;
        const int r = run(p, p.code[p.label_tab + label], p.code[p.self_tab + inst], inst, ch, v, res, aux, cur);
        if (r == R_DISABLED) return 0;
        if (r == R_ASSERT) return ST_ENABLED | ST_ASSERT;
        if (r == R_ERROR) return ST_ENABLED | ST_SPECERR;
        if (r == R_OVERFLOW) return ST_ENABLED | ST_OVERFLOW;
        return ST_ENABLED | inv_status(p, v);
    }
    template <class Ref>
    MC_HD static unsigned eval(const Params &p, const Local &l, Ref, int slot, uint64_t &fp) {
        int32_t v[MAX_VARS];
        const unsigned st = step(p, l.v, slot, v);
        if (st & ST_ENABLED) fp = fp_vals(p, v);
        return st;
    }
    template <class Ref>
    MC_HD static unsigned apply(const Params &p, Ref s, int slot, WordRef out) {
        int32_t cur[MAX_VARS], v[MAX_VARS];
        unpack(p, s, cur);
        const unsigned st = step(p, cur, slot, v);
        for (int w = 0; w < p.words; ++w) out.set(w, pack(v, w, p.nv));
        return st;
    }

    static int action_of(const Params &p, const uint64_t *parent, int slot) {
        int32_t v[MAX_VARS];
        unpack(p, CWordRef{parent, 1}, v);
        return vm_action_of(p.host, v, slot);
    }
    static const char *action_name(int a) { return a < 0 ? "Initial predicate" : "?"; }  // names live in the program: mc_action_name
    static int format(const Params &p, const uint64_t *w, char *buf, size_t cap) {
        int32_t v[MAX_VARS];
        unpack(p, CWordRef{w, 1}, v);
        return vm_format(p.host, v, buf, cap);
    }
};
using SpecVm = SpecVmT<128>;   // host-side helpers and the widest instantiation
using SpecVm16 = SpecVmT<16>;
using SpecVm32 = SpecVmT<32>;
using SpecVm64 = SpecVmT<64>;

}  // namespace mc
