// pcal_codegen.cpp — a compiled PlusCal program (pcal_compile.cpp: the bytecode image the interpreter of spec_vm.h runs) translated into
// straight-line C++ for spec_gen.h, and the load-time build of that text into a gfx950 engine (north_star: "lowering each spec's
// next-state relation to a fixed-width packed state vector so that successor generation runs as a ... HIP kernel"; VERDICT round 5,
// next 3: "stop interpreting on the device").
//
// The translation is a static recompilation of the stack code: from every entry point (the labels of the algorithm, the enumeration of
// the initial states, the invariants and constraints) the reachable instructions are walked once with an ABSTRACT stack — the depth is a
// compile-time fact of well-formed code, so stack slot k becomes the C++ local s<k> and a temporary the local t<k> —, every jump target
// becomes a C++ label, every instruction one or two statements.  What the interpreter indexes at run time is made static: a label
// function is a template over the process INSTANCE (so `pc[self]` is the cell PC_BASE + INST and `self` a constant), an array or a
// sequence indexed by a run-time value becomes a chain of selects over its (small, constant) extent, an array of sequences a switch
// over the selected element.  The variable cells v[0 .. NV) are then only ever indexed by constants: after inlining, the compiler keeps
// them in registers (the interpreter's per-lane arrays live in scratch memory: dynamic indexing).
// Not translated (mc_program_codegen fails, the interpreter runs): sets of records (VM_RSADD / RSDEL / RSHAS).
#include "pcal.h"
#include "spec_vm.h"
#include "../../include/tlamc.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

extern "C" void mc_set_error_internal(const char *msg);   // engine.hip (the C ABI's last-error text)
namespace mc {
static void set_error(const std::string &m) { mc_set_error_internal(m.c_str()); }
}

namespace pcal {

struct GenError { std::string msg; };

namespace {

using namespace mc;

int operands(int op) {
    switch (op) {
    case VM_PUSH: case VM_LOAD: case VM_STORE: case VM_LOADT: case VM_STORET: case VM_JMP: case VM_JZ: case VM_JNZ: case VM_CHOOSE: case VM_ASSERT: case VM_SETPC: return 1;
    case VM_LOADX: case VM_STOREX: case VM_LOADSEQ: case VM_STORESEQ: case VM_APPEND: case VM_TAIL: case VM_SEQCLR: case VM_SEQCOPY: case VM_SEQSEL: case VM_SEQLEN:
    case VM_RSADD: case VM_RSDEL: case VM_RSHAS: return 3;
    default: return 0;
    }
}

struct AState {      // what is known at an instruction: stack depth, depth of "read the old variables" nesting, sequence selections on the stack
    int depth = 0, old = 0;
    std::map<int, std::pair<int, int>> sel;   // stack slot -> (n, stride) of the VM_SEQSEL that produced it
    bool operator==(const AState &o) const { return depth == o.depth && old == o.old && sel == o.sel; }
};

struct Gen {
    const std::vector<int> &c;
    std::ostringstream out;
    explicit Gen(const std::vector<int> &image) : c(image) {}
    [[noreturn]] void fail(const std::string &m) { throw GenError{m}; }

    static std::string S(int k) { return "s" + std::to_string(k); }
    // the cell array a READ goes to: the state being built, or the state before the step inside a defined operator
    static std::string RD(const AState &a) { return a.old > 0 ? "old" : "v"; }
    // arr[base + i] for a run-time i in [0, n): a chain of selects (every index a constant)
    // a variable cell: a NAMED member of the generated struct Cells (v.c12), not an array element — the compiler turned chains of selects
    // over neighbouring array elements back into ONE load with a computed index ("switch.lookup"), and an array that is indexed by a
    // variable anywhere lives in scratch memory everywhere (first device run: 565 GB of scratch traffic for 6.5 GB of states); a struct
    // member cannot be indexed by a variable
    static std::string CELL(const std::string &arr, int k) { return arr + ".c" + std::to_string(k); }
    static std::string chain_load(const std::string &arr, int base, int n, const std::string &i) {
        std::string e = CELL(arr, base + n - 1);
        for (int k = n - 2; k >= 0; --k) e = "(" + i + " == " + std::to_string(k) + " ? " + CELL(arr, base + k) + " : " + e + ")";
        return e;
    }
    static std::string chain_store(int base, int n, const std::string &i, const std::string &val) {
        std::string s;
        for (int k = 0; k < n; ++k) s += " " + CELL("v", base + k) + " = " + i + " == " + std::to_string(k) + " ? " + val + " : " + CELL("v", base + k) + ";";
        return s;
    }

    // one entry point -> one function body (without the signature); `inst_templ`: label code (SELF_ / INST are template constants)
    std::string body(int entry) {
        std::map<int, AState> at;     // state BEFORE each reachable instruction
        std::vector<int> work{entry};
        at[entry] = AState{};
        std::set<int> targets;
        auto flow = [&](int pc, const AState &st) {
            auto it = at.find(pc);
            if (it == at.end()) { at[pc] = st; work.push_back(pc); }
            else if (!(it->second == st)) fail("stack shapes differ at a join (pc " + std::to_string(pc) + ")");
        };
        int max_depth = 0, max_temp = -1;
        while (!work.empty()) {
            const int pc = work.back();
            work.pop_back();
            if (pc < 0 || pc >= (int)c.size()) fail("jump outside the image");
            AState a = at[pc];
            const int op = c[(size_t)pc], n = operands(op), next = pc + 1 + n;
            auto pop = [&](int k) { for (int j = 0; j < k; ++j) { a.depth--; a.sel.erase(a.depth); } if (a.depth < 0) fail("stack underflow"); };
            auto push = [&]() { a.sel.erase(a.depth); a.depth++; if (a.depth > 16) fail("stack deeper than 16"); };
            const int o0 = n > 0 ? c[(size_t)pc + 1] : 0, o2 = n > 2 ? c[(size_t)pc + 3] : 0;
            switch (op) {
            case VM_HALT: case VM_FAIL: continue;
            case VM_PUSH: case VM_SELF: case VM_LOAD: case VM_LOADT: case VM_CHOOSE: push(); break;
            case VM_STORE: case VM_AWAIT: case VM_ASSERT: case VM_POP: pop(1); break;
            case VM_STORET: pop(1); if (o0 > max_temp) max_temp = o0; break;
            case VM_LOADX: pop(1); push(); break;
            case VM_STOREX: pop(2); break;
            case VM_ADD: case VM_SUB: case VM_MUL: case VM_DIV: case VM_MOD: case VM_EQ: case VM_NE: case VM_LT: case VM_LE: case VM_GT: case VM_GE:
            case VM_OR: case VM_AND: case VM_ANDN: pop(2); push(); break;
            case VM_NEG: case VM_NOT: case VM_BIT: case VM_POPCNT: pop(1); push(); break;
            case VM_JMP: targets.insert(o0); flow(o0, a); continue;
            case VM_JZ: case VM_JNZ: pop(1); targets.insert(o0); flow(o0, a); break;
            case VM_SETPC: case VM_NOP: break;
            case VM_OLD_ON: a.old++; break;
            case VM_OLD_OFF: a.old--; if (a.old < 0) fail("unbalanced VM_OLD_OFF"); break;
            case VM_SEQSEL: {
                pop(1);
                const int slot = a.depth;
                push();
                a.sel[slot] = {c[(size_t)pc + 2], c[(size_t)pc + 3]};   // (n, stride)
                break;
            }
            case VM_SEQLEN: if (o2) { if (!a.sel.count(a.depth - 1)) fail("sequence selection lost"); pop(1); } push(); break;
            case VM_LOADSEQ: if (o2) { if (!a.sel.count(a.depth - 1)) fail("sequence selection lost"); pop(1); } pop(1); push(); break;
            case VM_STORESEQ: if (o2) { if (!a.sel.count(a.depth - 1)) fail("sequence selection lost"); pop(1); } pop(2); break;
            case VM_APPEND: if (o2) { if (!a.sel.count(a.depth - 1)) fail("sequence selection lost"); pop(1); } pop(1); break;
            case VM_TAIL: case VM_SEQCLR: if (o2) { if (!a.sel.count(a.depth - 1)) fail("sequence selection lost"); pop(1); } break;
            case VM_SEQCOPY: break;
            case VM_RSADD: case VM_RSDEL: case VM_RSHAS: fail("sets of records are not translated (VM_RSADD / VM_RSDEL / VM_RSHAS)");
            default: fail("unknown instruction " + std::to_string(op));
            }
            if (a.depth > max_depth) max_depth = a.depth;
            flow(next, a);
        }
        // ---- emission, in address order
        std::ostringstream b;
        b << "        int32_t";
        for (int k = 0; k < (max_depth > 0 ? max_depth : 1); ++k) b << (k ? ", " : " ") << "s" << k << " = 0";
        for (int k = 0; k <= max_temp; ++k) b << ", t" << k << " = 0";
        b << ";\n        (void)s0; (void)&old; (void)aux; (void)result;\n";
        int prev_next = entry;
        bool prev_falls = true;
        for (const auto &kv : at) {
            const int pc = kv.first;
            const AState &a = kv.second;
            const int op = c[(size_t)pc], n = operands(op), next = pc + 1 + n;
            if (prev_falls && pc != prev_next) b << "        goto L" << prev_next << ";\n";   // (the code between was not reachable from this entry)
            if (targets.count(pc) || pc == entry) b << "    L" << pc << ": ;\n";
            const int d = a.depth;
            const int o0 = n > 0 ? c[(size_t)pc + 1] : 0, o1 = n > 1 ? c[(size_t)pc + 2] : 0, o2 = n > 2 ? c[(size_t)pc + 3] : 0;
            const std::string top = S(d - 1), sec = S(d - 2), rd = RD(a);
            auto I = [](int v) { return std::to_string(v); };
            b << "        ";
            prev_falls = true;
            // a sequence instruction on element `sel` of an array of sequences: a switch over the element (constant bases inside)
            auto seq_op = [&](bool indexed, int selslot, int base, auto &&emit_one) {
                if (!indexed) { b << "{ " << emit_one(base) << " }"; return; }
                const auto sl = a.sel.at(selslot);
                b << "switch (" << S(selslot) << ") {";
                for (int k = 0; k < sl.first; ++k) b << " case " << I(k * sl.second) << ": { " << emit_one(base + k * sl.second) << " } break;";
                b << " default: return R_ERROR; }";
            };
            switch (op) {
            case VM_HALT: b << "result = " << (d > 0 ? top : std::string("0")) << "; return ch == 0 ? R_OK : R_DISABLED;"; prev_falls = false; break;
            case VM_FAIL: b << "return R_ERROR;"; prev_falls = false; break;
            case VM_PUSH: b << S(d) << " = " << I(o0) << ";"; break;
            case VM_SELF: b << S(d) << " = SELF_;"; break;
            case VM_LOAD: b << S(d) << " = " << CELL(rd, o0) << ";"; break;
            case VM_LOADT: b << S(d) << " = t" << I(o0) << ";"; break;
            case VM_STORET: b << "t" << I(o0) << " = " << top << ";"; break;
            case VM_STORE: b << CELL("v", o0) << " = " << top << ";"; break;
            case VM_LOADX: b << "{ const int32_t i_ = " << top << " - (" << I(o1) << "); if (i_ < 0 || i_ >= " << I(o2) << ") return R_ERROR; " << top << " = "
                             << chain_load(rd, o0, o2, "i_") << "; }"; break;
            case VM_STOREX: b << "{ const int32_t val_ = " << top << ", i_ = " << sec << " - (" << I(o1) << "); if (i_ < 0 || i_ >= " << I(o2) << ") return R_ERROR;"
                              << chain_store(o0, o2, "i_", "val_") << " }"; break;
            case VM_ADD: b << "{ int32_t r_; if (__builtin_add_overflow(" << sec << ", " << top << ", &r_)) return R_ERROR; " << sec << " = r_; }"; break;
            case VM_SUB: b << "{ int32_t r_; if (__builtin_sub_overflow(" << sec << ", " << top << ", &r_)) return R_ERROR; " << sec << " = r_; }"; break;
            case VM_MUL: b << "{ int32_t r_; if (__builtin_mul_overflow(" << sec << ", " << top << ", &r_)) return R_ERROR; " << sec << " = r_; }"; break;
            case VM_DIV: b << "{ const int32_t a_ = " << sec << ", b_ = " << top << "; if (b_ == 0 || (a_ == INT32_MIN && b_ == -1)) return R_ERROR; int32_t q_ = a_ / b_; "
                              "if ((a_ % b_ != 0) && ((a_ < 0) != (b_ < 0))) --q_; " << sec << " = q_; }"; break;
            case VM_MOD: b << "{ const int32_t a_ = " << sec << ", b_ = " << top << "; if (b_ <= 0) return R_ERROR; int32_t r_ = a_ % b_; if (r_ < 0) r_ += b_; " << sec << " = r_; }"; break;
            case VM_NEG: b << "if (" << top << " == INT32_MIN) return R_ERROR; " << top << " = -" << top << ";"; break;
            case VM_EQ: b << sec << " = " << sec << " == " << top << ";"; break;
            case VM_NE: b << sec << " = " << sec << " != " << top << ";"; break;
            case VM_LT: b << sec << " = " << sec << " < " << top << ";"; break;
            case VM_LE: b << sec << " = " << sec << " <= " << top << ";"; break;
            case VM_GT: b << sec << " = " << sec << " > " << top << ";"; break;
            case VM_GE: b << sec << " = " << sec << " >= " << top << ";"; break;
            case VM_NOT: b << top << " = !" << top << ";"; break;
            case VM_JMP: b << "goto L" << I(o0) << ";"; prev_falls = false; break;
            case VM_JZ: b << "if (!" << top << ") goto L" << I(o0) << ";"; break;
            case VM_JNZ: b << "if (" << top << ") goto L" << I(o0) << ";"; break;
            case VM_CHOOSE: b << S(d) << " = (int32_t)(ch % " << I(o0) << "ull); ch /= " << I(o0) << "ull;"; break;
            case VM_AWAIT: b << "if (!" << top << ") return R_DISABLED;"; break;
            case VM_ASSERT: b << "if (!" << top << ") { aux = " << I(o0) << "; return R_ASSERT; }"; break;
            case VM_SETPC: b << "pc_cell<INST>(v) = " << I(o0) << ";"; break;
            case VM_POP: case VM_NOP: case VM_OLD_ON: case VM_OLD_OFF: b << ";"; break;
            case VM_BIT: b << "if (" << top << " < 0 || " << top << " > 31) return R_OVERFLOW; " << top << " = (int32_t)(1u << " << top << ");"; break;
            case VM_OR: b << sec << " |= " << top << ";"; break;
            case VM_AND: b << sec << " &= " << top << ";"; break;
            case VM_ANDN: b << sec << " &= ~" << top << ";"; break;
            case VM_POPCNT: b << top << " = (int32_t)__builtin_popcount((unsigned)" << top << ");"; break;
            case VM_SEQSEL: b << "{ const int32_t i_ = " << top << " - (" << I(o0) << "); if (i_ < 0 || i_ >= " << I(o1) << ") return R_ERROR; " << top << " = i_ * " << I(o2) << "; }"; break;
            case VM_SEQLEN: {
                const int dst = o2 ? d - 1 : d;
                seq_op(o2 != 0, d - 1, o0, [&](int B) { return S(dst) + " = " + CELL(rd, B) + ";"; });
                break;
            }
            case VM_LOADSEQ: {
                const int idx = o2 ? d - 2 : d - 1;
                seq_op(o2 != 0, d - 1, o0, [&](int B) {
                    return "const int32_t i_ = " + S(idx) + "; if (i_ < 1 || i_ > " + CELL(rd, B) + " || i_ > " + I(o1) + ") return R_ERROR; " + S(idx) + " = " +
                           chain_load(rd, B + 1, o1, "(i_ - 1)") + ";";
                });
                break;
            }
            case VM_STORESEQ: {
                const int val = o2 ? d - 2 : d - 1, idx = val - 1;
                seq_op(o2 != 0, d - 1, o0, [&](int B) {
                    return "const int32_t val_ = " + S(val) + ", i_ = " + S(idx) + "; if (i_ < 1 || i_ > " + CELL("v", B) + " || i_ > " + I(o1) + ") return R_ERROR;" +
                           chain_store(B + 1, o1, "(i_ - 1)", "val_");
                });
                break;
            }
            case VM_APPEND: {
                const int val = o2 ? d - 2 : d - 1;
                seq_op(o2 != 0, d - 1, o0, [&](int B) {
                    return "const int32_t val_ = " + S(val) + ", n_ = " + CELL("v", B) + "; if (n_ >= " + I(o1) + ") return R_OVERFLOW;" + chain_store(B + 1, o1, "n_", "val_") +
                           " " + CELL("v", B) + " = n_ + 1;";
                });
                break;
            }
            case VM_TAIL:
                seq_op(o2 != 0, d - 1, o0, [&](int B) {
                    std::string s = "const int32_t n_ = " + CELL("v", B) + "; if (n_ < 1) return R_ERROR;";
                    for (int k = 1; k < o1; ++k) s += " " + CELL("v", B + k) + " = " + I(k) + " < n_ ? " + CELL("v", B + k + 1) + " : 0;";
                    s += " " + CELL("v", B + o1) + " = 0; " + CELL("v", B) + " = n_ - 1;";
                    return s;
                });
                break;
            case VM_SEQCLR:
                seq_op(o2 != 0, d - 1, o0, [&](int B) {
                    std::string s;
                    for (int k = 0; k <= o1; ++k) s += " " + CELL("v", B + k) + " = 0;";
                    return s;
                });
                break;
            case VM_SEQCOPY:
                for (int k = 0; k <= o2; ++k) b << " " << CELL("v", o0 + k) << " = " << CELL("v", o1 + k) << ";";
                break;
            default: fail("unknown instruction " + std::to_string(op));
            }
            b << "\n";
            prev_next = next;
        }
        if (prev_falls) b << "        return R_ERROR;\n";
        return b.str();
    }

    // labels instance `inst` can ever stand at: its first label (stored by the Init code) and everything VM_SETPC reaches from there
    // largest product of the VM_CHOOSE operands along a path from `entry` (capped at `cap`; a cycle counts as the cap)
    long long max_choices(int entry, long long cap) {
        std::map<int, long long> memo;
        std::set<int> on_path;
        std::function<long long(int)> go = [&](int pc) -> long long {
            if (pc < 0 || pc >= (int)c.size()) return cap;
            auto it = memo.find(pc);
            if (it != memo.end()) return it->second;
            if (!on_path.insert(pc).second) return cap;
            const int op = c[(size_t)pc], n = operands(op);
            long long r;
            if (op == VM_HALT || op == VM_FAIL) r = 1;
            else if (op == VM_JMP) r = go(c[(size_t)pc + 1]);
            else if (op == VM_JZ || op == VM_JNZ) r = std::max(go(c[(size_t)pc + 1]), go(pc + 1 + n));
            else if (op == VM_CHOOSE) r = std::min(cap, (long long)c[(size_t)pc + 1] * go(pc + 1 + n));
            else r = go(pc + 1 + n);
            on_path.erase(pc);
            if (r > cap) r = cap;
            memo[pc] = r;
            return r;
        };
        return go(entry);
    }
    int ninst_ = 0;
    std::set<int> labels_of(int inst, int pc_base, int init_entry, int label_tab, int nlabels) {
        int first = -1;
        for (int pc = init_entry; pc < (int)c.size();) {   // Init ends with  PUSH label; STORE pc_base + k  per instance, then HALT
            const int op = c[(size_t)pc];
            if (op == VM_HALT) break;
            if (op == VM_PUSH && pc + 3 < (int)c.size() && c[(size_t)pc + 2] == VM_STORE && c[(size_t)pc + 3] == pc_base + inst) first = c[(size_t)pc + 1];
            pc += 1 + operands(op);
        }
        if (first < 0) fail("no initial label for a process instance");
        std::set<int> seen{first};
        std::vector<int> todo{first};
        while (!todo.empty()) {
            const int l = todo.back();
            todo.pop_back();
            if (l < 0 || l >= nlabels) fail("label out of range");
            const int e = c[(size_t)(label_tab + l)];
            if (e < 0) continue;   // "Done"
            // every instruction reachable from the label's entry
            std::set<int> vis;
            std::vector<int> w{e};
            while (!w.empty()) {
                const int pc = w.back();
                w.pop_back();
                if (!vis.insert(pc).second) continue;
                const int op = c[(size_t)pc], n = operands(op);
                if (op == VM_SETPC && seen.insert(c[(size_t)pc + 1]).second) todo.push_back(c[(size_t)pc + 1]);
                if ((op == VM_STORE || op == VM_STOREX) && c[(size_t)pc + 1] >= pc_base && c[(size_t)pc + 1] < pc_base + ninst_)
                    fail("a label stores a computed value into pc");   // (the label sets per instance would not be closed)
                if (op == VM_HALT || op == VM_FAIL) continue;
                if (op == VM_JMP) { w.push_back(c[(size_t)pc + 1]); continue; }
                if (op == VM_JZ || op == VM_JNZ) w.push_back(c[(size_t)pc + 1]);
                w.push_back(pc + 1 + n);
            }
        }
        return seen;
    }
};


// ------------------------------------------------------------------------------------------------ cell ranges -> the PACKED row
// north_star: "a fixed-width packed state vector"; VERDICT round 5, next 3: "variables bit-packed to their inferred ranges (a pc in
// ceil(log2 labels) bits, booleans in 1)".  The interpreter keeps every cell in 32 bits, two per word; the generated code knows its program,
// so an interval analysis of the stack code (abstract interpretation: flow-sensitive for the stack and the temporaries, flow-INsensitive for
// the variable cells — one interval per cell that holds in every reachable state, the least fixed point over the initial-state code and every
// label's code, widened to the full 32 bits when a bound keeps moving: `x := x + 1` is not bounded by the test in front of it here) gives
// every cell a value set [lo, hi], optionally plus the one constant VM_DEFAULT_INIT (defaultInitValue: a huge negative number that would
// otherwise cost the cell its 32 bits).  A cell is stored as (value - lo) [+ 1 when code 0 means "default"] in ceil(log2) bits; the cells
// are laid into 64-bit words first-fit in order of decreasing width, none straddling a word; a cell with one possible value takes no bits.
// Sound by construction — every transfer function over-approximates, checks of the code (index in range, Len < cap) only ever narrow —
// and CHECKED where it costs nothing to be sure: tests/_gen/harness.cpp packs and unpacks every reachable state of every test program, and
// the generated to_words reports a value outside its cell's range (R_ERROR: an evaluation error, never a silently wrong state).
struct AV {   // abstract value: bot (no value yet) | [lo, hi] possibly with VM_DEFAULT_INIT beside it
    bool bot = true, dflt = false, num = false;
    int64_t lo = 0, hi = 0;
    static AV of(int64_t a, int64_t b) { AV v; v.bot = false; v.num = true; v.lo = a; v.hi = b; return v; }
    static AV top() { return of(INT32_MIN, INT32_MAX); }
    static AV cst(int64_t x) {
        if (x == (int64_t)VM_DEFAULT_INIT) { AV v; v.bot = false; v.dflt = true; return v; }
        return of(x, x);
    }
    bool operator==(const AV &o) const { return bot == o.bot && dflt == o.dflt && num == o.num && (!num || (lo == o.lo && hi == o.hi)); }
    // as plain numbers (arithmetic does not know the default apart)
    AV plain() const {
        if (bot) return *this;
        if (!dflt) return *this;
        AV v = of(num ? std::min<int64_t>(lo, VM_DEFAULT_INIT) : VM_DEFAULT_INIT, num ? std::max<int64_t>(hi, VM_DEFAULT_INIT) : VM_DEFAULT_INIT);
        return v;
    }
};
static AV av_join(const AV &a, const AV &b) {
    if (a.bot) return b;
    if (b.bot) return a;
    AV r;
    r.bot = false;
    r.dflt = a.dflt || b.dflt;
    r.num = a.num || b.num;
    if (a.num && b.num) { r.lo = std::min(a.lo, b.lo); r.hi = std::max(a.hi, b.hi); }
    else if (a.num) { r.lo = a.lo; r.hi = a.hi; }
    else if (b.num) { r.lo = b.lo; r.hi = b.hi; }
    if (r.num && r.dflt && r.lo <= (int64_t)VM_DEFAULT_INIT) r.dflt = false;   // the interval holds it anyway
    return r;
}
static AV av_clip(int64_t lo, int64_t hi) { return AV::of(std::max<int64_t>(lo, INT32_MIN), std::min<int64_t>(hi, INT32_MAX)); }

struct CellEnc { int bits = 32, word = 0, shift = 0; int64_t lo = INT32_MIN; bool dflt = false; uint64_t span = 0xffffffffull; /* largest code */ };

struct Ranges {
    const std::vector<int> &c;
    int nv, ninst, pc_base, self_tab;
    std::vector<AV> R;            // per cell
    std::vector<int> grow;        // how often a cell's set grew (widening)
    bool changed = false;
    std::vector<std::string> notes;
    Ranges(const std::vector<int> &image, int nv_, int ninst_, int pc_base_, int self_tab_) : c(image), nv(nv_), ninst(ninst_), pc_base(pc_base_), self_tab(self_tab_), R((size_t)nv_), grow((size_t)nv_, 0) {}
    [[noreturn]] void fail(const std::string &m) { throw GenError{"range analysis: " + m}; }

    void store(int cell, const AV &v) {
        if (cell < 0 || cell >= nv) fail("store outside the cells");
        if (v.bot) return;
        AV j = av_join(R[(size_t)cell], v);
        if (j == R[(size_t)cell]) return;
        if (++grow[(size_t)cell] > 12) j = AV::top();   // widening: a bound that keeps moving is not bounded here
        R[(size_t)cell] = j;
        changed = true;
    }
    AV load(int cell) const {
        if (cell < 0 || cell >= nv) return AV::top();
        return R[(size_t)cell];
    }
    struct Slot { AV v; int sel_n = 0, sel_stride = 0; bool operator==(const Slot &o) const { return v == o.v && sel_n == o.sel_n && sel_stride == o.sel_stride; } };
    struct FS { std::vector<Slot> st; AV t[16]; int visits = 0; };

    // one entry point; `selfs`: the values SELF can have here; `pcs`: the pc cells VM_SETPC writes here
    void run(int entry, const AV &self, const std::vector<int> &pcs) {
        std::map<int, FS> at;
        std::vector<int> work{entry};
        at[entry] = FS{};
        auto flow = [&](int pc, FS st) {
            auto it = at.find(pc);
            if (it == at.end()) { st.visits = 0; at[pc] = st; work.push_back(pc); return; }
            FS &o = it->second;
            if (o.st.size() != st.st.size()) fail("stack shapes differ at a join");
            bool ch = false;
            for (size_t k = 0; k < st.st.size(); ++k) {
                if (o.st[k].sel_n != st.st[k].sel_n || o.st[k].sel_stride != st.st[k].sel_stride) fail("sequence selections differ at a join");
                AV j = av_join(o.st[k].v, st.st[k].v);
                if (!(j == o.st[k].v)) { if (o.visits > 12) j = AV::top(); o.st[k].v = j; ch = true; }
            }
            for (int k = 0; k < 16; ++k) {
                AV j = av_join(o.t[k], st.t[k]);
                if (!(j == o.t[k])) { if (o.visits > 12) j = AV::top(); o.t[k] = j; ch = true; }
            }
            if (ch) { o.visits++; work.push_back(pc); }
        };
        long budget = 4000000;
        while (!work.empty()) {
            if (--budget < 0) fail("no fixed point");
            const int pc = work.back();
            work.pop_back();
            if (pc < 0 || pc >= (int)c.size()) fail("jump outside the image");
            FS a = at[pc];
            const int op = c[(size_t)pc], n = operands(op), next = pc + 1 + n;
            const int o0 = n > 0 ? c[(size_t)pc + 1] : 0, o1 = n > 1 ? c[(size_t)pc + 2] : 0, o2 = n > 2 ? c[(size_t)pc + 3] : 0;
            auto pop = [&]() -> Slot { if (a.st.empty()) fail("stack underflow"); Slot s = a.st.back(); a.st.pop_back(); return s; };
            auto push = [&](const AV &v) { Slot s; s.v = v; a.st.push_back(s); if (a.st.size() > 16) fail("stack deeper than 16"); };
            // the bases a sequence instruction works on: base, or base + k * stride for the selection on top of the stack
            auto bases = [&](bool indexed) -> std::vector<int> {
                if (!indexed) return {o0};
                const Slot s = pop();
                if (!s.sel_n) fail("sequence selection lost");
                std::vector<int> b;
                for (int k = 0; k < s.sel_n; ++k) b.push_back(o0 + k * s.sel_stride);
                return b;
            };
            auto arith = [&](int which) {
                const AV y = pop().v.plain(), x = pop().v.plain();
                if (x.bot || y.bot) { push(AV()); return; }
                int64_t lo, hi;
                if (which == VM_ADD) { lo = x.lo + y.lo; hi = x.hi + y.hi; }
                else if (which == VM_SUB) { lo = x.lo - y.hi; hi = x.hi - y.lo; }
                else { const int64_t p[4] = {x.lo * y.lo, x.lo * y.hi, x.hi * y.lo, x.hi * y.hi}; lo = *std::min_element(p, p + 4); hi = *std::max_element(p, p + 4); }
                push(av_clip(lo, hi));   // (a result outside 32 bits is R_ERROR, not a value)
            };
            switch (op) {
            case VM_HALT: case VM_FAIL: continue;
            case VM_PUSH: push(AV::cst(o0)); break;
            case VM_SELF: push(self); break;
            case VM_LOAD: push(load(o0)); break;
            case VM_LOADT: if (o0 < 0 || o0 >= 16) fail("temporary out of range"); push(a.t[o0].bot ? AV::top() : a.t[o0]); break;   // (never written on this path: anything)
            case VM_STORET: if (o0 < 0 || o0 >= 16) fail("temporary out of range"); a.t[o0] = pop().v; break;
            case VM_STORE: store(o0, pop().v); break;
            case VM_LOADX: { pop(); AV j; for (int k = 0; k < o2; ++k) j = av_join(j, load(o0 + k)); push(j); break; }
            case VM_STOREX: { const AV val = pop().v; pop(); for (int k = 0; k < o2; ++k) store(o0 + k, val); break; }
            case VM_ADD: case VM_SUB: case VM_MUL: arith(op); break;
            case VM_DIV: {
                pop(); const AV x = pop().v.plain();
                if (x.bot) { push(AV()); break; }
                const int64_t m = std::max(std::llabs(x.lo), std::llabs(x.hi));
                push(av_clip(-m - 1, m + 1));
                break;
            }
            case VM_MOD: {
                const AV y = pop().v.plain(); pop();
                if (y.bot) { push(AV()); break; }
                push(av_clip(0, std::max<int64_t>(y.hi - 1, 0)));
                break;
            }
            case VM_NEG: { const AV x = pop().v.plain(); if (x.bot) push(AV()); else push(av_clip(-x.hi, -x.lo)); break; }
            case VM_EQ: case VM_NE: case VM_LT: case VM_LE: case VM_GT: case VM_GE: pop(); pop(); push(AV::of(0, 1)); break;
            case VM_NOT: pop(); push(AV::of(0, 1)); break;
            case VM_JMP: flow(o0, a); continue;
            case VM_JZ: case VM_JNZ: pop(); flow(o0, a); break;
            case VM_CHOOSE: push(AV::of(0, std::max(o0 - 1, 0))); break;
            case VM_AWAIT: case VM_ASSERT: case VM_POP: pop(); break;
            case VM_SETPC: for (int cell : pcs) store(cell, AV::cst(o0)); break;
            case VM_NOP: case VM_OLD_ON: case VM_OLD_OFF: break;
            case VM_BIT: {
                const AV x = pop().v.plain();
                if (x.bot) { push(AV()); break; }
                const int64_t l = std::max<int64_t>(x.lo, 0), h = std::min<int64_t>(x.hi, 31);
                if (h >= 31 || l > h) push(AV::top()); else push(AV::of((int64_t)1 << l, (int64_t)1 << h));
                break;
            }
            case VM_OR: case VM_AND: case VM_ANDN: {
                const AV y = pop().v.plain(), x = pop().v.plain();
                if (x.bot || y.bot) { push(AV()); break; }
                auto mask_of = [](int64_t h) { int64_t m = 1; while (m <= h) m <<= 1; return m - 1; };
                if (op == VM_OR) { if (x.lo >= 0 && y.lo >= 0) push(AV::of(0, mask_of(std::max(x.hi, y.hi)))); else push(AV::top()); }
                else if (op == VM_AND) { if (x.lo >= 0 && y.lo >= 0) push(AV::of(0, std::min(x.hi, y.hi))); else if (x.lo >= 0) push(AV::of(0, x.hi)); else if (y.lo >= 0) push(AV::of(0, y.hi)); else push(AV::top()); }
                else { if (x.lo >= 0) push(AV::of(0, x.hi)); else push(AV::top()); }
                break;
            }
            case VM_POPCNT: pop(); push(AV::of(0, 32)); break;
            case VM_SEQSEL: { pop(); Slot s; s.v = AV::of(0, (int64_t)std::max(o1 - 1, 0) * o2); s.sel_n = o1; s.sel_stride = o2; a.st.push_back(s); break; }
            case VM_SEQLEN: { AV j; for (int B : bases(o2 != 0)) j = av_join(j, load(B)); push(j); break; }
            case VM_LOADSEQ: { const std::vector<int> bs = bases(o2 != 0); pop(); AV j; for (int B : bs) for (int k = 1; k <= o1; ++k) j = av_join(j, load(B + k)); push(j); break; }
            case VM_STORESEQ: { const std::vector<int> bs = bases(o2 != 0); const AV val = pop().v; pop(); for (int B : bs) for (int k = 1; k <= o1; ++k) store(B + k, val); break; }
            case VM_APPEND: {
                const std::vector<int> bs = bases(o2 != 0);
                const AV val = pop().v;
                for (int B : bs) {
                    for (int k = 1; k <= o1; ++k) store(B + k, val);
                    const AV len = load(B).plain();
                    if (!len.bot) { const int64_t l = std::max<int64_t>(len.lo, 0), h = std::min<int64_t>(len.hi, (int64_t)o1 - 1); if (l <= h) store(B, AV::of(l + 1, h + 1)); }
                }
                break;
            }
            case VM_TAIL:
                for (int B : bases(o2 != 0)) {
                    for (int k = 1; k < o1; ++k) { store(B + k, load(B + k + 1)); store(B + k, AV::cst(0)); }
                    store(B + o1, AV::cst(0));
                    const AV len = load(B).plain();
                    if (!len.bot) { const int64_t l = std::max<int64_t>(len.lo, 1), h = len.hi; if (l <= h) store(B, av_clip(l - 1, h - 1)); }
                }
                break;
            case VM_SEQCLR: for (int B : bases(o2 != 0)) for (int k = 0; k <= o1; ++k) store(B + k, AV::cst(0)); break;
            case VM_SEQCOPY: for (int k = 0; k <= o2; ++k) store(o0 + k, load(o1 + k)); break;
            case VM_RSADD: case VM_RSDEL: case VM_RSHAS: throw GenError{"sets of records are not translated (VM_RSADD / VM_RSDEL / VM_RSHAS)"};
            default: fail("instruction " + std::to_string(op) + " is not analysed");
            }
            flow(next, a);
        }
    }
};

static int bits_for(uint64_t span) { int b = 0; while (b < 64 && (span >> b)) ++b; return b; }

}  // namespace

std::string codegen(const Program &P, bool pack) {
    const std::vector<int> &c = P.image;
    if (c.size() < (size_t)mc::VMH_SIZE || c[mc::VMH_MAGIC] != mc::VM_MAGIC) throw GenError{"not a program image"};
    Gen g(c);
    const int nv = c[mc::VMH_NV], ninst = c[mc::VMH_NINST], maxch = c[mc::VMH_MAXCH], pc_base = c[mc::VMH_PC_BASE], done = c[mc::VMH_DONE],
              init_entry = c[mc::VMH_INIT_ENTRY], ninv = c[mc::VMH_NINV], ncon = c[mc::VMH_NCON], label_tab = c[mc::VMH_LABEL_TAB],
              self_tab = c[mc::VMH_SELF_TAB], nlabels = c[mc::VMH_NLABELS];
    const unsigned long long num_init = (unsigned long long)(uint32_t)c[mc::VMH_NUM_INIT_LO] | (unsigned long long)(uint32_t)c[mc::VMH_NUM_INIT_HI] << 32;
    std::ostringstream o;
    o << "// generated by tla_rust_amd/csrc/pcal_codegen.cpp from the compiled program of module " << P.module << ": do not edit\n"
      << "#pragma once\n#include \"spec_gen.h\"\nnamespace mc {\nstruct GenProg {\n"
      << "    static constexpr int NV = " << nv << ", NINST = " << ninst << ", MAXCH = " << maxch << ", PC_BASE = " << pc_base << ", DONE = " << done
      << ", NINV = " << ninv << ", NCON = " << ncon << ", NLABELS = " << nlabels << ";\n    static constexpr unsigned long long NUM_INIT = " << num_init << "ull;\n"
      << "    enum { R_DISABLED = 0, R_OK = 1, R_ASSERT = 2, R_ERROR = 3, R_OVERFLOW = 4 };\n";
    // entries: init, invariants, the labels
    // the variable cells as named members; helpers spec_gen.h asks for (words <-> cells, the pc cell of an instance)
    // which labels an instance can stand at (needed by the range analysis and by run_inst below)
    g.ninst_ = ninst;
    std::set<int> used;
    std::vector<std::set<int>> per_inst;
    for (int i = 0; i < ninst; ++i) {
        per_inst.push_back(g.labels_of(i, pc_base, init_entry, label_tab, nlabels));
        for (int l : per_inst.back()) if (c[(size_t)(label_tab + l)] >= 0) used.insert(l);
    }
    // ---- the row layout: the interpreter's (32 bits per cell, two per word) or cells packed to their inferred ranges
    std::vector<CellEnc> enc((size_t)nv);
    int nw = (nv + 1) / 2;
    const int vmw = (nv + 1) / 2;
    if (pack) {
        Ranges rg(c, nv, ninst, pc_base, self_tab);
        for (int k = 0; k < nv; ++k) rg.R[(size_t)k] = AV::cst(0);   // (zero() before the initial-state code)
        int rounds = 0;
        do {
            if (++rounds > 200) throw GenError{"range analysis: no fixed point"};
            rg.changed = false;
            rg.run(init_entry, AV::cst(0), {pc_base});
            for (int l : used) {
                AV self;
                std::vector<int> pcs;
                for (int i = 0; i < ninst; ++i)
                    if (per_inst[(size_t)i].count(l)) { self = av_join(self, AV::cst(c[(size_t)(self_tab + i)])); pcs.push_back(pc_base + i); }
                rg.run(c[(size_t)(label_tab + l)], self, pcs);
            }
            for (int k = 0; k < ninv + ncon; ++k) rg.run(c[(size_t)mc::VMH_INV0 + (size_t)k], AV::cst(0), {pc_base});
        } while (rg.changed);
        for (int k = 0; k < nv; ++k) {
            const AV &a = rg.R[(size_t)k];
            CellEnc &e = enc[(size_t)k];
            e.dflt = a.dflt;
            if (a.num) {
                const uint64_t n = (uint64_t)(a.hi - a.lo) + (a.dflt ? 1u : 0u);   // largest code
                if (n >= 0xffffffffull || a.hi - a.lo >= 0x7fffffffll) { e = CellEnc{}; continue; }   // the full 32 bits, raw
                e.lo = a.lo; e.span = n; e.bits = bits_for(n);
            } else {   // only ever the default value
                e.lo = 0; e.span = 0; e.bits = 0;
            }
        }
        // first fit, widest first; no cell straddles a word
        std::vector<int> order((size_t)nv);
        for (int k = 0; k < nv; ++k) order[(size_t)k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return enc[(size_t)x].bits > enc[(size_t)y].bits; });
        std::vector<int> fill;
        for (int k : order) {
            CellEnc &e = enc[(size_t)k];
            if (e.bits == 0) { e.word = 0; e.shift = 0; continue; }
            size_t wd = 0;
            while (wd < fill.size() && fill[wd] + e.bits > 64) ++wd;
            if (wd == fill.size()) fill.push_back(0);
            e.word = (int)wd; e.shift = fill[wd]; fill[wd] += e.bits;
        }
        nw = fill.empty() ? 1 : (int)fill.size();
        if (nw >= vmw) pack = false;   // nothing gained: keep the interpreter's rows
    }
    if (!pack) {
        nw = vmw;
        for (int k = 0; k < nv; ++k) { CellEnc e; e.word = k / 2; e.shift = (k & 1) * 32; enc[(size_t)k] = e; }
    }
    auto LOu = [](const CellEnc &e) { return std::to_string((uint32_t)(int32_t)e.lo) + "u"; };
    // code of cell k as an unsigned expression; decode of the code `x` (an unsigned expression) as an int32_t expression
    auto encode = [&](int k) -> std::string {
        const CellEnc &e = enc[(size_t)k];
        const std::string v = "v.c" + std::to_string(k);
        if (e.bits == 32 && e.span == 0xffffffffull) return "(uint32_t)" + v;
        if (e.dflt) return "(" + v + " == (" + std::to_string(mc::VM_DEFAULT_INIT) + ") ? 0u : (uint32_t)" + v + " - " + LOu(e) + " + 1u)";
        return "((uint32_t)" + v + " - " + LOu(e) + ")";
    };
    auto decode = [&](int k, const std::string &x) -> std::string {
        const CellEnc &e = enc[(size_t)k];
        if (e.bits == 32 && e.span == 0xffffffffull) return "(int32_t)(" + x + ")";
        if (e.dflt) return "((" + x + ") == 0u ? (int32_t)(" + std::to_string(mc::VM_DEFAULT_INIT) + ") : (int32_t)((" + x + ") - 1u + " + LOu(e) + "))";
        return "(int32_t)((" + x + ") + " + LOu(e) + ")";
    };
    auto field = [&](int k, const std::string &word) -> std::string {   // the code of cell k inside its word
        const CellEnc &e = enc[(size_t)k];
        if (e.bits == 0) return "0u";
        std::string x = word;
        if (e.shift) x = "(" + x + " >> " + std::to_string(e.shift) + ")";
        if (e.shift + e.bits < 64) x = "(" + x + " & " + std::to_string(e.bits >= 64 ? ~0ull : ((1ull << e.bits) - 1ull)) + "ull)";
        return "(uint32_t)" + x;
    };
    o << "    static constexpr int NW = " << nw << ", VMW = " << vmw << ";\n    static constexpr bool PACKED = " << (pack ? "true" : "false") << ";\n";
    o << "    // cell: bits @ word.shift [lo, lo + span]" << (pack ? "" : " (the interpreter's layout)") << "\n    //";
    for (int k = 0; k < nv; ++k) {
        const CellEnc &e = enc[(size_t)k];
        o << " c" << k << ":" << e.bits << "@" << e.word << "." << e.shift;
        if (pack && e.bits < 32) o << "[" << e.lo << (e.dflt ? ",D" : "") << "+" << e.span << "]";
    }
    o << "\n    struct Cells {";
    for (int k = 0; k < nv; ++k) o << " int32_t c" << k << ";";
    o << " };\n    MC_HD static void zero(Cells &v) {";
    for (int k = 0; k < nv; ++k) o << " v.c" << k << " = 0;";
    // to_words: false = a cell holds a value outside the range the analysis proved (never, if the analysis is sound: reported, not stored)
    o << " }\n    MC_HD static bool to_words(const Cells &v, uint64_t *w) {\n        uint32_t bad_ = 0;\n";
    for (int k = 0; k < nv; ++k) o << "        const uint32_t e" << k << " = " << encode(k) << ";\n";
    o << "#ifndef MC_GEN_NO_RANGE_CHECK   // (A/B knob: what the check costs)\n";
    for (int k = 0; k < nv; ++k) {
        const CellEnc &e = enc[(size_t)k];
        if (!(e.bits == 32 && e.span == 0xffffffffull)) o << "        bad_ |= (uint32_t)(e" << k << " > " << (uint32_t)e.span << "u);\n";
    }
    o << "#endif\n";
    for (int wd = 0; wd < nw; ++wd) {
        o << "        w[" << wd << "] = 0ull";
        for (int k = 0; k < nv; ++k) {
            const CellEnc &e = enc[(size_t)k];
            if (e.bits == 0 || e.word != wd) continue;
            o << " | (uint64_t)e" << k;
            if (e.shift) o << " << " << e.shift;
        }
        o << ";\n";
    }
    o << "        return bad_ == 0;\n    }\n    MC_HD static void from_words(const uint64_t *w, Cells &v) {\n";
    for (int k = 0; k < nv; ++k) o << "        v.c" << k << " = " << decode(k, field(k, "w[" + std::to_string(enc[(size_t)k].word) + "]")) << ";\n";
    // the interpreter's rows (two 32-bit cells per word): what leaves the engine (traces, read_states) and what the host helpers read
    o << "    }\n    MC_HD static void cells_to_vm(const Cells &v, uint64_t *w) {";
    for (int k = 0; k < vmw; ++k) {
        o << " w[" << k << "] = (uint64_t)(uint32_t)v.c" << 2 * k;
        if (2 * k + 1 < nv) o << " | (uint64_t)(uint32_t)v.c" << 2 * k + 1 << " << 32";
        o << ";";
    }
    o << " }\n    MC_HD static void cells_from_vm(const uint64_t *w, Cells &v) {";
    for (int k = 0; k < nv; ++k) o << " v.c" << k << " = (int32_t)(uint32_t)(w[" << k / 2 << "]" << (k & 1 ? " >> 32" : "") << ");";
    // the label instance `inst` stands at, straight from a packed row (the by-pairs kernel's sort key)
    o << " }\n    template <class Ref> MC_HD static int32_t pc_from_row(Ref row, int inst) {\n        switch (inst) {\n";
    for (int i = 0; i < ninst; ++i) {
        const int k = pc_base + i;
        o << "        case " << i << ": return " << decode(k, field(k, "row.get(" + std::to_string(enc[(size_t)k].word) + ")")) << ";\n";
    }
    o << "        default: return DONE;\n        }\n    }\n    template <int INST> MC_HD static int32_t &pc_cell(Cells &v) {";
    for (int i = 0; i < ninst; ++i) o << " if constexpr (INST == " << i << ") return v.c" << pc_base + i << ";";
    o << " }\n    template <int INST> MC_HD static int32_t pc_of(const Cells &v) {";
    for (int i = 0; i < ninst; ++i) o << " if constexpr (INST == " << i << ") return v.c" << pc_base + i << ";";
    o << " }\n";
    o << "    MC_HD static int run_init(uint64_t &ch, Cells &v) {\n        constexpr int32_t SELF_ = 0; constexpr int INST = 0; const Cells &old = v; int aux = 0; int32_t result = 0;\n        (void)SELF_; (void)INST;\n"
      << g.body(init_entry) << "    }\n";
    for (int k = 0; k < ninv + ncon; ++k) {
        o << "    MC_HD static int inv" << k << "(Cells &v, int32_t &result) {\n        constexpr int32_t SELF_ = 0; constexpr int INST = 0; const Cells &old = v; int aux = 0; uint64_t ch = 0;\n        (void)SELF_; (void)INST;\n"
          << g.body(c[(size_t)mc::VMH_INV0 + (size_t)k]) << "    }\n";
    }
    o << "    MC_HD static int run_inv(int k, Cells &v, int32_t &result) {\n        switch (k) {\n";
    for (int k = 0; k < ninv + ncon; ++k) o << "        case " << k << ": return inv" << k << "(v, result);\n";
    o << "        default: return R_ERROR;\n        }\n    }\n";
    for (int l : used) {
        o << "    template <int INST, int SELF_>\n    MC_HD static int label" << l << "(uint64_t &ch, Cells &v, const Cells &old, int32_t &result, int &aux) {\n"
          << g.body(c[(size_t)(label_tab + l)]) << "    }\n";
    }
    // choices a label's code can consume: the largest product of VM_CHOOSE operands along a path from its entry (a choice index beyond
    // it is never fully consumed, i.e. never enabled: the by-pairs kernel does not even queue it)
    o << "    MC_HD static int nch(int32_t label) {\n        switch (label) {\n";
    for (int l : used) o << "        case " << l << ": return " << g.max_choices(c[(size_t)(label_tab + l)], maxch) << ";\n";
    o << "        default: return " << maxch << ";\n        }\n    }\n";
    // the by-pairs kernel's sort key: one key per (instance, label the instance can stand at) + a last one ("Done", the terminating slot).  A label
    // function is a template over the INSTANCE (its cells are named members): a batch of 64 pairs that stand at ONE label but belong to N
    // different instances runs N copies of the label's code one after the other, each with 1 / N of the lanes.  Sorted by (instance, label)
    // a batch runs one copy.  (spec_gen.h uses these keys while they fit the kernel's 64; the label alone beyond that.)
    {
        int nk = 0;
        std::ostringstream sw;
        for (int i = 0; i < ninst; ++i) {
            sw << "        case " << i << ": switch (label) {";
            for (int l : per_inst[(size_t)i])
                if (c[(size_t)(label_tab + l)] >= 0) sw << " case " << l << ": return " << nk++ << ";";
            sw << " default: break; } break;\n";
        }
        o << "    static constexpr int NKEYS = " << nk + 1 << ";\n    MC_HD static int key_of(int inst, int32_t label) {\n        switch (inst) {\n" << sw.str()
          << "        default: break;\n        }\n        return NKEYS - 1;\n    }\n";
    }
    o << "    template <int INST>\n    MC_HD static int run_inst(int32_t label, uint64_t &ch, Cells &v, const Cells &old, int &aux) {\n        int32_t result = 0;\n";
    for (int i = 0; i < ninst; ++i) {
        o << "        if constexpr (INST == " << i << ") {\n            switch (label) {\n";
        for (int l : per_inst[(size_t)i])
            if (c[(size_t)(label_tab + l)] >= 0) o << "            case " << l << ": return label" << l << "<" << i << ", " << c[(size_t)(self_tab + i)] << ">(ch, v, old, result, aux);\n";
        o << "            default: return R_ERROR;   // (a label this instance cannot stand at)\n            }\n        }\n";
    }
    o << "        return R_DISABLED;\n    }\n};\nusing SpecGen = SpecGenT<GenProg>;\n}  // namespace mc\n";
    return o.str();
}

}  // namespace pcal

// ------------------------------------------------------------------------------------------------ C ABI + load-time build
// (the C ABI around these two is in frontend.cpp, where mc_program is defined: mc_program_codegen; engine.hip's mc_engine_create calls
//  mc_jit_factory with the program a MC_SPEC_PCAL descriptor carries)
// $TLAMC_JIT_PACK=0: keep the interpreter's rows (32 bits per cell) in generated code — the A/B knob of the packed layout
static bool pack_by_default() {
    const char *e = getenv("TLAMC_JIT_PACK");
    return !(e && *e == '0');
}
extern "C" long pcal_codegen_text(const pcal::Program *p, char *buf, size_t cap) {
    if (!p) return MC_EBADCFG;
    try {
        const std::string s = pcal::codegen(*p, pack_by_default());
        if (buf && cap) { const size_t n = s.size() < cap - 1 ? s.size() : cap - 1; memcpy(buf, s.data(), n); buf[n] = 0; }
        return (long)s.size();
    } catch (const pcal::GenError &e) {
        mc::set_error(std::string("codegen: ") + e.msg);
        return MC_EBADCFG;
    }
}

namespace {
uint64_t fnv(const std::string &s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char ch : s) { h ^= ch; h *= 0x100000001b3ull; }
    return h;
}
std::string dir_of_this_library() {
    Dl_info info;
    if (!dladdr((const void *)&pcal_codegen_text, &info) || !info.dli_fname) return "";
    std::string f = info.dli_fname;
    const size_t k = f.rfind('/');
    return k == std::string::npos ? "." : f.substr(0, k);
}
bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
}  // namespace

// Build (or find in the cache) the engine library of program `p` and return its factory: int (*)(const mc_spec_desc *, const mc_config *, mc::EngineBase **).
// The library is engine.hip compiled as translation unit 9 around the generated header (hipcc --offload-arch=gfx950: seconds to a minute,
// once per program text: the object is cached under $TLAMC_JIT_CACHE, default /tmp/tlamc_jit_<uid>, by the hash of the generated text and
// of the engine sources).  nullptr + mc_last_error when the program cannot be translated or the compiler fails: the caller interprets.
// pack: rows packed to the cells' inferred ranges (one-GPU engines; a sharded engine's rows travel between ranks and leave through
// mc_shard_* in the interpreter's layout, so it keeps that layout).
extern "C" void *mc_jit_factory_opts(const void *program /* pcal::Program * */, int pack) {
    std::string gen;
    try {
        gen = pcal::codegen(*reinterpret_cast<const pcal::Program *>(program), pack != 0 && pack_by_default());
    } catch (const pcal::GenError &e) {
        mc::set_error(std::string("codegen: ") + e.msg);
        return nullptr;
    }
    const std::string lib = dir_of_this_library();                 // .../tla_rust_amd/_build
    const std::string csrc = lib + "/../csrc", inc = lib + "/../../include";
    if (!exists(csrc + "/engine.hip")) { mc::set_error("jit: the engine sources are not beside the library (" + csrc + ")"); return nullptr; }
    uint64_t h = fnv(gen);
    for (const char *f : {"/engine.hip", "/engine_kernels.h", "/engine_pairs.h", "/spec_gen.h", "/spec_vm.h", "/mc_common.h"}) {
        struct stat st;
        if (stat((csrc + f).c_str(), &st) == 0) h = (h ^ (uint64_t)st.st_mtime ^ ((uint64_t)st.st_size << 20)) * 0x100000001b3ull;
    }
    // (A/B knob: $TLAMC_JIT_DEFS replaces the register / workgroup shape of the by-pairs kernel in the generated unit; part of the cache key)
    // (-fno-slp-vectorize below: clang 22's SLP vectorizer — of no use to per-lane scalar code — crashes on the shift-and-or chains of some
    //  programs' packed to_words: ms_queue, epoch_gc; first device run of the packed rows, profiles/r06zt)
    const char *jd = getenv("TLAMC_JIT_DEFS");
    const std::string shape = jd && *jd ? jd : "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1";
    h = (h ^ fnv(shape)) * 0x100000001b3ull;
    const char *cd = getenv("TLAMC_JIT_CACHE");
    const std::string cache = cd && *cd ? cd : "/tmp/tlamc_jit_" + std::to_string((unsigned)getuid());
    mkdir(cache.c_str(), 0700);
    char tag[32];
    snprintf(tag, sizeof tag, "%016llx", (unsigned long long)h);
    const std::string so = cache + "/libtlamc_gen_" + tag + ".so", hdr = cache + "/gen_" + tag + ".h";
    if (!exists(so)) {
        FILE *f = fopen(hdr.c_str(), "w");
        if (!f) { mc::set_error("jit: cannot write " + hdr); return nullptr; }
        fwrite(gen.data(), 1, gen.size(), f);
        fclose(f);
        const char *hc = getenv("HIPCC");
        const std::string tmp = so + "." + std::to_string((int)getpid()) + ".tmp";
        // (the shell itself moves the finished object into place: a build whose process has gone — `mc` starts one beside the interpreter and
        //  may be done before it — still completes the cache entry, and never leaves a half-written library under the final name)
        const std::string cmd = "( " + std::string(hc && *hc ? hc : "/opt/rocm/bin/hipcc") + " --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result -w -I " + inc +
                                " -I " + csrc + " -x hip -DMC_TU=9 -DMC_EXPAND_INSERT_MINW=2 " + shape + " -DMC_GEN_HEADER='\"" + hdr + "\"' " + csrc + "/engine.hip -o " + tmp + " -L" + lib + " -ltlamc -Wl,-rpath," + lib +
                                " && mv -f " + tmp + " " + so + " ) > " + cache + "/gen_" + tag + ".log 2>&1";
        if (system(cmd.c_str()) != 0 || !exists(so)) {
            if (pack != 0 && pack_by_default() && gen.find("PACKED = true") != std::string::npos) {
                // (a compiler that cannot digest the packed form — it has happened: see above — is no reason to interpret: the interpreter's rows)
                fprintf(stderr, "tlamc: jit: hipcc failed on the packed rows (see %s/gen_%s.log); building the generated code with the interpreter's rows\n", cache.c_str(), tag);
                return mc_jit_factory_opts(program, 0);
            }
            mc::set_error("jit: hipcc failed (see " + cache + "/gen_" + tag + ".log)");
            return nullptr;
        }
    }
    void *dl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) { mc::set_error(std::string("jit: dlopen: ") + dlerror()); return nullptr; }
    void *fn = dlsym(dl, "mc_make_engine_gen");
    if (!fn) mc::set_error("jit: the generated library has no mc_make_engine_gen");
    return fn;
}
extern "C" void *mc_jit_factory(const void *program) { return mc_jit_factory_opts(program, 1); }
