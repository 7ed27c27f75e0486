// pcal_compile.cpp — compiles a parsed PlusCal algorithm (pcal.h) to the program image spec_vm.h executes.
// Semantics follow the translation (examples/p-manual.pdf App. B): one atomic action per label; statements of a
// step see the assignments made earlier in the same step; `either` / `with` alternatives are separate
// successors; `await` disables; `assert` is an error raised while the successor is generated; the terminating
// disjunct (p.63) is the engine's last slot.  INVARIANTs are the zero-argument definitions the cfg names.
#include "pcal.h"
#include "spec_vm.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>

namespace pcal {
namespace {

struct CompileError { std::string msg; };
[[noreturn]] void cfail(const std::string &msg, Pos p = {}) {
    throw CompileError{p.line ? msg + " (line " + std::to_string(p.line) + ", column " + std::to_string(p.col) + ")" : msg};
}

bool has_label(const std::vector<SP> &v);
bool has_label(const SP &s) {
    for (const auto &b : s->blocks) if (has_label(b)) return true;
    return false;
}
bool has_label(const std::vector<SP> &v) {  // a label or a goto: either one makes the enclosing statement set pc itself
    for (const auto &s : v) if (!s->label.empty() || s->k == Stmt::GOTO || has_label(s)) return true;
    return false;
}
void all_labels(const std::vector<SP> &v, std::vector<std::string> &out) {
    for (const auto &s : v) {
        if (!s->label.empty()) out.push_back(s->label);
        for (const auto &b : s->blocks) all_labels(b, out);
    }
}

// the right-hand side of the first `name := e` (whole variable or name[self]) in a statement list, or null
const Expr *first_assignment(const std::vector<SP> &v, const std::string &name) {
    // (`x := defaultInitValue` — the expansion of a procedure's `return` — says nothing about what x holds otherwise)
    // (... and neither does `x := IF c THEN defaultInitValue ELSE x`: a frame slot of a recursive procedure being cleared by a `return`)
    auto is_default = [](const EP &e) { return e && e->k == Expr::ID && e->s == "defaultInitValue"; };
    auto tells = [&](const SP &a) { return !(is_default(a->e) || (a->e && a->e->k == Expr::IF && is_default(a->e->a[1]))); };
    for (const auto &s : v) {
        if (s->k == Stmt::ASSIGN) {
            if (s->var == name && tells(s)) return s->e.get();
            for (const auto &o : s->more) if (o->var == name && tells(o)) return o->e.get();
        }
        for (const auto &b : s->blocks) if (const Expr *e = first_assignment(b, name)) return e;
    }
    return nullptr;
}

struct Compiler {
    const Module &m;
    const Config &cfg;
    Program &P;
    std::vector<int> &c;  // the image being built
    std::map<std::string, int> var_index, str_id;
    std::map<std::string, ConstVal> consts;
    struct Bind {
        std::string name; int temp; bool is_const; long long value;
        const VarInfo *rs = nullptr;   // an ELEMENT of a set of records: `temp` holds its index (quantifiers) ...
        std::vector<int> ftemps;       // ... or these temporaries hold its fields (`with`: the body may change the set)
    };
    std::vector<Bind> binds;
    int next_temp = 0;
    const Proc *proc = nullptr;
    std::set<std::string> proc_locals;
    bool have_self_const = false;
    long long self_const = 0;
    std::vector<long long> procset;
    int inline_depth = 0;

    Compiler(const Module &mod, const Config &cf, Program &prog) : m(mod), cfg(cf), P(prog), c(prog.image) {}

    int intern(const std::string &s) {
        auto it = str_id.find(s);
        if (it != str_id.end()) return it->second;
        const int id = (int)P.strings.size();
        P.strings.push_back(s);
        str_id[s] = id;
        return id;
    }
    // conservative bound of the interpreter's stack depth (branches are tracked linearly: never an underestimate)
    int depth = 0, max_depth = 0;
    bool fallthrough = true;          // false right after JMP / HALT / FAIL: the next code is reached through a jump only
    std::map<int, int> jump_depth;    // operand slot of a jump -> stack depth at its target
    void track(int op) {
        fallthrough = !(op == mc::VM_JMP || op == mc::VM_HALT || op == mc::VM_FAIL);
        switch (op) {
        case mc::VM_PUSH: case mc::VM_SELF: case mc::VM_LOAD: case mc::VM_LOADT: case mc::VM_CHOOSE: depth++; break;
        case mc::VM_STORE: case mc::VM_STORET: case mc::VM_AWAIT: case mc::VM_ASSERT: case mc::VM_JZ: case mc::VM_JNZ: case mc::VM_POP:
        case mc::VM_ADD: case mc::VM_SUB: case mc::VM_MUL: case mc::VM_DIV: case mc::VM_MOD: case mc::VM_EQ: case mc::VM_NE: case mc::VM_LT:
        case mc::VM_LE: case mc::VM_GT: case mc::VM_GE: case mc::VM_OR: case mc::VM_AND: case mc::VM_ANDN: case mc::VM_APPEND: depth--; break;
        case mc::VM_SEQLEN: depth++; break;
        // (VM_RSADD / VM_RSDEL / VM_RSHAS pop their k fields: emit_rs adjusts the depth)
        case mc::VM_STOREX: case mc::VM_STORESEQ: depth -= 2; break;
        case mc::VM_HALT: depth = 0; break;
        default: break;
        }
        if (depth < 0) depth = 0;
        if (depth > max_depth) max_depth = depth;
    }
    void emit(int op) { c.push_back(op); track(op); }
    void emit(int op, int a) { c.push_back(op); c.push_back(a); track(op); }
    int emit_jump(int op) {
        c.push_back(op);
        c.push_back(-1);
        track(op);
        jump_depth[(int)c.size() - 1] = depth;
        return (int)c.size() - 1;
    }
    void patch(int at) {  // the code that follows is a jump target: its depth is the jump's (and the fall-through's, if any)
        c[(size_t)at] = (int)c.size();
        const int d = jump_depth.count(at) ? jump_depth[at] : depth;
        depth = fallthrough ? std::max(depth, d) : d;
        fallthrough = true;
        if (depth > max_depth) max_depth = depth;
    }
    bool copy_read = false;   // ex_rhs: the variable being compiled is only copied
    int new_temp(Pos p) {
        if (next_temp >= mc::SpecVm::TEMPS) cfail("expression too deeply nested (temporaries exhausted)", p);
        return next_temp++;
    }

    // ---- constants
    bool const_scalar(const EP &e, long long &out) {
        switch (e->k) {
        case Expr::NUM: out = e->num; return true;
        case Expr::BOOL: out = e->num; return true;
        case Expr::STR: out = intern(e->s); return true;
        case Expr::ID: {
            for (size_t i = binds.size(); i-- > 0;) if (binds[i].name == e->s) { if (binds[i].is_const) { out = binds[i].value; return true; } return false; }
            if (e->s == "self" && have_self_const) { out = self_const; return true; }
            auto it = consts.find(e->s);
            if (it == consts.end()) return false;
            if (it->second.k == ConstVal::INT) { out = it->second.i; return true; }
            if (it->second.k == ConstVal::STR) { out = intern(it->second.s); return true; }
            return false;
        }
        case Expr::UNOP: { long long a; if (e->s == "-" && const_scalar(e->a[0], a)) { out = -a; return true; } return false; }
        case Expr::BINOP: {
            long long a, b;
            if (!const_scalar(e->a[0], a) || !const_scalar(e->a[1], b)) return false;
            if (e->s == "+") out = a + b; else if (e->s == "-") out = a - b; else if (e->s == "*") out = a * b;
            else if (e->s == "\\div" && b > 0) out = a / b - ((a % b != 0 && a < 0) ? 1 : 0);
            else if (e->s == "%" && b > 0) out = ((a % b) + b) % b;
            else return false;
            return true;
        }
        default: return false;
        }
    }
    bool const_from(const ConstVal &v, std::vector<long long> &out) {
        if (v.k != ConstVal::SET) return false;
        for (const auto &x : v.elems) {
            if (x.k == ConstVal::INT) out.push_back(x.i);
            else if (x.k == ConstVal::STR) out.push_back(intern(x.s));
            else return false;
        }
        return true;
    }
    bool const_set(const EP &e0, std::vector<long long> &out) {
        const EP e = e0->k == Expr::UNOP && e0->s == "DOMAIN" && !seq_ref(e0->a[0]) ? resolve_domain(e0) : e0;
        if (e->k == Expr::SETENUM) {
            for (const auto &x : e->a) { long long v; if (!const_scalar(x, v)) return false; out.push_back(v); }
            return true;
        }
        if (e->k == Expr::BINOP && e->s == "..") {
            long long a, b;
            if (!const_scalar(e->a[0], a) || !const_scalar(e->a[1], b)) return false;
            if (b - a > 100000) return false;
            for (long long x = a; x <= b; x++) out.push_back(x);
            return true;
        }
        if (e->k == Expr::BINOP && (e->s == "\\cup" || e->s == "\\union")) {
            std::vector<long long> a, b;
            if (!const_set(e->a[0], a) || !const_set(e->a[1], b)) return false;
            out = a;
            for (long long x : b) if (std::find(out.begin(), out.end(), x) == out.end()) out.push_back(x);
            return true;
        }
        if (e->k == Expr::ID) {
            if (e->s == "BOOLEAN") { out = {0, 1}; return true; }
            if (e->s == "ProcSet" && !procset.empty()) { out = procset; return true; }
            auto it = consts.find(e->s);
            if (it != consts.end()) return const_from(it->second, out);
            for (const auto &d : m.defs) if (d.name == e->s && d.params.empty() && inline_depth < 8) { inline_depth++; const bool ok = const_set(d.body, out); inline_depth--; return ok; }
        }
        return false;
    }
    char type_of(const EP &e) {
        switch (e->k) {
        case Expr::STR: return 's';
        case Expr::BOOL: return 'b';
        case Expr::QUANT: return e->s == "CHOOSE" ? type_of(e->a[0]) : 'b';
        case Expr::UNOP: return e->s == "~" ? 'b' : 'i';
        case Expr::BINOP: {
            static const char *boolops[] = {"/\\", "\\/", "=>", "<=>", "=", "#", "<", ">", "<=", ">=", "\\in", "\\notin", "\\subseteq"};
            for (const char *o : boolops) if (e->s == o) return 'b';
            if (e->s == "..") return type_of(e->a[0]);
            return 'i';
        }
        case Expr::IF: return type_of(e->a[1]);
        case Expr::SETENUM: case Expr::TUPLE: return e->a.empty() ? 'i' : type_of(e->a[0]);
        case Expr::SETOF: return e->s == "map" ? type_of(e->a[1]) : type_of(e->a[0]);
        case Expr::FUNCDEF: return type_of(e->a[1]);
        case Expr::ID: {
            if (e->s == "BOOLEAN") return 'b';
            auto it = consts.find(e->s);
            if (it != consts.end()) {
                if (it->second.k == ConstVal::STR) return 's';
                if (it->second.k == ConstVal::SET && !it->second.elems.empty() && it->second.elems[0].k == ConstVal::STR) return 's';
                return 'i';
            }
            auto vi = var_index.find(e->s);
            if (vi != var_index.end()) return P.vars[(size_t)vi->second].type;
            return 'i';
        }
        case Expr::INDEX: return type_of(e->a[0]);
        case Expr::DOT:
            if (const Bind *b = rs_bind(e->a[0]))
                for (size_t k = 0; k < b->rs->fields.size(); k++) if (b->rs->fields[k] == e->s) return b->rs->ftypes[k];
            return 'i';
        case Expr::CALL:
            if (e->s == "Head" && e->a.size() == 1) return type_of(e->a[0]);
            for (const auto &d : m.defs) if (d.name == e->s && d.params.size() == e->a.size()) return type_of(d.body);
            return 'i';
        default: return 'i';
        }
    }

    // ---- array access helpers
    static bool contiguous(const std::vector<long long> &ids) {
        for (size_t i = 1; i < ids.size(); i++) if (ids[i] != ids[0] + (long long)i) return false;
        return !ids.empty();
    }
    void emit_indexed(int op, const VarInfo &vi, Pos p) {
        if (!contiguous(vi.ids)) cfail("`" + vi.name + "` is indexed but its domain is not an integer interval", p);
        c.push_back(op);
        track(op);
        c.push_back(vi.base);
        c.push_back((int)vi.ids[0]);
        c.push_back((int)vi.ids.size());
    }
    void push_self(Pos p) {
        if (have_self_const) emit(mc::VM_PUSH, (int)self_const);
        else if (proc) emit(mc::VM_SELF);
        else cfail("`self` outside a process", p);
    }

    // ---- bounded sequences
    // element cells per sequence; a longer sequence is MC_EOVERFLOW, never truncated.  $TLAMC_PCAL_SEQ (1 .. 16) changes it: an ARRAY of
    // sequences (the channels of a message-passing algorithm) costs |domain| x (cells + 1) of the 128 cells a state has
    static int seq_cap() {
        const char *e = getenv("TLAMC_PCAL_SEQ");
        const int n = e ? atoi(e) : 8;
        return n < 1 ? 1 : n > 16 ? 16 : n;
    }
    const VarInfo *seq_var(const EP &e) {
        if (e->k != Expr::ID) return nullptr;
        for (size_t i = binds.size(); i-- > 0;) if (binds[i].name == e->s) return nullptr;
        auto vi = var_index.find(e->s);
        return vi != var_index.end() && P.vars[(size_t)vi->second].seq ? &P.vars[(size_t)vi->second] : nullptr;
    }
    // a sequence: a sequence variable q, or the element box[i] of an ARRAY of sequences (`box = [p \in S |-> <<>>]`)
    struct SeqRef {
        const VarInfo *v = nullptr;
        EP idx;    // array: the index as it is evaluated (possibly a temporary that holds it)
        EP same;   // array: the index as it was written (what `box[i] := Append(box[i], e)` compares)
        explicit operator bool() const { return v != nullptr; }
    };
    std::map<std::string, EP> bound_orig;   // a temporary that holds an index -> the expression it was evaluated from
    SeqRef seq_ref(const EP &e) {
        if (e->k == Expr::ID) {
            const VarInfo *v = seq_var(e);
            return v && !v->array ? SeqRef{v, nullptr, nullptr} : SeqRef{};
        }
        if (e->k == Expr::INDEX && e->a[0]->k == Expr::ID) {
            const VarInfo *v = seq_var(e->a[0]);
            if (v && v->array) return SeqRef{v, e->a[1], e->a[1]};
        }
        return SeqRef{};
    }
    static bool same_expr(const EP &a, const EP &b) {
        if (!a || !b) return !a && !b;
        if (a->k != b->k || a->num != b->num || a->s != b->s || a->bound != b->bound || a->names != b->names || a->a.size() != b->a.size()) return false;
        for (size_t k = 0; k < a->a.size(); k++) if (!same_expr(a->a[k], b->a[k])) return false;
        return true;
    }
    void emit_sel(const SeqRef &r) {  // selects the element of the array the next sequence instruction works on
        if (!r.idx) return;
        if (!contiguous(r.v->ids)) cfail("`" + r.v->name + "` is indexed but its domain is not an integer interval", r.idx->pos);
        ex(r.idx);
        c.push_back(mc::VM_SEQSEL);
        track(mc::VM_SEQSEL);
        c.push_back((int)r.v->ids[0]);
        c.push_back((int)r.v->ids.size());
        c.push_back(r.v->cap + 1);
    }
    void emit_seq(int op, const SeqRef &r) {
        emit_sel(r);
        c.push_back(op);
        track(op);
        c.push_back(r.v->base);
        c.push_back(r.v->cap);
        c.push_back(r.idx ? 1 : 0);   // 1: the offset VM_SEQSEL left on the stack is added to base
        if (r.idx && depth > 0) depth--;
    }
    void emit_len(const SeqRef &r) {
        if (!r.idx) { emit(mc::VM_LOAD, r.v->base); return; }
        emit_seq(mc::VM_SEQLEN, r);
    }
    // an array index that is neither a constant nor `self` is evaluated ONCE, into a temporary the reference then reads
    // (returns the number of temporaries taken: the caller gives them back)
    int pin_index(SeqRef &r) {
        long long cv;
        if (!r.idx || const_scalar(r.idx, cv) || (r.idx->k == Expr::ID && r.idx->s == "self")) return 0;
        if (r.idx->k == Expr::ID && bound_orig.count(r.idx->s)) return 0;
        const int t = new_temp(r.idx->pos);
        ex(r.idx);
        emit(mc::VM_STORET, t);
        const std::string name = "\001s" + std::to_string(t);
        binds.push_back({name, t, false, 0});
        bound_orig[name] = r.same;
        auto id = std::make_shared<Expr>();
        id->k = Expr::ID;
        id->s = name;
        id->pos = r.idx->pos;
        r.idx = id;
        return 1;
    }
    void unpin_index(int n) {
        if (!n) return;
        bound_orig.erase(binds.back().name);
        binds.pop_back();
        next_temp--;
    }
    // q = <<a, b>>  (also #): Len(q) = n /\ q[1] = a /\ ...
    void seq_equals(SeqRef q, const EP &tuple, bool negate) {
        const int pinned = pin_index(q);
        emit_len(q); emit(mc::VM_PUSH, (int)tuple->a.size()); emit(mc::VM_EQ);
        std::vector<int> fails;
        fails.push_back(emit_jump(mc::VM_JZ));
        for (size_t k = 0; k < tuple->a.size(); k++) {
            if (q.idx) { emit(mc::VM_PUSH, 1 + (int)k); emit_seq(mc::VM_LOADSEQ, q); } else emit(mc::VM_LOAD, q.v->base + 1 + (int)k);
            ex(tuple->a[k]); emit(mc::VM_EQ);
            fails.push_back(emit_jump(mc::VM_JZ));
        }
        emit(mc::VM_PUSH, negate ? 0 : 1);
        const int jend = emit_jump(mc::VM_JMP);
        for (int f : fails) patch(f);
        emit(mc::VM_PUSH, negate ? 1 : 0);
        patch(jend);
        unpin_index(pinned);
    }
    // dst := <sequence expression>: <<..>>, q, Append(q, e), Tail(q), q \o <<..>>
    void assign_seq(SeqRef dst, const EP &e) {
        const int pinned = pin_index(dst);
        auto copy_from = [&](const EP &src) {
            const SeqRef sv = seq_ref(src);
            if (!sv) cfail("expected a sequence variable", src->pos);
            if (sv.v == dst.v) {
                const EP want = dst.same && dst.same->k == Expr::ID && bound_orig.count(dst.same->s) ? bound_orig[dst.same->s] : dst.same;
                if (dst.v->array && !same_expr(sv.same, want))
                    cfail("`" + dst.v->name + "[i] := ...` can only start from the same element " + dst.v->name + "[i] (copying one sequence of the array to another is not supported)", src->pos);
                return;
            }
            if (sv.v->array || dst.v->array) cfail("copying a sequence into / out of the array `" + (dst.v->array ? dst.v->name : sv.v->name) + "` is not supported", src->pos);
            c.push_back(mc::VM_SEQCOPY); c.push_back(dst.v->base); c.push_back(sv.v->base); c.push_back(dst.v->cap);
        };
        auto elems_to_temps = [&](const std::vector<EP> &xs, std::vector<int> &tmp) {
            for (const auto &x : xs) ex(x);
            for (size_t k = 0; k < xs.size(); k++) tmp.push_back(new_temp(e->pos));
            for (size_t k = xs.size(); k-- > 0;) emit(mc::VM_STORET, tmp[k]);
        };
        if (e->k == Expr::TUPLE) {
            // the elements may read dst (q := <<Head(q)>>): evaluate them first
            if ((int)e->a.size() > dst.v->cap) cfail("sequence literal longer than the " + std::to_string(dst.v->cap) + " cells a sequence variable has", e->pos);
            std::vector<int> tmp;
            elems_to_temps(e->a, tmp);
            emit_seq(mc::VM_SEQCLR, dst);
            for (int t : tmp) { emit(mc::VM_LOADT, t); emit_seq(mc::VM_APPEND, dst); }
            next_temp -= (int)tmp.size();
        } else if (e->k == Expr::ID || (e->k == Expr::INDEX && seq_ref(e))) {
            copy_from(e);
        } else if (e->k == Expr::CALL && e->s == "Append" && e->a.size() == 2) {
            ex(e->a[1]);  // evaluated on the old value of every variable
            copy_from(e->a[0]);
            emit_seq(mc::VM_APPEND, dst);
        } else if (e->k == Expr::CALL && e->s == "Tail" && e->a.size() == 1) {
            copy_from(e->a[0]);
            emit_seq(mc::VM_TAIL, dst);
        } else if (e->k == Expr::BINOP && (e->s == "\\o" || e->s == "\\circ") && e->a[1]->k == Expr::TUPLE) {
            std::vector<int> tmp;
            elems_to_temps(e->a[1]->a, tmp);
            copy_from(e->a[0]);
            for (int t : tmp) { emit(mc::VM_LOADT, t); emit_seq(mc::VM_APPEND, dst); }
            next_temp -= (int)tmp.size();
        } else {
            cfail("a sequence variable can be assigned <<...>>, another sequence, Append(q, e), Tail(q) or q \\o <<...>>", e->pos);
        }
        unpin_index(pinned);
    }
    // the element type of a sequence that starts empty: that of the first value an assignment puts into it
    char seq_elem_type(const std::string &name) {
        char found = 0;
        std::function<void(const std::vector<SP> &)> walk = [&](const std::vector<SP> &v) {
            for (const auto &s : v) {
                if (found) return;
                if (s->k == Stmt::ASSIGN) {
                    std::vector<const Stmt *> all{s.get()};
                    for (const auto &o : s->more) all.push_back(o.get());
                    for (const Stmt *a : all) {
                        if (a->var != name || !a->e || found) continue;
                        const EP &e = a->e;
                        if (e->k == Expr::TUPLE && !e->a.empty()) found = type_of(e->a[0]);
                        else if (e->k == Expr::CALL && e->s == "Append" && e->a.size() == 2) found = type_of(e->a[1]);
                        else if (e->k == Expr::BINOP && (e->s == "\\o" || e->s == "\\circ") && e->a[1]->k == Expr::TUPLE && !e->a[1]->a.empty()) found = type_of(e->a[1]->a[0]);
                    }
                }
                for (const auto &b : s->blocks) walk(b);
            }
        };
        for (const auto &p : m.procs) walk(p.body);
        return found ? found : 'i';
    }
    // the types of the fields of a set of records: per field, the first value put into it (initial elements, then `\\cup {r}` / `{r}` in the
    // algorithm) that is not the defaultInitValue of a record parameter
    std::string rset_field_types(const VarInfo &v) {
        std::string types(v.fields.size(), 0);
        auto see = [&](const EP &lit) {
            if (!lit || lit->k != Expr::SETENUM) return;
            for (const auto &r : lit->a) {
                if (r->k != Expr::RECORD) continue;
                for (size_t f = 0; f < v.fields.size(); f++) {
                    if (types[f]) continue;
                    for (size_t k = 0; k < r->names.size(); k++)
                        if (r->names[k] == v.fields[f] && !(r->a[k]->k == Expr::ID && r->a[k]->s == "defaultInitValue")) types[f] = type_of(r->a[k]);
                }
            }
        };
        auto decl = [&](const std::vector<VarDecl> &ds) { for (const auto &d : ds) if (d.name == v.name) see(d.init); };
        decl(m.globals);
        for (const auto &p : m.procs) decl(p.locals);
        std::function<void(const EP &)> chain = [&](const EP &e) {
            if (!e) return;
            if (e->k == Expr::SETENUM) { see(e); return; }
            if (e->k == Expr::BINOP) { chain(e->a[0]); chain(e->a[1]); }
        };
        std::function<void(const std::vector<SP> &)> walk = [&](const std::vector<SP> &b) {
            for (const auto &s : b) {
                if (s->k == Stmt::ASSIGN && s->var == v.name) chain(s->e);
                for (const auto &x : s->blocks) walk(x);
            }
        };
        for (const auto &p : m.procs) walk(p.body);
        for (auto &t : types) if (!t) t = 'i';
        return types;
    }
    // the scalar operands of a sequence expression (the elements of <<..>>, Append's value, the elements behind \o): `||` evaluates
    // them before anything is stored, and hands assign_seq the expression with temporaries in their places
    EP seq_operands_to_temps(const EP &e, int &taken) {
        auto hold = [&](const EP &x) {
            const int t = new_temp(x->pos);
            ex(x);
            emit(mc::VM_STORET, t);
            const std::string name = "\001o" + std::to_string(t);
            binds.push_back({name, t, false, 0});
            taken++;
            auto id = std::make_shared<Expr>();
            id->k = Expr::ID;
            id->s = name;
            id->pos = x->pos;
            return EP(id);
        };
        auto c2 = std::make_shared<Expr>(*e);
        if (e->k == Expr::TUPLE) { for (auto &x : c2->a) x = hold(x); return c2; }
        if (e->k == Expr::CALL && e->s == "Append" && e->a.size() == 2) { c2->a[1] = hold(e->a[1]); return c2; }
        if (e->k == Expr::BINOP && (e->s == "\\o" || e->s == "\\circ") && e->a[1]->k == Expr::TUPLE) {
            auto t2 = std::make_shared<Expr>(*e->a[1]);
            for (auto &x : t2->a) x = hold(x);
            c2->a[1] = t2;
            return c2;
        }
        return c2;  // Tail(q), q: nothing to evaluate
    }

    // ---- sets of records (pcal.h VarInfo::rset; spec_vm.h VM_RSADD)
    const RecordVar *rset_record(const std::string &name) const {
        for (const auto &r : m.records) if (r.set && r.name == name) return &r;
        return nullptr;
    }
    const VarInfo *rset_var(const EP &e) {
        if (e->k != Expr::ID) return nullptr;
        for (size_t i = binds.size(); i-- > 0;) if (binds[i].name == e->s) return nullptr;
        auto vi = var_index.find(e->s);
        return vi != var_index.end() && P.vars[(size_t)vi->second].rset ? &P.vars[(size_t)vi->second] : nullptr;
    }
    void push_record(const VarInfo &v, const EP &r) {  // the fields of a constructor, in the order of the set's cells
        if (r->k != Expr::RECORD) cfail("expected a record constructor as an element of the set of records `" + v.name + "`", r->pos);
        for (const auto &f : v.fields) {
            size_t k = 0;
            while (k < r->names.size() && r->names[k] != f) k++;
            if (k == r->names.size()) cfail("the record has no field " + f, r->pos);
            ex(r->a[k]);
        }
    }
    void emit_rs(int op, const VarInfo &v) {
        c.push_back(op); track(op);
        c.push_back(v.base); c.push_back(v.cap); c.push_back((int)v.fields.size());
        depth -= (int)v.fields.size() - (op == mc::VM_RSHAS ? 1 : 0);
        if (depth < 0) depth = 0;
    }
    const Bind *rs_bind(const EP &e) {  // the binding of a name that stands for an element of a set of records
        if (e->k != Expr::ID) return nullptr;
        for (size_t i = binds.size(); i-- > 0;) if (binds[i].name == e->s) return binds[i].rs ? &binds[i] : nullptr;
        return nullptr;
    }
    void load_field(const Bind &b, const std::string &f, Pos p) {
        const VarInfo &v = *b.rs;
        size_t k = 0;
        while (k < v.fields.size() && v.fields[k] != f) k++;
        if (k == v.fields.size()) cfail("the record has no field " + f, p);
        if (!b.ftemps.empty()) {
            if (b.ftemps[k] < 0) cfail("internal: field " + f + " of a `with` element was not copied", p);
            emit(mc::VM_LOADT, b.ftemps[k]);
            return;
        }
        emit(mc::VM_LOADT, b.temp);
        c.push_back(mc::VM_LOADX); track(mc::VM_LOADX);
        c.push_back(v.base + 1 + (int)k * v.cap); c.push_back(0); c.push_back(v.cap);
    }
    // msgs := {r, ...} | msgs \cup {r, ...} | msgs \ {r, ...} | msgs
    void assign_rset(const VarInfo &v, const EP &e) {
        auto each = [&](const EP &lit, int op) { for (const auto &r : lit->a) { push_record(v, r); emit_rs(op, v); } };
        if (e->k == Expr::SETENUM) {
            // the elements may read the set (`msgs := {[n |-> Cardinality(msgs)]}`): every field of every element first, then the clear
            for (const auto &r : e->a) push_record(v, r);
            c.push_back(mc::VM_SEQCLR); track(mc::VM_SEQCLR); c.push_back(v.base); c.push_back((int)v.fields.size() * v.cap); c.push_back(0);
            for (size_t k = 0; k < e->a.size(); k++) emit_rs(mc::VM_RSADD, v);
            return;
        }
        // msgs | chain \cup {r, ...} | chain \ {r, ...}: every field of every record is evaluated first (on the set as it was), pushed so
        // that the first operation's record is on top; then the operations run in the order written
        std::vector<std::pair<int, EP>> ops;
        std::function<bool(const EP &)> chain = [&](const EP &x) -> bool {
            if (x->k == Expr::ID) return rset_var(x) == &v;
            if (x->k != Expr::BINOP || x->a[1]->k != Expr::SETENUM || !chain(x->a[0])) return false;
            const int op = (x->s == "\\cup" || x->s == "\\union") ? (int)mc::VM_RSADD : x->s == "\\" ? (int)mc::VM_RSDEL : -1;
            if (op < 0) return false;
            for (const auto &r : x->a[1]->a) ops.push_back({op, r});
            return true;
        };
        if (chain(e)) {
            for (size_t k = ops.size(); k-- > 0;) push_record(v, ops[k].second);
            for (const auto &o : ops) emit_rs(o.first, v);
            return;
        }
        cfail("a set of records can be assigned {r, ...}, " + v.name + " \\cup {r, ...} or " + v.name + " \\ {r, ...}", e->pos);
    }

    // ---- sets of small naturals as masks
    const VarInfo *set_var(const EP &e) {
        if (e->k != Expr::ID) return nullptr;
        for (size_t i = binds.size(); i-- > 0;) if (binds[i].name == e->s) return nullptr;
        auto vi = var_index.find(e->s);
        return vi != var_index.end() && P.vars[(size_t)vi->second].set ? &P.vars[(size_t)vi->second] : nullptr;
    }
    // DOMAIN f of a function variable is its (constant) index set, DOMAIN q of a sequence the interval 1..Len(q)
    EP resolve_domain(const EP &e) {
        if (!e || e->k != Expr::UNOP || e->s != "DOMAIN") return e;
        const EP &x = e->a[0];
        if (seq_ref(x)) {
            auto len = std::make_shared<Expr>();
            len->k = Expr::CALL; len->s = "Len"; len->a = {x}; len->pos = e->pos;
            auto one = std::make_shared<Expr>();
            one->k = Expr::NUM; one->num = 1; one->pos = e->pos;
            auto iv = std::make_shared<Expr>();
            iv->k = Expr::BINOP; iv->s = ".."; iv->a = {one, len}; iv->pos = e->pos;
            return iv;
        }
        if (x->k == Expr::ID) {
            bool shadowed = false;
            for (const auto &b : binds) shadowed |= b.name == x->s;
            auto vi = var_index.find(x->s);
            if (!shadowed && vi != var_index.end() && P.vars[(size_t)vi->second].array && !P.vars[(size_t)vi->second].seq) {
                auto lit = std::make_shared<Expr>();
                lit->k = Expr::SETENUM; lit->pos = e->pos;
                for (long long id : P.vars[(size_t)vi->second].ids) {
                    auto n = std::make_shared<Expr>();
                    n->k = Expr::NUM; n->num = id; n->pos = e->pos;
                    lit->a.push_back(n);
                }
                return lit;
            }
        }
        cfail("DOMAIN is supported on a function variable and on a sequence", e->pos);
    }
    // does the expression denote a set whose value depends on the state (a set variable somewhere inside)?
    bool dynamic_set(const EP &e0) {
        const EP e = resolve_domain(e0);
        if (e->k == Expr::SETOF) return true;
        if (set_var(e)) return true;
        if (e->k == Expr::BINOP && (e->s == "\\cup" || e->s == "\\union" || e->s == "\\cap" || e->s == "\\intersect" || e->s == "\\"))
            return dynamic_set(e->a[0]) || dynamic_set(e->a[1]);
        if (e->k == Expr::SETENUM) { for (const auto &x : e->a) { long long v; if (!const_scalar(x, v)) return true; } }
        return false;
    }
    // leave the 32-bit mask of a set expression on the stack
    void ex_set(const EP &e0) {
        const EP e = resolve_domain(e0);
        if (e->k == Expr::SETOF) {   // {x \\in S : P}: the members of S for which P holds; {f : x \\in S}: the values f takes (all within 0..31)
            const bool filter = e->s == "filter";
            const EP dom = resolve_domain(e->a[0]);
            std::vector<long long> elems;
            if (!dynamic_set(dom) && !(dom->k == Expr::BINOP && dom->s == "..") && const_set(dom, elems)) {
                emit(mc::VM_PUSH, 0);
                for (long long x : elems) {
                    if (filter && (x < 0 || x > 31)) cfail("only 0..31 can be members of a set value", e->pos);
                    binds.push_back({e->bound, 0, true, x});
                    if (filter) {
                        ex(e->a[1]);
                        const int skip = emit_jump(mc::VM_JZ);
                        emit(mc::VM_PUSH, (int)(1u << x)); emit(mc::VM_OR);
                        patch(skip);
                    } else {
                        ex(e->a[1]); emit(mc::VM_BIT); emit(mc::VM_OR);
                    }
                    binds.pop_back();
                }
                return;
            }
            // a state-dependent domain: an interval (walked from its lower to its upper bound) or a set value (its members among 0..31)
            const bool interval = dom->k == Expr::BINOP && dom->s == "..";
            const int tx = new_temp(e->pos), th = new_temp(e->pos), tacc = new_temp(e->pos);
            if (interval) { ex(dom->a[0]); emit(mc::VM_STORET, tx); ex(dom->a[1]); emit(mc::VM_STORET, th); }
            else { emit(mc::VM_PUSH, 0); emit(mc::VM_STORET, tx); ex_set(dom); emit(mc::VM_STORET, th); }
            emit(mc::VM_PUSH, 0); emit(mc::VM_STORET, tacc);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx);
            if (interval) emit(mc::VM_LOADT, th); else emit(mc::VM_PUSH, 31);
            emit(mc::VM_LE);
            const int jdone = emit_jump(mc::VM_JZ);
            int jskip = -1;
            if (!interval) { emit(mc::VM_LOADT, tx); emit(mc::VM_BIT); emit(mc::VM_LOADT, th); emit(mc::VM_AND); jskip = emit_jump(mc::VM_JZ); }
            binds.push_back({e->bound, tx, false, 0});
            if (filter) {
                ex(e->a[1]);
                const int no = emit_jump(mc::VM_JZ);
                emit(mc::VM_LOADT, tacc); emit(mc::VM_LOADT, tx); emit(mc::VM_BIT); emit(mc::VM_OR); emit(mc::VM_STORET, tacc);
                patch(no);
            } else {
                emit(mc::VM_LOADT, tacc); ex(e->a[1]); emit(mc::VM_BIT); emit(mc::VM_OR); emit(mc::VM_STORET, tacc);
            }
            binds.pop_back();
            if (jskip >= 0) patch(jskip);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jdone);
            emit(mc::VM_LOADT, tacc);
            next_temp -= 3;
            return;
        }
        if (const VarInfo *v = set_var(e)) {
            if (proc && proc_locals.count(e->s) && proc->is_set && P.multi) { push_self(e->pos); emit_indexed(mc::VM_LOADX, *v, e->pos); }
            else emit(mc::VM_LOAD, v->base);
            return;
        }
        if (e->k == Expr::BINOP && (e->s == "\\cup" || e->s == "\\union" || e->s == "\\cap" || e->s == "\\intersect" || e->s == "\\")) {
            ex_set(e->a[0]);
            ex_set(e->a[1]);
            emit(e->s == "\\" ? mc::VM_ANDN : (e->s == "\\cap" || e->s == "\\intersect") ? mc::VM_AND : mc::VM_OR);
            return;
        }
        if (e->k == Expr::SETENUM) {
            emit(mc::VM_PUSH, 0);
            for (const auto &x : e->a) { ex(x); emit(mc::VM_BIT); emit(mc::VM_OR); }
            return;
        }
        std::vector<long long> elems;
        if (const_set(e, elems)) {
            unsigned mask = 0;
            for (long long x : elems) { if (x < 0 || x > 31) cfail("only 0..31 can be members of a set value", e->pos); mask |= 1u << x; }
            emit(mc::VM_PUSH, (int)mask);
            return;
        }
        cfail("expected a set: a set variable, {...}, a constant set, or \\cup / \\cap / \\ of those", e->pos);
    }

    // an argument that names a whole sequence / set / set of records / function variable (or an element of an array of sequences)
    bool whole_variable(const EP &a) {
        if (seq_ref(a)) return true;
        if (a->k != Expr::ID) return false;
        for (const auto &b : binds) if (b.name == a->s) return false;
        auto vi = var_index.find(a->s);
        if (vi == var_index.end()) return false;
        const VarInfo &v = P.vars[(size_t)vi->second];
        return v.seq || v.set || v.rset || (v.array && !(proc && proc_locals.count(a->s)));
    }
    static EP subst_ids(const EP &e, std::map<std::string, EP> m) {
        if (!e) return e;
        if (e->k == Expr::ID) { auto it = m.find(e->s); return it == m.end() ? e : it->second; }
        auto c = std::make_shared<Expr>(*e);
        if ((e->k == Expr::QUANT || e->k == Expr::FUNCDEF || e->k == Expr::SETOF) && !e->bound.empty()) {
            if (!c->a.empty()) c->a[0] = subst_ids(e->a[0], m);
            m.erase(e->bound);
            for (size_t j = 1; j < c->a.size(); j++) c->a[j] = subst_ids(e->a[j], m);
            return c;
        }
        for (auto &x : c->a) x = subst_ids(x, m);
        return c;
    }
    // ---- expressions: leave one value on the stack
    void ex(const EP &e) {
        switch (e->k) {
        case Expr::NUM: emit(mc::VM_PUSH, (int)e->num); return;
        case Expr::BOOL: emit(mc::VM_PUSH, (int)e->num); return;
        case Expr::STR: emit(mc::VM_PUSH, intern(e->s)); return;
        case Expr::ID: {
            for (size_t i = binds.size(); i-- > 0;)
                if (binds[i].name == e->s) {
                    if (binds[i].is_const) emit(mc::VM_PUSH, (int)binds[i].value);
                    else emit(mc::VM_LOADT, binds[i].temp);
                    return;
                }
            if (e->s == "self") { push_self(e->pos); return; }
            if (e->s == "defaultInitValue") cfail("defaultInitValue can only be the (implicit) initial value of a variable, not part of an expression", e->pos);
            auto vi = var_index.find(e->s);
            if (vi != var_index.end()) {
                const VarInfo &v = P.vars[(size_t)vi->second];
                const bool per_self = proc && proc_locals.count(e->s) && proc->is_set && P.multi;
                if (v.defval && !v.array == !per_self && !copy_read) {  // reading a variable that still is defaultInitValue is an evaluation error, as in TLC
                    if (per_self) { push_self(e->pos); emit_indexed(mc::VM_LOADX, v, e->pos); } else emit(mc::VM_LOAD, v.base);
                    emit(mc::VM_PUSH, mc::VM_DEFAULT_INIT);
                    emit(mc::VM_NE);
                    const int ok = emit_jump(mc::VM_JNZ);
                    emit(mc::VM_FAIL);
                    patch(ok);
                }
                if (per_self) { push_self(e->pos); emit_indexed(mc::VM_LOADX, v, e->pos); return; }
                if (v.rset) cfail("the set of records `" + e->s + "` is used as a value here; supported: \\cup {r}, \\ {r}, r \\in, = {}, Cardinality, with / quantifiers over it", e->pos);
                if (v.set) cfail("the set `" + e->s + "` is used as a number here; supported: \\in, \\cup, \\cap, \\, =, #, \\subseteq, Cardinality, with / quantifiers over it", e->pos);
                if (v.seq) cfail("the sequence `" + e->s + "` is used as a value here; supported: Len, Head, " + e->s + "[i], = / # <<...>>", e->pos);
                if (v.array) cfail("the function `" + e->s + "` is used as a value; only `" + e->s + "[i]` is supported", e->pos);
                emit(mc::VM_LOAD, v.base);
                return;
            }
            long long cv;
            if (const_scalar(e, cv)) { emit(mc::VM_PUSH, (int)cv); return; }
            for (const auto &d : m.defs)
                if (d.name == e->s && d.params.empty()) {
                    if (++inline_depth > 16) cfail("definitions nest too deeply (recursion?)", e->pos);
                    const std::vector<Bind> saved = binds;  // a definition sees no local binding of its call site
                    binds.clear();
                    const Proc *sp = proc;
                    proc = nullptr;
                    emit(mc::VM_OLD_ON);   // a defined operator speaks about the unprimed variables
                    ex(d.body);
                    emit(mc::VM_OLD_OFF);
                    proc = sp;
                    binds = saved;
                    inline_depth--;
                    return;
                }
            cfail("unknown identifier `" + e->s + "`", e->pos);
        }
        case Expr::CALL: {  // an operator of the define block / of the module, inlined: arguments evaluated once
            if (e->s == "Cardinality" && e->a.size() == 1) {
                if (const VarInfo *rs = rset_var(e->a[0])) { emit(mc::VM_LOAD, rs->base); return; }
                ex_set(e->a[0]); emit(mc::VM_POPCNT); return;
            }
            if ((e->s == "Len" || e->s == "Head") && e->a.size() == 1) {
                const SeqRef q = seq_ref(e->a[0]);
                if (!q) cfail(e->s + " needs a sequence variable (or an element of an array of sequences)", e->pos);
                if (e->s == "Len") emit_len(q);
                else { emit(mc::VM_PUSH, 1); emit_seq(mc::VM_LOADSEQ, q); }
                return;
            }
            const Definition *def = nullptr;
            for (const auto &d : m.defs) if (d.name == e->s && d.params.size() == e->a.size()) def = &d;
            if (!def) cfail("unknown operator `" + e->s + "` with " + std::to_string(e->a.size()) + " argument(s)", e->pos);
            if (++inline_depth > 16) cfail("definitions nest too deeply (recursion?)", e->pos);
            std::vector<Bind> inner;
            const int temp0 = next_temp;
            std::map<std::string, EP> whole;   // parameters that stand for a whole sequence / set / function variable: substituted, not evaluated
            for (size_t k = 0; k < e->a.size(); k++) {
                long long cv;
                if (const_scalar(e->a[k], cv)) { inner.push_back({def->params[k], 0, true, cv}); continue; }
                if (!proc && whole_variable(e->a[k])) {   // (a state predicate: Last(q), IsSorted(box[i]), Size(msgs))
                    EP arg = e->a[k];
                    long long ci;
                    if (arg->k == Expr::INDEX && const_scalar(arg->a[1], ci)) {   // box[i], i a constant of the caller (an unrolled quantifier's variable): its value
                        auto c2 = std::make_shared<Expr>(*arg);
                        auto lit = std::make_shared<Expr>();
                        lit->k = Expr::NUM; lit->num = ci; lit->pos = arg->pos;
                        c2->a[1] = lit;
                        arg = c2;
                    } else if (arg->k == Expr::INDEX) {   // box[i]: i is the caller's — evaluated here, named inside
                        const int t = new_temp(e->pos);
                        ex(arg->a[1]);
                        emit(mc::VM_STORET, t);
                        const std::string nm = "\001a" + std::to_string(t);
                        inner.push_back({nm, t, false, 0});
                        auto c2 = std::make_shared<Expr>(*arg);
                        auto idn = std::make_shared<Expr>();
                        idn->k = Expr::ID; idn->s = nm; idn->pos = arg->pos;
                        c2->a[1] = idn;
                        arg = c2;
                    }
                    whole[def->params[k]] = arg;
                    continue;
                }
                const int t = new_temp(e->pos);
                ex(e->a[k]);
                emit(mc::VM_STORET, t);
                inner.push_back({def->params[k], t, false, 0});
            }
            const std::vector<Bind> saved = binds;
            binds = inner;
            const Proc *sp = proc;
            proc = nullptr;
            emit(mc::VM_OLD_ON);   // the arguments were evaluated in the caller's context; the body reads unprimed variables
            ex(whole.empty() ? def->body : subst_ids(def->body, whole));
            emit(mc::VM_OLD_OFF);
            proc = sp;
            binds = saved;
            next_temp = temp0;
            inline_depth--;
            return;
        }
        case Expr::INDEX: {
            if (const SeqRef q = seq_ref(e->a[0])) { ex(e->a[1]); emit_seq(mc::VM_LOADSEQ, q); return; }   // q[k], box[i][k]
            if (e->a[0]->k != Expr::ID) cfail("only `name[index]` is supported", e->pos);
            if (seq_var(e->a[0])) cfail("the sequence `" + e->a[0]->s + "[..]` is used as a value here; supported: Len, Head, " + e->a[0]->s + "[i][k], = / # <<...>>", e->pos);
            auto vi = var_index.find(e->a[0]->s);
            if (vi == var_index.end() || !P.vars[(size_t)vi->second].array) cfail("`" + e->a[0]->s + "` is not a function variable", e->pos);
            if (proc && proc_locals.count(e->a[0]->s) && proc->is_set && P.multi)
                cfail("process-local function variables are not supported (`" + e->a[0]->s + "`)", e->pos);
            ex(e->a[1]);
            emit_indexed(mc::VM_LOADX, P.vars[(size_t)vi->second], e->pos);
            return;
        }
        case Expr::DOT: {   // m.f of an element of a set of records (`with m \in msgs`, `\E m \in msgs`)
            const Bind *b = rs_bind(e->a[0]);
            if (!b) cfail("`." + e->s + "`: field access is compiled for the elements of a set of records only", e->pos);
            load_field(*b, e->s, e->pos);
            return;
        }
        case Expr::UNOP: ex(e->a[0]); emit(e->s == "~" ? mc::VM_NOT : mc::VM_NEG); return;
        case Expr::IF: {
            ex(e->a[0]);
            const int je = emit_jump(mc::VM_JZ);
            ex(e->a[1]);
            const int jend = emit_jump(mc::VM_JMP);
            patch(je);
            ex(e->a[2]);
            patch(jend);
            return;
        }
        case Expr::BINOP: binop(e); return;
        case Expr::QUANT: quant(e); return;
        default: cfail("this kind of expression is only supported as a constant set / initial value", e->pos);
        }
    }
    void binop(const EP &e) {
        const std::string &o = e->s;
        if ((o == "\\in" || o == "\\notin") && rset_var(e->a[1])) {   // [type |-> "ack", from |-> self] \in msgs
            const VarInfo &v = *rset_var(e->a[1]);
            push_record(v, e->a[0]);
            emit_rs(mc::VM_RSHAS, v);
            if (o == "\\notin") emit(mc::VM_NOT);
            return;
        }
        if (o == "=" || o == "#")
            for (int side = 0; side < 2; side++)
                if (const VarInfo *v = rset_var(e->a[(size_t)side])) {
                    const EP &other = e->a[(size_t)(1 - side)];
                    if (other->k != Expr::SETENUM || !other->a.empty()) cfail("a set of records can only be compared with {}", e->pos);
                    emit(mc::VM_LOAD, v->base); emit(mc::VM_PUSH, 0); emit(o == "=" ? mc::VM_EQ : mc::VM_NE);
                    return;
                }
        if (o == "/\\" || o == "\\/" || o == "=>") {
            ex(e->a[0]);
            const int j1 = emit_jump(o == "\\/" ? mc::VM_JNZ : mc::VM_JZ);
            ex(e->a[1]);
            const int jend = emit_jump(mc::VM_JMP);
            patch(j1);
            emit(mc::VM_PUSH, o == "/\\" ? 0 : 1);
            patch(jend);
            return;
        }
        if ((o == "\\in" || o == "\\notin") && dynamic_set(e->a[1])) {
            ex(e->a[0]); emit(mc::VM_BIT); ex_set(e->a[1]); emit(mc::VM_AND); emit(mc::VM_PUSH, 0); emit(o == "\\in" ? mc::VM_NE : mc::VM_EQ);
            return;
        }
        if (o == "<=>") {   // both sides are booleans (0 / 1)
            ex(e->a[0]); emit(mc::VM_PUSH, 0); emit(mc::VM_NE);
            ex(e->a[1]); emit(mc::VM_PUSH, 0); emit(mc::VM_NE);
            emit(mc::VM_EQ);
            return;
        }
        if ((o == "\\in" || o == "\\notin") && e->a[1]->k == Expr::ID && (e->a[1]->s == "Nat" || e->a[1]->s == "Int") && !var_index.count(e->a[1]->s)) {
            ex(e->a[0]);                               // x \\in Nat (a TypeOK conjunct): x >= 0; every cell is an integer
            if (e->a[1]->s == "Nat") { emit(mc::VM_PUSH, 0); emit(mc::VM_GE); }
            else { emit(mc::VM_POP); emit(mc::VM_PUSH, 1); }
            if (o == "\\notin") emit(mc::VM_NOT);
            return;
        }
        if (o == "\\in" || o == "\\notin") {
            const int t = new_temp(e->pos);
            ex(e->a[0]);
            emit(mc::VM_STORET, t);
            member(t, resolve_domain(e->a[1]));
            if (o == "\\notin") emit(mc::VM_NOT);
            next_temp--;
            return;
        }
        if ((o == "=" || o == "#") && (dynamic_set(e->a[0]) || dynamic_set(e->a[1]))) {
            ex_set(e->a[0]); ex_set(e->a[1]); emit(o == "=" ? mc::VM_EQ : mc::VM_NE);
            return;
        }
        if (o == "\\subseteq") { ex_set(e->a[0]); ex_set(e->a[1]); emit(mc::VM_ANDN); emit(mc::VM_PUSH, 0); emit(mc::VM_EQ); return; }
        if (o == "=" || o == "#") {
            for (int side = 0; side < 2; side++)
                if (const SeqRef q = seq_ref(e->a[(size_t)side]))
                    if (e->a[(size_t)(1 - side)]->k == Expr::TUPLE) { seq_equals(q, e->a[(size_t)(1 - side)], o == "#"); return; }
        }
        static const std::pair<const char *, int> ops[] = {{"=", mc::VM_EQ},  {"#", mc::VM_NE},  {"<", mc::VM_LT},    {">", mc::VM_GT},
                                                          {"<=", mc::VM_LE}, {">=", mc::VM_GE}, {"+", mc::VM_ADD},   {"-", mc::VM_SUB},
                                                          {"*", mc::VM_MUL}, {"%", mc::VM_MOD}, {"\\div", mc::VM_DIV}};
        for (const auto &p : ops)
            if (o == p.first) { ex(e->a[0]); ex(e->a[1]); emit(p.second); return; }
        cfail("operator `" + o + "` is not supported here", e->pos);
    }
    // value in temp t is a member of the set expression s
    void member(int t, const EP &s) {
        if (s->k == Expr::BINOP && s->s == "..") {
            emit(mc::VM_LOADT, t); ex(s->a[0]); emit(mc::VM_GE);
            const int jf = emit_jump(mc::VM_JZ);
            emit(mc::VM_LOADT, t); ex(s->a[1]); emit(mc::VM_LE);
            const int jend = emit_jump(mc::VM_JMP);
            patch(jf);
            emit(mc::VM_PUSH, 0);
            patch(jend);
            return;
        }
        std::vector<long long> elems;
        if (s->k == Expr::SETENUM) {  // elements may be state dependent
            std::vector<int> hits;
            for (const auto &x : s->a) { emit(mc::VM_LOADT, t); ex(x); emit(mc::VM_EQ); hits.push_back(emit_jump(mc::VM_JNZ)); }
            emit(mc::VM_PUSH, 0);
            const int jend = emit_jump(mc::VM_JMP);
            for (int h : hits) patch(h);
            emit(mc::VM_PUSH, 1);
            patch(jend);
            return;
        }
        if (!const_set(s, elems)) cfail("`\\in` needs an interval a..b, a set enumeration or a constant set", s->pos);
        std::vector<int> hits;
        for (long long x : elems) { emit(mc::VM_LOADT, t); emit(mc::VM_PUSH, (int)x); emit(mc::VM_EQ); hits.push_back(emit_jump(mc::VM_JNZ)); }
        emit(mc::VM_PUSH, 0);
        const int jend = emit_jump(mc::VM_JMP);
        for (int h : hits) patch(h);
        emit(mc::VM_PUSH, 1);
        patch(jend);
    }
    void quant(const EP &e) {
        const bool all = e->s == "\\A";
        const EP dom = resolve_domain(e->a[0]);
        if (e->s == "CHOOSE" && !(dom->k == Expr::BINOP && dom->s == "..")) {
            // CHOOSE x \in S : P over a set of small naturals (a set variable, {...}, a filter ...): the smallest member that satisfies P
            const int tx = new_temp(e->pos), tm = new_temp(e->pos);
            ex_set(dom); emit(mc::VM_STORET, tm);
            emit(mc::VM_PUSH, 0); emit(mc::VM_STORET, tx);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 31); emit(mc::VM_LE);
            const int jnone = emit_jump(mc::VM_JZ);
            emit(mc::VM_LOADT, tx); emit(mc::VM_BIT); emit(mc::VM_LOADT, tm); emit(mc::VM_AND);
            const int jskip = emit_jump(mc::VM_JZ);
            binds.push_back({e->bound, tx, false, 0});
            ex(e->a[1]);
            binds.pop_back();
            const int jhit = emit_jump(mc::VM_JNZ);
            patch(jskip);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jnone);
            emit(mc::VM_FAIL);
            patch(jhit);
            emit(mc::VM_LOADT, tx);
            next_temp -= 2;
            return;
        }
        if (e->s == "CHOOSE") {
            // CHOOSE x \in a..b : P — TLC takes the first element (ascending) that satisfies P and raises an error when none does
            const int tx = new_temp(e->pos), th = new_temp(e->pos);
            ex(dom->a[0]); emit(mc::VM_STORET, tx);
            ex(dom->a[1]); emit(mc::VM_STORET, th);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx); emit(mc::VM_LOADT, th); emit(mc::VM_LE);
            const int jnone = emit_jump(mc::VM_JZ);
            binds.push_back({e->bound, tx, false, 0});
            ex(e->a[1]);
            binds.pop_back();
            const int jhit = emit_jump(mc::VM_JNZ);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jnone);
            emit(mc::VM_FAIL);  // "Attempted to compute the value of CHOOSE x \in S: P, but no element of S satisfied P"
            patch(jhit);
            emit(mc::VM_LOADT, tx);
            next_temp -= 2;
            return;
        }
        // (an interval with constant bounds and at most four elements is unrolled like a constant set, further down: its bound variable is
        //  then a constant — no temporaries, and operators called with it take it as a constant too; nested quantifiers over small
        //  intervals, `\A b1 \in 1..NB : \A b2 \in 1..NB : \A v1 \in 1..NV : ...`, would otherwise exhaust the interpreter's eight)
        long long clo = 0, chi = -1;
        const bool small_const_interval = dom->k == Expr::BINOP && dom->s == ".." && const_scalar(dom->a[0], clo) && const_scalar(dom->a[1], chi) && chi - clo < 4;
        if (dom->k == Expr::BINOP && dom->s == ".." && !small_const_interval) {
            const int tx = new_temp(e->pos), th = new_temp(e->pos);
            ex(dom->a[0]); emit(mc::VM_STORET, tx);
            ex(dom->a[1]); emit(mc::VM_STORET, th);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx); emit(mc::VM_LOADT, th); emit(mc::VM_LE);
            const int jdone = emit_jump(mc::VM_JZ);
            binds.push_back({e->bound, tx, false, 0});
            ex(e->a[1]);
            binds.pop_back();
            const int jhit = emit_jump(all ? mc::VM_JZ : mc::VM_JNZ);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jdone);
            emit(mc::VM_PUSH, all ? 1 : 0);
            const int jend = emit_jump(mc::VM_JMP);
            patch(jhit);
            emit(mc::VM_PUSH, all ? 0 : 1);
            patch(jend);
            next_temp -= 2;
            return;
        }
        if (const VarInfo *rs = rset_var(dom)) {  // the elements of a set of records, by index (an expression changes nothing)
            const int tx = new_temp(e->pos);
            emit(mc::VM_PUSH, 0); emit(mc::VM_STORET, tx);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx); emit(mc::VM_LOAD, rs->base); emit(mc::VM_LT);
            const int jdone = emit_jump(mc::VM_JZ);
            Bind b{e->bound, tx, false, 0};
            b.rs = rs;
            binds.push_back(b);
            ex(e->a[1]);
            binds.pop_back();
            const int jhit = emit_jump(all ? mc::VM_JZ : mc::VM_JNZ);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jdone);
            emit(mc::VM_PUSH, all ? 1 : 0);
            const int jend = emit_jump(mc::VM_JMP);
            patch(jhit);
            emit(mc::VM_PUSH, all ? 0 : 1);
            patch(jend);
            next_temp -= 1;
            return;
        }
        if (dynamic_set(dom)) {  // x ranges over 0..31, the body counts only for members
            const int tx = new_temp(e->pos), tm = new_temp(e->pos);
            ex_set(dom); emit(mc::VM_STORET, tm);
            emit(mc::VM_PUSH, 0); emit(mc::VM_STORET, tx);
            const int loop = (int)c.size();
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 31); emit(mc::VM_LE);
            const int jdone = emit_jump(mc::VM_JZ);
            emit(mc::VM_LOADT, tx); emit(mc::VM_BIT); emit(mc::VM_LOADT, tm); emit(mc::VM_AND);
            const int jskip = emit_jump(mc::VM_JZ);
            binds.push_back({e->bound, tx, false, 0});
            ex(e->a[1]);
            binds.pop_back();
            const int jhit = emit_jump(all ? mc::VM_JZ : mc::VM_JNZ);
            patch(jskip);
            emit(mc::VM_LOADT, tx); emit(mc::VM_PUSH, 1); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
            emit(mc::VM_JMP, loop);
            patch(jdone);
            emit(mc::VM_PUSH, all ? 1 : 0);
            const int jend = emit_jump(mc::VM_JMP);
            patch(jhit);
            emit(mc::VM_PUSH, all ? 0 : 1);
            patch(jend);
            next_temp -= 2;
            return;
        }
        std::vector<long long> elems;
        if (!const_set(dom, elems)) cfail("a quantifier needs an interval, a constant set or a set variable as its domain", dom->pos);
        std::vector<int> hits;
        for (long long x : elems) {
            binds.push_back({e->bound, 0, true, x});
            ex(e->a[1]);
            binds.pop_back();
            hits.push_back(emit_jump(all ? mc::VM_JZ : mc::VM_JNZ));
        }
        emit(mc::VM_PUSH, all ? 1 : 0);
        const int jend = emit_jump(mc::VM_JMP);
        for (int h : hits) patch(h);
        emit(mc::VM_PUSH, all ? 0 : 1);
        patch(jend);
    }

    // CHOOSE over a constant set: leaves the chosen element on the stack; returns the number of alternatives
    unsigned long long choose_from(const EP &set, Pos p) {
        std::vector<long long> elems;
        if (!const_set(set, elems)) cfail("the set must be an interval with constant bounds or a constant set", set->pos);
        if (elems.empty()) cfail("empty set: the step would never be enabled", p);
        emit(mc::VM_CHOOSE, (int)elems.size());
        if (contiguous(elems)) { emit(mc::VM_PUSH, (int)elems[0]); emit(mc::VM_ADD); return elems.size(); }
        const int t = new_temp(p);
        emit(mc::VM_STORET, t);
        std::vector<int> ends;
        for (size_t k = 0; k < elems.size(); k++) {
            emit(mc::VM_LOADT, t); emit(mc::VM_PUSH, (int)k); emit(mc::VM_EQ);
            const int jn = emit_jump(mc::VM_JZ);
            emit(mc::VM_PUSH, (int)elems[k]);
            ends.push_back(emit_jump(mc::VM_JMP));
            patch(jn);
        }
        emit(mc::VM_FAIL);
        for (int x : ends) patch(x);
        next_temp--;
        return elems.size();
    }

    // `x \\in a..b` with state-dependent bounds: any of a .. a+31 that is <= b (a longer interval is reported as an
    // overflow, never cut short).  Leaves the chosen value on the stack; 32 alternatives.
    bool dynamic_interval(const EP &e) {
        if (!(e->k == Expr::BINOP && e->s == "..")) return false;
        long long a, b;
        return !(const_scalar(e->a[0], a) && const_scalar(e->a[1], b));
    }
    unsigned long long choose_interval(const EP &e, Pos p) {
        const int tl = new_temp(p), th = new_temp(p), tx = new_temp(p);
        ex(e->a[0]); emit(mc::VM_STORET, tl);
        ex(e->a[1]); emit(mc::VM_STORET, th);
        // more than 32 values: refuse (VM_BIT on the width - 1 overflows exactly then; an empty interval passes)
        emit(mc::VM_LOADT, th); emit(mc::VM_LOADT, tl); emit(mc::VM_GE);
        const int jempty = emit_jump(mc::VM_JZ);
        emit(mc::VM_LOADT, th); emit(mc::VM_LOADT, tl); emit(mc::VM_SUB); emit(mc::VM_BIT); emit(mc::VM_POP);
        patch(jempty);
        emit(mc::VM_CHOOSE, 32);
        emit(mc::VM_LOADT, tl); emit(mc::VM_ADD); emit(mc::VM_STORET, tx);
        emit(mc::VM_LOADT, tx); emit(mc::VM_LOADT, th); emit(mc::VM_LE); emit(mc::VM_AWAIT);
        emit(mc::VM_LOADT, tx);
        next_temp -= 3;
        return 32;
    }

    // ---- statements
    // the right-hand side of an assignment; `x := defaultInitValue` (what the expansion of a procedure's `return` assigns to a
    // parameter declared without a default: back to the model value it held before the call) is the one place where the name is a value
    void ex_rhs(const EP &e) {
        if (e->k == Expr::ID && e->s == "defaultInitValue") {
            bool bound = false;
            for (const auto &b : binds) bound |= b.name == e->s;
            if (!bound) { emit(mc::VM_PUSH, mc::VM_DEFAULT_INIT); return; }
        }
        if (e->k == Expr::ID && var_index.count(e->s)) {  // `x := y`: COPYING a variable that still holds defaultInitValue is not an error (a frame
            copy_read = true;                             //  of a recursive procedure saves its parameters as they are; TLC copies a model value too)
            ex(e);
            copy_read = false;
            return;
        }
        if (e->k == Expr::IF) {  // ... and the branches of a conditional right-hand side (the frame slots of a recursive procedure's
            ex(e->a[0]);         //     `return`: IF sp = K THEN defaultInitValue ELSE slot)
            const int je = emit_jump(mc::VM_JZ);
            ex_rhs(e->a[1]);
            const int jend = emit_jump(mc::VM_JMP);
            patch(je);
            ex_rhs(e->a[2]);
            patch(jend);
            return;
        }
        ex(e);
    }
    void assign(const SP &s) {
        if (!s->more.empty()) {  // a := e || b := f: evaluate every right-hand side (and index) first
            std::vector<SP> all{s};
            for (const auto &x : s->more) all.push_back(x);
            struct Saved { int t_idx, t_val; };
            std::vector<Saved> sv;
            std::vector<EP> seq_rhs;   // per assignment: the right-hand side of a whole-sequence assignment with its operands in temporaries (else null)
            int seq_taken = 0;
            for (const auto &x : all) {
                auto vi = var_index.find(x->var);
                if (vi == var_index.end()) cfail("assignment to `" + x->var + "`, which is not a variable of the algorithm", x->pos);
                const VarInfo &xv = P.vars[(size_t)vi->second];
                if (xv.set || xv.rset) cfail("`||` with a set variable is not supported", x->pos);
                if (xv.seq && (xv.array || !x->idx)) {
                    // a whole sequence (q, box[i]): its scalar operands are evaluated now, the sequence it starts from must be itself — and then
                    // nothing another assignment of the statement stores can be read by this one
                    Saved q{-1, -1};
                    if (x->idx) { q.t_idx = new_temp(x->pos); ex(x->idx); emit(mc::VM_STORET, q.t_idx); }
                    seq_rhs.push_back(seq_operands_to_temps(x->e, seq_taken));
                    sv.push_back(q);
                    continue;
                }
                seq_rhs.push_back(nullptr);
                Saved q{-1, new_temp(x->pos)};
                if (x->idx) { q.t_idx = new_temp(x->pos); ex(x->idx); emit(mc::VM_STORET, q.t_idx); }
                ex_rhs(x->e);
                emit(mc::VM_STORET, q.t_val);
                sv.push_back(q);
            }
            for (size_t k = 0; k < all.size(); k++) {  // then store, through single assignments of the saved values
                auto one = std::make_shared<Stmt>(*all[k]);
                one->more.clear();
                const std::string vname = "\001v" + std::to_string(k), iname = "\001i" + std::to_string(k);
                binds.push_back({vname, sv[k].t_val, false, 0});
                auto ve = std::make_shared<Expr>();
                ve->k = Expr::ID;
                ve->s = vname;
                ve->pos = one->pos;
                one->e = seq_rhs[k] ? seq_rhs[k] : ve;
                if (one->idx) {
                    binds.push_back({iname, sv[k].t_idx, false, 0});
                    auto ie = std::make_shared<Expr>(*ve);
                    ie->s = iname;
                    bound_orig[iname] = all[k]->idx;
                    one->idx = ie;
                }
                assign(one);
                binds.pop_back();
                if (all[k]->idx) { binds.pop_back(); bound_orig.erase(iname); }
            }
            for (int k = 0; k < seq_taken; k++) binds.pop_back();   // (pushed before the \001v / \001i names, which are gone by now)
            next_temp -= seq_taken;
            for (const auto &q : sv) { if (q.t_val >= 0) next_temp--; if (q.t_idx >= 0) next_temp--; }
            return;
        }
        auto vi = var_index.find(s->var);
        if (vi == var_index.end() || s->var == "pc") cfail("assignment to `" + s->var + "`, which is not a variable of the algorithm", s->pos);
        const VarInfo &v = P.vars[(size_t)vi->second];
        if (v.rset) {
            for (const auto &pr : m.procs) for (const auto &l : pr.locals) if (l.name == s->var && &pr != proc) cfail("`" + s->var + "` cannot be assigned here", s->pos);
            if (s->idx) cfail("a set variable cannot be indexed", s->pos);
            assign_rset(v, s->e);
            return;
        }
        if (v.set) {
            for (const auto &pr : m.procs) for (const auto &l : pr.locals) if (l.name == s->var && &pr != proc) cfail("`" + s->var + "` cannot be assigned here", s->pos);
            if (s->idx) cfail("a set variable cannot be indexed", s->pos);
            if (proc && proc_locals.count(s->var) && proc->is_set && P.multi) { push_self(s->pos); ex_set(s->e); emit_indexed(mc::VM_STOREX, v, s->pos); }
            else { ex_set(s->e); emit(mc::VM_STORE, v.base); }
            return;
        }
        if (v.seq) {
            for (const auto &pr : m.procs) for (const auto &l : pr.locals) if (l.name == s->var && &pr != proc) cfail("`" + s->var + "` cannot be assigned here", s->pos);
            if (v.array) {   // box[i] := <sequence expression>
                if (!s->idx) cfail("assigning the whole array of sequences `" + s->var + "` is not supported: assign its elements (" + s->var + "[i] := Append(" + s->var + "[i], e), ...)", s->pos);
                assign_seq(SeqRef{&v, s->idx, s->idx}, s->e);
            } else if (s->idx) { ex(s->idx); ex(s->e); emit_seq(mc::VM_STORESEQ, SeqRef{&v, nullptr, nullptr}); }
            else assign_seq(SeqRef{&v, nullptr, nullptr}, s->e);
            return;
        }
        const bool self_indexed = proc && proc_locals.count(s->var) && proc->is_set && P.multi;
        bool global_or_own = self_indexed || !v.array || s->idx;
        for (const auto &pr : m.procs) for (const auto &l : pr.locals) if (l.name == s->var && &pr != proc) global_or_own = false;
        if (!global_or_own) cfail("`" + s->var + "` cannot be assigned here", s->pos);
        if (self_indexed) {
            if (s->idx) cfail("process-local function variables are not supported", s->pos);
            push_self(s->pos);
            ex_rhs(s->e);
            emit_indexed(mc::VM_STOREX, v, s->pos);
        } else if (s->idx) {
            if (!v.array) cfail("`" + s->var + "` is not a function variable", s->pos);
            ex(s->idx);
            ex(s->e);
            emit_indexed(mc::VM_STOREX, v, s->pos);
        } else {
            if (v.array) cfail("assigning a whole function (`" + s->var + " := ...`) is not supported: assign its elements", s->pos);
            ex_rhs(s->e);
            emit(mc::VM_STORE, v.base);
        }
    }
    // `await e` is the conjunct e of the action.  TLC evaluates an action formula disjunct by disjunct: for
    // `A \\/ B` it continues once for every disjunct that holds (two successors — equal states — when both do), for a
    // bounded `\\E x \\in S : p` once per witness, `/\\` and IF are walked in the same mode, and anything else is a
    // boolean.  The alternatives become choice digits so that `states generated` counts what TLC counts.
    unsigned long long await_action(const EP &e) {
        if (e->k == Expr::BINOP && e->s == "\\/") {
            std::vector<EP> ds;
            std::function<void(const EP &)> flat = [&](const EP &x) {
                if (x->k == Expr::BINOP && x->s == "\\/") { flat(x->a[0]); flat(x->a[1]); }
                else ds.push_back(x);
            };
            flat(e);
            const int t = new_temp(e->pos);
            emit(mc::VM_CHOOSE, (int)ds.size());
            emit(mc::VM_STORET, t);
            std::vector<int> ends;
            unsigned long long mx = 1;
            for (size_t k = 0; k < ds.size(); k++) {
                emit(mc::VM_LOADT, t); emit(mc::VM_PUSH, (int)k); emit(mc::VM_EQ);
                const int jn = emit_jump(mc::VM_JZ);
                mx = std::max(mx, await_action(ds[k]));
                ends.push_back(emit_jump(mc::VM_JMP));
                patch(jn);
            }
            emit(mc::VM_FAIL);
            for (int x : ends) patch(x);
            next_temp--;
            return mx * ds.size();
        }
        if (e->k == Expr::BINOP && e->s == "/\\") return await_action(e->a[0]) * await_action(e->a[1]);
        if (e->k == Expr::IF) {
            ex(e->a[0]);
            const int je = emit_jump(mc::VM_JZ);
            const unsigned long long a = await_action(e->a[1]);
            const int jend = emit_jump(mc::VM_JMP);
            patch(je);
            const unsigned long long b = await_action(e->a[2]);
            patch(jend);
            return std::max(a, b);
        }
        if (e->k == Expr::QUANT && e->s == "\\E") {
            std::vector<long long> elems;
            if (dynamic_set(e->a[0])) {  // a set variable: any of 0..31 that is a member
                const int t = new_temp(e->pos);
                emit(mc::VM_CHOOSE, 32);
                emit(mc::VM_STORET, t);
                emit(mc::VM_LOADT, t); emit(mc::VM_BIT); ex_set(e->a[0]); emit(mc::VM_AND); emit(mc::VM_AWAIT);
                binds.push_back({e->bound, t, false, 0});
                const unsigned long long b = await_action(e->a[1]);
                binds.pop_back();
                next_temp--;
                return 32 * b;
            }
            if (dynamic_interval(e->a[0])) {  // e.g. 1..Len(q): one successor per witness among the first 32 values
                const int t = new_temp(e->pos);
                const unsigned long long n = choose_interval(e->a[0], e->pos);
                emit(mc::VM_STORET, t);
                binds.push_back({e->bound, t, false, 0});
                const unsigned long long b = await_action(e->a[1]);
                binds.pop_back();
                next_temp--;
                return n * b;
            }
            if (const_set(e->a[0], elems) && !elems.empty()) {  // one successor per witness
                const int t = new_temp(e->pos);
                const unsigned long long n = choose_from(e->a[0], e->pos);
                emit(mc::VM_STORET, t);
                binds.push_back({e->bound, t, false, 0});
                const unsigned long long b = await_action(e->a[1]);
                binds.pop_back();
                next_temp--;
                return n * b;
            }
        }
        ex(e);
        emit(mc::VM_AWAIT);
        return 1;
    }

    // a statement with no label inside; returns the number of alternatives it introduces
    unsigned long long nolabel(const SP &s) {
        switch (s->k) {
        case Stmt::ASSIGN: assign(s); return 1;
        case Stmt::AWAIT: return await_action(s->e);
        case Stmt::ASSERT:
            if (s->var.rfind("$stack ", 0) == 0) {
                // the bounded call stack of a recursive procedure is full (pcal.cpp expand_procedures): a CAPACITY limit, reported like a
                // sequence that outgrows its cells (R_OVERFLOW -> MC_EOVERFLOW), not as an assertion of the algorithm.
                // With the instructions there are: VM_BIT refuses an element beyond 31 with R_OVERFLOW — (~cond) * 32 is 0 or 32.
                ex(s->e);
                emit(mc::VM_NOT);
                emit(mc::VM_PUSH, 32);
                emit(mc::VM_MUL);
                emit(mc::VM_BIT);
                emit(mc::VM_POP);
                return 1;
            }
            ex(s->e);
            emit(mc::VM_ASSERT, (int)P.asserts.size());
            P.asserts.push_back({s->pos.line, s->pos.col});
            return 1;
        case Stmt::SKIP: case Stmt::PRINT: return 1;
        case Stmt::IF: {
            ex(s->e);
            const int je = emit_jump(mc::VM_JZ);
            const unsigned long long a = block(s->blocks[0]);
            const int jend = emit_jump(mc::VM_JMP);
            patch(je);
            const unsigned long long b = block(s->blocks[1]);
            patch(jend);
            return std::max(a, b);
        }
        case Stmt::EITHER: {
            const int t = new_temp(s->pos);
            emit(mc::VM_CHOOSE, (int)s->blocks.size());
            emit(mc::VM_STORET, t);
            std::vector<int> ends;
            unsigned long long mx = 1;
            for (size_t b = 0; b < s->blocks.size(); b++) {
                emit(mc::VM_LOADT, t); emit(mc::VM_PUSH, (int)b); emit(mc::VM_EQ);
                const int jn = emit_jump(mc::VM_JZ);
                mx = std::max(mx, block(s->blocks[b]));
                ends.push_back(emit_jump(mc::VM_JMP));
                patch(jn);
            }
            emit(mc::VM_FAIL);
            for (int x : ends) patch(x);
            next_temp--;
            return mx * s->blocks.size();
        }
        case Stmt::WITH: {
            const int t = new_temp(s->pos);
            unsigned long long n = 1;
            if (!s->with_eq && rset_var(s->e)) {   // any element of a set of records: its fields are COPIED (the body may change the set)
                const VarInfo &v = *rset_var(s->e);
                emit(mc::VM_CHOOSE, v.cap);
                emit(mc::VM_STORET, t);
                emit(mc::VM_LOADT, t); emit(mc::VM_LOAD, v.base); emit(mc::VM_LT); emit(mc::VM_AWAIT);
                Bind b{s->var, t, false, 0};
                b.rs = &v;
                Bind byidx = b;
                // (only the fields the body reads are copied: a message of five fields of which a step looks at two costs two temporaries)
                std::set<std::string> used;
                std::function<void(const EP &)> in_expr = [&](const EP &e) {
                    if (!e) return;
                    if (e->k == Expr::DOT && e->a[0]->k == Expr::ID && e->a[0]->s == s->var) used.insert(e->s);
                    for (const auto &x : e->a) in_expr(x);
                };
                std::function<void(const std::vector<SP> &)> in_stmts = [&](const std::vector<SP> &v2) {
                    for (const auto &x : v2) {
                        in_expr(x->e);
                        in_expr(x->idx);
                        for (const auto &o : x->more) { in_expr(o->e); in_expr(o->idx); }
                        for (const auto &bl : x->blocks) in_stmts(bl);
                    }
                };
                in_stmts(s->blocks[0]);
                int taken = 0;
                for (size_t k = 0; k < v.fields.size(); k++) {
                    if (!used.count(v.fields[k])) { b.ftemps.push_back(-1); continue; }
                    const int tf = new_temp(s->pos);
                    taken++;
                    load_field(byidx, v.fields[k], s->pos);
                    emit(mc::VM_STORET, tf);
                    b.ftemps.push_back(tf);
                }
                binds.push_back(b);
                const unsigned long long nb = block(s->blocks[0]);
                binds.pop_back();
                next_temp -= 1 + taken;
                return (unsigned long long)v.cap * nb;
            }
            const EP wdom = s->with_eq ? s->e : resolve_domain(s->e);   // (with x \\in DOMAIN f / DOMAIN q)
            if (s->with_eq) ex(s->e);
            else if (dynamic_set(wdom)) {  // any of 0..31, enabled only for the members
                emit(mc::VM_CHOOSE, 32);
                emit(mc::VM_STORET, t);
                emit(mc::VM_LOADT, t); emit(mc::VM_BIT); ex_set(wdom); emit(mc::VM_AND); emit(mc::VM_AWAIT);
                emit(mc::VM_LOADT, t);
                n = 32;
            } else if (dynamic_interval(wdom)) n = choose_interval(wdom, s->pos);
            else n = choose_from(wdom, s->pos);
            emit(mc::VM_STORET, t);
            binds.push_back({s->var, t, false, 0});
            const unsigned long long b = block(s->blocks[0]);
            binds.pop_back();
            next_temp--;
            return n * b;
        }
        case Stmt::GOTO: case Stmt::WHILE:
            cfail(std::string(s->k == Stmt::GOTO ? "goto" : "while") + " must end its step: put a label after the enclosing statement", s->pos);
        case Stmt::CALL: case Stmt::RETURN:
            cfail("internal: a `call` / `return` survived the expansion of procedures", s->pos);
        }
        return 1;
    }
    unsigned long long block(const std::vector<SP> &v) {
        unsigned long long ch = 1;
        for (const auto &s : v) ch *= nolabel(s);
        return ch;
    }
    void set_pc(const std::string &label, Pos p) {
        auto it = str_id.find(label);
        if (it == str_id.end() || it->second >= P.nlabels) cfail("unknown label `" + label + "`", p);
        emit(mc::VM_SETPC, it->second);
        emit(mc::VM_HALT);
    }
    // statements [i, n) of v inside one action; cont = label reached when the sequence ends
    unsigned long long seq(const std::vector<SP> &v, size_t i, const std::string &cont, bool top) {
        unsigned long long ch = 1;
        for (size_t j = i; j < v.size(); j++) {
            const SP &s = v[j];
            if (!s->label.empty() && !(top && j == i)) { set_pc(s->label, s->pos); return ch; }
            const std::string after = j + 1 < v.size() ? v[j + 1]->label : cont;
            switch (s->k) {
            case Stmt::GOTO: set_pc(s->var, s->pos); return ch;
            case Stmt::WHILE: {
                if (!(top && j == i)) cfail("a while needs a label", s->pos);
                ex(s->e);
                const int je = emit_jump(mc::VM_JZ);
                const unsigned long long a = seq(s->blocks[0], 0, s->label, false);
                patch(je);
                const unsigned long long b = seq(v, j + 1, cont, false);
                return ch * std::max(a, b);
            }
            case Stmt::IF:
                if (has_label(s)) {
                    if (after.empty()) cfail("the statement after this if needs a label (a label occurs inside it)", s->pos);
                    ex(s->e);
                    const int je = emit_jump(mc::VM_JZ);
                    const unsigned long long a = seq(s->blocks[0], 0, after, false);
                    patch(je);
                    const unsigned long long b = seq(s->blocks[1], 0, after, false);
                    return ch * std::max(a, b);
                }
                ch *= nolabel(s);
                break;
            case Stmt::EITHER:
                if (has_label(s)) {
                    if (after.empty()) cfail("the statement after this either needs a label (a label occurs inside it)", s->pos);
                    const int t = new_temp(s->pos);
                    emit(mc::VM_CHOOSE, (int)s->blocks.size());
                    emit(mc::VM_STORET, t);
                    next_temp--;  // consumed before any branch runs
                    unsigned long long mx = 1;
                    for (size_t b = 0; b < s->blocks.size(); b++) {
                        emit(mc::VM_LOADT, t); emit(mc::VM_PUSH, (int)b); emit(mc::VM_EQ);
                        const int jn = emit_jump(mc::VM_JZ);
                        mx = std::max(mx, seq(s->blocks[b], 0, after, false));
                        patch(jn);
                    }
                    emit(mc::VM_FAIL);
                    return ch * mx * s->blocks.size();
                }
                ch *= nolabel(s);
                break;
            case Stmt::WITH:
                if (has_label(s)) cfail("labels inside `with` are not allowed", s->pos);
                ch *= nolabel(s);
                break;
            default: ch *= nolabel(s); break;
            }
        }
        set_pc(cont, v.empty() ? Pos{} : v.back()->pos);
        return ch;
    }

    struct Site { const std::vector<SP> *seq; size_t idx; std::string cont; };
    void sites(const std::vector<SP> &v, const std::string &cont, std::vector<Site> &out) {
        for (size_t j = 0; j < v.size(); j++) {
            const SP &s = v[j];
            const std::string after = j + 1 < v.size() ? v[j + 1]->label : cont;
            if (!s->label.empty()) out.push_back({&v, j, cont});
            for (const auto &b : s->blocks) sites(b, s->k == Stmt::WHILE ? s->label : after, out);
        }
    }

    // ---- initial values
    void init_scalar(const VarDecl &d, int slot, unsigned long long &ninit) {
        if (d.no_init) emit(mc::VM_PUSH, mc::VM_DEFAULT_INIT);
        else if (d.in_set) ninit *= choose_from(d.init, d.pos);
        else ex(d.init);
        emit(mc::VM_STORE, slot);
        if (ninit > (1ull << 40)) cfail("too many initial states", d.pos);
    }

    void run() {
        P.module = m.name;
        P.multi = !(m.procs.size() == 1 && m.procs[0].name.empty());
        for (const auto &kv : cfg.constants) consts[kv.first] = kv.second;
        if (!consts.count("defaultInitValue")) {  // TLC needs `defaultInitValue = defaultInitValue` in the cfg; here it is implied
            ConstVal dv;
            dv.k = ConstVal::STR;
            dv.s = "defaultInitValue";
            consts["defaultInitValue"] = dv;
        }
        for (const auto &cn : m.constants) if (!consts.count(cn)) cfail("CONSTANT " + cn + " has no value in the configuration");
        // labels first: label id == string id
        std::vector<std::string> labels;
        for (const auto &p : m.procs) all_labels(p.body, labels);
        for (const auto &l : labels) { if (str_id.count(l)) cfail("label `" + l + "` is used twice"); intern(l); }
        if (str_id.count("Done")) cfail("`Done` cannot be used as a label");
        const int done = intern("Done");
        P.nlabels = (int)P.strings.size();
        // process instances
        struct Inst { const Proc *p; long long self; };
        std::vector<Inst> insts;
        std::map<const Proc *, std::vector<long long>> ids_of;
        for (const auto &p : m.procs) {
            std::vector<long long> ids;
            if (!P.multi) ids.push_back(0);
            else if (p.is_set) { if (!const_set(p.id, ids)) cfail("the identifier set of process " + p.name + " must be a constant set", p.id->pos); }
            else { long long v; if (!const_scalar(p.id, v)) cfail("the identifier of process " + p.name + " must be a constant", p.id->pos); ids.push_back(v); }
            if (ids.empty()) cfail("process " + p.name + " has no instance");
            for (long long id : ids) {
                for (const auto &x : insts) if (x.self == id) cfail("two processes share the identifier " + std::to_string(id));
                insts.push_back({&p, id});
                procset.push_back(id);
            }
            ids_of[&p] = ids;
            if (p.body.empty() || p.body[0]->label.empty()) cfail("the first statement of " + (p.name.empty() ? std::string("the algorithm") : "process " + p.name) + " needs a label");
        }
        // instances in ascending identifier order: pc[i] can then be indexed whenever ProcSet is an integer interval
        std::stable_sort(insts.begin(), insts.end(), [](const Inst &a, const Inst &b) { return a.self < b.self; });
        procset.clear();
        for (const auto &x : insts) procset.push_back(x.self);
        P.ninst = (int)insts.size();
        // variables: globals, pc, process locals (the VARIABLES order of the translation)
        int nv = 0;
        auto add_var = [&](const std::string &name, bool array, const std::vector<long long> &ids, char type) {
            if (var_index.count(name)) cfail("variable `" + name + "` is declared twice");
            VarInfo v;
            v.name = name;
            v.array = array;
            v.base = nv;
            v.ids = ids;
            v.type = type;
            nv += array ? (int)ids.size() : 1;
            var_index[name] = (int)P.vars.size();
            P.vars.push_back(v);
        };
        std::vector<std::string> untyped_seqs;   // sequences that start empty: typed by what is put into them, once every variable is known
        auto decl_var = [&](const VarDecl &d, const Proc *owner) {
            const bool per_inst = owner && owner->is_set && P.multi;
            if (d.no_init) {  // a scalar whose type is that of its first assignment (fixed below, once every variable is known)
                if (per_inst) add_var(d.name, true, ids_of[owner], 'i');
                else add_var(d.name, false, {}, 'i');
                P.vars.back().defval = true;
            } else if (d.init->k == Expr::SETENUM && !d.in_set && rset_record(d.name)) {   // a SET of records: a count cell + fields x cells
                const RecordVar &rv = *rset_record(d.name);
                if (per_inst) cfail("a set of records local to a process SET is not supported (`" + d.name + "`)", d.pos);
                add_var(d.name, false, {}, 'i');
                VarInfo &v = P.vars.back();
                v.rset = true;
                v.fields = rv.fields;
                v.ftypes = std::string(rv.fields.size(), 'i');   // (fixed below, once every variable is known)
                v.cap = seq_cap();
                nv += v.cap * (int)v.fields.size();
            } else if ((d.init->k == Expr::SETENUM || (d.init->k == Expr::BINOP && d.init->s == "..")) && !d.in_set) {
                // a set of small naturals / strings — {...}, or an interval a..b of constants as its initial value (`free = 3..K`): one mask cell (per instance)
                const char et = d.init->k == Expr::BINOP || d.init->a.empty() ? 'i' : type_of(d.init->a[0]);
                if (per_inst) add_var(d.name, true, ids_of[owner], et);
                else add_var(d.name, false, {}, et);
                P.vars.back().set = true;
            } else if (d.init->k == Expr::TUPLE && !d.in_set) {  // a sequence
                if (per_inst) cfail("sequence variables local to a process SET are not supported (`" + d.name + "`)", d.pos);
                add_var(d.name, false, {}, d.init->a.empty() ? 'i' : type_of(d.init->a[0]));
                VarInfo &v = P.vars.back();
                v.seq = true;
                v.cap = seq_cap();
                nv += v.cap;  // add_var counted the Len cell
                if (d.init->a.empty()) untyped_seqs.push_back(d.name);
            } else if (d.init->k == Expr::FUNCDEF && !d.in_set && d.init->a[1]->k == Expr::TUPLE) {   // an ARRAY of sequences: the channels of a message-passing algorithm
                if (per_inst) cfail("process-local function variables are not supported (`" + d.name + "`)", d.pos);
                std::vector<long long> dom;
                if (!const_set(d.init->a[0], dom)) cfail("the domain of `" + d.name + "` must be a constant set", d.pos);
                if (!contiguous(dom)) cfail("the domain of the array of sequences `" + d.name + "` must be an integer interval", d.pos);
                add_var(d.name, true, dom, d.init->a[1]->a.empty() ? 'i' : type_of(d.init->a[1]->a[0]));
                if (d.init->a[1]->a.empty()) untyped_seqs.push_back(d.name);
                VarInfo &v = P.vars.back();
                v.seq = true;
                v.cap = seq_cap();
                nv += v.cap * (int)dom.size();  // add_var counted one (Len) cell per element
            } else if (d.init->k == Expr::FUNCDEF && !d.in_set) {
                if (per_inst) cfail("process-local function variables are not supported (`" + d.name + "`)", d.pos);
                std::vector<long long> dom;
                if (!const_set(d.init->a[0], dom)) cfail("the domain of `" + d.name + "` must be a constant set", d.pos);
                add_var(d.name, true, dom, type_of(d.init->a[1]));
            } else if (per_inst) {
                add_var(d.name, true, ids_of[owner], type_of(d.init));
            } else {
                add_var(d.name, false, {}, type_of(d.init));
            }
        };
        for (const auto &g : m.globals) decl_var(g, nullptr);
        P.pc_base = nv;
        add_var("pc", P.multi, P.multi ? procset : std::vector<long long>{}, 's');
        for (const auto &p : m.procs) for (const auto &l : p.locals) decl_var(l, &p);
        // types that come from what the algorithm assigns: sequences that start empty, variables without an initial value (a procedure's
        // parameters), the fields of a set of records — each may be typed by another, so: sequences, then variables, sequences again, sets
        auto type_seqs = [&]() { for (const auto &name : untyped_seqs) P.vars[(size_t)var_index[name]].type = seq_elem_type(name); };
        type_seqs();
        for (auto &v : P.vars) {
            if (!v.defval) continue;
            const Expr *rhs = nullptr;
            for (const auto &p : m.procs) if (!rhs) rhs = first_assignment(p.body, v.name);
            if (!rhs) continue;  // never assigned: it stays defaultInitValue
            if (rhs->k == Expr::SETENUM || rhs->k == Expr::TUPLE || rhs->k == Expr::FUNCDEF)
                cfail("`" + v.name + "` is declared without an initial value and later holds a set / sequence / function: give it an initial value of that kind");
            auto probe = std::make_shared<Expr>(*rhs);
            v.type = type_of(probe);
        }
        type_seqs();
        for (auto &v : P.vars) if (v.rset) v.ftypes = rset_field_types(v);
        if (nv > mc::SpecVm::MAX_VARS) cfail("the state has " + std::to_string(nv) + " scalar variables; at most " + std::to_string(mc::SpecVm::MAX_VARS) + " are supported");
        P.nv = nv;
        // image: header, label table, self table, code
        c.assign((size_t)mc::VMH_SIZE, 0);
        const int label_tab = (int)c.size();
        c.resize(c.size() + (size_t)P.nlabels, -1);
        const int self_tab = (int)c.size();
        for (const auto &x : insts) c.push_back((int)x.self);
        // ---- Init
        const int init_entry = (int)c.size();
        unsigned long long ninit = 1;
        auto init_decl = [&](const VarDecl &d, const Proc *owner) {
            const VarInfo &v = P.vars[(size_t)var_index[d.name]];
            const bool per_inst = owner && owner->is_set && P.multi;
            if (v.rset) {
                if (owner && P.multi) { have_self_const = true; self_const = ids_of[owner][0]; }
                assign_rset(v, d.init);
                have_self_const = false;
            } else if (v.set) {
                for (size_t k = 0; k < (v.array ? v.ids.size() : (size_t)1); k++) {
                    if (owner && P.multi) { have_self_const = true; self_const = v.array ? v.ids[k] : ids_of[owner][0]; }
                    ex_set(d.init);
                    emit(mc::VM_STORE, v.base + (int)k);
                    have_self_const = false;
                }
            } else if (v.seq && v.array) {
                for (size_t k = 0; k < v.ids.size(); k++) {
                    binds.push_back({d.init->bound, 0, true, v.ids[k]});
                    auto at = std::make_shared<Expr>();
                    at->k = Expr::NUM;
                    at->num = v.ids[k];
                    at->pos = d.pos;
                    assign_seq(SeqRef{&v, at, at}, d.init->a[1]);
                    binds.pop_back();
                }
            } else if (v.seq) {
                if (owner && P.multi) { have_self_const = true; self_const = ids_of[owner][0]; }
                assign_seq(SeqRef{&v, nullptr, nullptr}, d.init);
                have_self_const = false;
            } else if (per_inst) {
                for (size_t k = 0; k < v.ids.size(); k++) {
                    have_self_const = true;
                    self_const = v.ids[k];
                    init_scalar(d, v.base + (int)k, ninit);
                    have_self_const = false;
                }
            } else if (v.array) {
                for (size_t k = 0; k < v.ids.size(); k++) {
                    binds.push_back({d.init->bound, 0, true, v.ids[k]});
                    ex(d.init->a[1]);
                    binds.pop_back();
                    emit(mc::VM_STORE, v.base + (int)k);
                }
            } else {
                if (owner && P.multi) { have_self_const = true; self_const = ids_of[owner][0]; }
                init_scalar(d, v.base, ninit);
                have_self_const = false;
            }
        };
        for (const auto &g : m.globals) init_decl(g, nullptr);
        for (const auto &p : m.procs) for (const auto &l : p.locals) init_decl(l, &p);
        for (size_t k = 0; k < insts.size(); k++) { emit(mc::VM_PUSH, str_id[insts[k].p->body[0]->label]); emit(mc::VM_STORE, P.pc_base + (int)k); }
        emit(mc::VM_HALT);
        P.num_init = ninit;
        // ---- actions
        unsigned long long maxch = 1;
        for (const auto &p : m.procs) {
            proc = &p;
            proc_locals.clear();
            for (const auto &l : p.locals) proc_locals.insert(l.name);
            if (P.multi && !p.is_set) { have_self_const = true; self_const = ids_of[&p][0]; }
            std::vector<Site> ss;
            sites(p.body, "Done", ss);
            for (const auto &site : ss) {
                const SP &s = (*site.seq)[site.idx];
                c[(size_t)(label_tab + str_id[s->label])] = (int)c.size();
                next_temp = 0;
                maxch = std::max(maxch, seq(*site.seq, site.idx, site.cont, true));
            }
            have_self_const = false;
            proc = nullptr;
        }
        proc_locals.clear();
        if (maxch * (unsigned long long)P.ninst + 1 > 255)
            cfail("too many alternatives per state (" + std::to_string(maxch) + " per process step x " + std::to_string(P.ninst) + " processes > 254)");
        P.maxch = (int)maxch;
        if (max_depth > mc::SpecVm::STACK) cfail("an expression is too deeply nested for the interpreter's stack (" + std::to_string(max_depth) + " > " + std::to_string(mc::SpecVm::STACK) + ")");
        // ---- invariants
        if (cfg.invariants.size() + cfg.constraints.size() > 8) cfail("at most 8 invariants + constraints are supported");
        std::vector<int> inv_entry;
        for (int pass = 0; pass < 2; pass++)  // INVARIANTs, then CONSTRAINTs: both are state predicates named by the cfg
            for (const auto &name : pass == 0 ? cfg.invariants : cfg.constraints) {
                const Definition *def = nullptr;
                for (const auto &d : m.defs) if (d.name == name && d.params.empty()) def = &d;
                if (!def) cfail(std::string(pass == 0 ? "INVARIANT " : "CONSTRAINT ") + name + " is not a definition of the module this front-end can read");
                inv_entry.push_back((int)c.size());
                next_temp = 0;
                ex(def->body);
                emit(mc::VM_HALT);
                if (pass == 0) P.invariants.push_back(name);
            }
        // ---- header
        c[mc::VMH_MAGIC] = mc::VM_MAGIC;
        c[mc::VMH_NV] = nv;
        c[mc::VMH_NINST] = P.ninst;
        c[mc::VMH_MAXCH] = P.maxch;
        c[mc::VMH_PC_BASE] = P.pc_base;
        c[mc::VMH_DONE] = done;
        c[mc::VMH_INIT_ENTRY] = init_entry;
        c[mc::VMH_NINV] = (int)cfg.invariants.size();
        c[mc::VMH_NCON] = (int)cfg.constraints.size();
        for (size_t k = 0; k < inv_entry.size(); k++) c[(size_t)mc::VMH_INV0 + k] = inv_entry[k];
        c[mc::VMH_LABEL_TAB] = label_tab;
        c[mc::VMH_SELF_TAB] = self_tab;
        c[mc::VMH_NLABELS] = P.nlabels;
        c[mc::VMH_NUM_INIT_LO] = (int)(uint32_t)ninit;
        c[mc::VMH_NUM_INIT_HI] = (int)(uint32_t)(ninit >> 32);
        c[mc::VMH_CODE_LEN] = (int)c.size();
    }
};

std::string fmt_val(const Program &P, char type, int32_t v);
std::string fmt_set(const Program &P, char type, int32_t mask) {  // elements in TLC's order: numbers ascending, strings sorted
    std::vector<std::string> items;
    for (int b = 0; b < 32; b++) if ((uint32_t)mask >> b & 1u) items.push_back(fmt_val(P, type, b));
    if (type == 's') std::sort(items.begin(), items.end());
    std::string s = "{";
    for (size_t i = 0; i < items.size(); i++) s += (i ? ", " : "") + items[i];
    return s + "}";
}
std::string fmt_val(const Program &P, char type, int32_t v) {
    if (v == mc::VM_DEFAULT_INIT) return "defaultInitValue";  // a model value: TLC prints it bare
    if (type == 'b') return v ? "TRUE" : "FALSE";
    if (type == 's') return v >= 0 && (size_t)v < P.strings.size() ? "\"" + P.strings[(size_t)v] + "\"" : "\"?\"";
    return std::to_string(v);
}

}  // namespace

std::string compile(const Module &m, const std::string &module_text, const Config &cfg, Program &out) {
    out = Program();
    const std::string tr = translate(m);
    if (tr.rfind("\\* TRANSLATION ERROR: ", 0) == 0) return tr.substr(strlen("\\* TRANSLATION ERROR: "), tr.size() - strlen("\\* TRANSLATION ERROR: ") - 1);
    out.translated = transpile_text(module_text, m);
    try {
        Compiler cc(m, cfg, out);
        cc.run();
    } catch (const CompileError &e) {
        return e.msg;
    }
    return "";
}

}  // namespace pcal

// ------------------------------------------------------------------------------------------------ spec_vm.h host side
namespace mc {

int vm_make_params(const int64_t *p, unsigned np, VmParams &o) {
    if (np < 1 || !p[0]) return -1;
    const pcal::Program *P = (const pcal::Program *)(intptr_t)p[0];
    if (P->magic != VM_MAGIC || P->image.size() < (size_t)VMH_SIZE || P->image[VMH_MAGIC] != VM_MAGIC) return -1;
    const int32_t *c = P->image.data();
    memset(&o, 0, sizeof o);
    o.code = c;
    o.host = P;
    o.nv = c[VMH_NV];
    o.words = (o.nv + 1) / 2;
    o.ninst = c[VMH_NINST];
    o.maxch = c[VMH_MAXCH];
    o.pc_base = c[VMH_PC_BASE];
    o.done = c[VMH_DONE];
    o.init_entry = c[VMH_INIT_ENTRY];
    o.ninv = c[VMH_NINV];
    o.ncon = c[VMH_NCON];
    for (int k = 0; k < 8; k++) o.inv_entry[k] = c[VMH_INV0 + k];
    o.label_tab = c[VMH_LABEL_TAB];
    o.self_tab = c[VMH_SELF_TAB];
    o.code_len = c[VMH_CODE_LEN];
    o.num_init = (uint64_t)(uint32_t)c[VMH_NUM_INIT_LO] | (uint64_t)(uint32_t)c[VMH_NUM_INIT_HI] << 32;
    return 0;
}

int vm_format(const void *host, const int32_t *vals, char *buf, size_t cap) {
    const pcal::Program &P = *(const pcal::Program *)host;
    std::string s;
    for (const auto &v : P.vars) {
        if (!s.empty()) s += "\n";
        s += "/\\ " + v.name + " = ";
        auto seq_at = [&](int base) {
            std::string t = "<<";
            for (int k = 0; k < vals[base] && k < v.cap; k++) t += (k ? ", " : "") + pcal::fmt_val(P, v.type, vals[base + 1 + k]);
            return t + ">>";
        };
        if (v.rset) {
            // a set of records as the evaluator of the test suite prints it: fields in name order, elements ascending by their fields in that
            // order (numbers by value, strings by text).  TLC's own order of a record's fields is not pinned by anything in the reference
            std::vector<size_t> fo(v.fields.size());
            for (size_t k = 0; k < fo.size(); k++) fo[k] = k;
            std::sort(fo.begin(), fo.end(), [&](size_t a, size_t b) { return v.fields[a] < v.fields[b]; });
            std::vector<int> el((size_t)std::max(0, std::min(vals[v.base], v.cap)));
            for (size_t i = 0; i < el.size(); i++) el[i] = (int)i;
            auto cell = [&](int i, size_t f) { return vals[v.base + 1 + (int)f * v.cap + i]; };
            std::sort(el.begin(), el.end(), [&](int a, int b) {
                for (size_t f : fo) {
                    const int32_t x = cell(a, f), y = cell(b, f);
                    if (x == y) continue;
                    if (v.ftypes[f] == 's') {
                        const std::string sx = x >= 0 && (size_t)x < P.strings.size() ? P.strings[(size_t)x] : "", sy = y >= 0 && (size_t)y < P.strings.size() ? P.strings[(size_t)y] : "";
                        return sx < sy;
                    }
                    return x < y;
                }
                return false;
            });
            s += "{";
            for (size_t i = 0; i < el.size(); i++) {
                s += i ? ", [" : "[";
                for (size_t k = 0; k < fo.size(); k++) s += (k ? ", " : "") + v.fields[fo[k]] + " |-> " + pcal::fmt_val(P, v.ftypes[fo[k]], cell(el[i], fo[k]));
                s += "]";
            }
            s += "}";
            continue;
        }
        if (v.seq && !v.array) { s += seq_at(v.base); continue; }
        auto one = [&](int k) {   // element k of the array (or the variable itself)
            if (v.seq) return seq_at(v.base + k * (v.cap + 1));
            const int32_t x = vals[v.base + k];
            return v.set ? pcal::fmt_set(P, v.type, x) : pcal::fmt_val(P, v.type, x);
        };
        if (!v.array) { s += one(0); continue; }
        // TLC prints a function whose domain is 1..n as a tuple, any other as (k :> v @@ ...) in ascending key order
        std::vector<size_t> order(v.ids.size());
        for (size_t k = 0; k < order.size(); k++) order[k] = k;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return v.ids[a] < v.ids[b]; });
        bool seq = true;
        for (size_t k = 0; k < order.size(); k++) seq &= v.ids[order[k]] == (long long)k + 1;
        if (seq) {
            s += "<<";
            for (size_t k = 0; k < order.size(); k++) s += (k ? ", " : "") + one((int)order[k]);
            s += ">>";
        } else {
            s += "(";
            for (size_t k = 0; k < order.size(); k++)
                s += (k ? " @@ " : "") + std::to_string(v.ids[order[k]]) + " :> " + one((int)order[k]);
            s += ")";
        }
    }
    const size_t n = s.size() < cap ? s.size() : (cap ? cap - 1 : 0);
    if (cap) { memcpy(buf, s.data(), n); buf[n] = 0; }
    return (int)n;
}

// action id = label id of the instance that moves; nlabels = the terminating disjunct
int vm_action_of(const void *host, const int32_t *parent_vals, int slot) {
    const pcal::Program &P = *(const pcal::Program *)host;
    if (slot < 0) return -1;
    if (slot >= P.ninst * P.maxch) return P.nlabels;
    return parent_vals[P.pc_base + slot / P.maxch];
}

const char *vm_action_name(const void *host, int action) {
    const pcal::Program &P = *(const pcal::Program *)host;
    if (action < 0) return "Initial predicate";
    if (action >= P.nlabels) return "Terminating";
    return P.strings[(size_t)action].c_str();
}

// the assertion that fails when the state `vals` is expanded: its index (Program::asserts) or -1; *label = the label
// (action) being evaluated
int vm_failed_assert(const void *host, const int32_t *vals, int *label) {
    const pcal::Program &P = *(const pcal::Program *)host;
    VmParams prm;
    const int64_t h = (int64_t)(intptr_t)host;
    if (vm_make_params(&h, 1, prm)) return -1;
    for (int slot = 0; slot < P.ninst * P.maxch; slot++) {
        const int inst = slot / P.maxch;
        const int32_t lab = vals[P.pc_base + inst];
        if (lab == prm.done) continue;
        int32_t v[SpecVm::MAX_VARS], res;
        for (int i = 0; i < P.nv; i++) v[i] = vals[i];
        int aux = 0;
        const int r = SpecVm::run(prm, prm.code[prm.label_tab + lab], prm.code[prm.self_tab + inst], inst, (uint64_t)(slot % P.maxch), v, res, aux, vals);
        if (r == SpecVm::R_ASSERT) { if (label) *label = lab; return aux; }
    }
    return -1;
}

}  // namespace mc
