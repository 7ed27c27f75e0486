// tlaeval.cpp — general TLA+ evaluator on the host (see tlaeval.h).  One translation unit: values, lexer, parser, module
// loading (EXTENDS / INSTANCE), the evaluator (value mode and action mode), cfg binding and TLC's breadth-first search.
//
// Semantics followed (TLC is the external Java tool of reference Makefile:6-7; p-manual.pdf section 4, *Specifying Systems*
// ch. 14 as quoted in SURVEY.md App. B; operators: examples/SpecifyingSystems/Standard/{Naturals,Sequences,FiniteSets}.tla,
// TLC/TLC.tla; cfg grammar: TLC/ConfigFileGrammar.tla:4-32):
//   * an action is evaluated left to right; `x' = e` ASSIGNS x' when x' has no value yet and is an equality TEST otherwise;
//     `x' \in S` enumerates; UNCHANGED <<a, b>> is a' = a /\ b' = b; `\/`, `\E`, IF, CASE and LET bodies branch; operator
//     applications are expanded with lazily evaluated arguments; a successor with an unassigned variable is an error;
//   * breadth-first search with exact de-duplication on whole states; counters as TLC prints them (README.md:319-321):
//     generated = initial states + every successor produced (duplicates and out-of-CONSTRAINT ones included); a successor
//     outside the CONSTRAINT is invariant-checked but neither stored nor expanded; deadlock = no successor at all;
//   * CHOOSE takes the first satisfying element in a fixed total order on values (model values in cfg order).
#include "tlaeval.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

#include "../../include/tlamc.h"

namespace tlaeval {
namespace {

struct TlaError {  // an evaluation error TLC would report (function applied outside its domain, CHOOSE without witness, ...)
    std::string msg;
    bool is_assert = false;
};
[[noreturn]] void fail(const std::string &m) { throw TlaError{m, false}; }
struct SyntaxErr { std::string msg; };

// =============================================================================================== values
struct Val;
using V = std::shared_ptr<const Val>;
enum Kind { K_BOOL = 0, K_INT = 1, K_STR = 2, K_MV = 3, K_TUPLE = 4, K_FN = 5, K_SET = 6, K_LAZY = 7 };
enum LazyKind { L_NAT, L_INT, L_STRING, L_SEQ, L_POWERSET, L_FNSET, L_RECSET };
struct Val {
    Kind k = K_BOOL;
    long i = 0;                              // bool / int / model-value index / lazy kind
    std::string s;                           // string / model-value name
    std::vector<V> items;                    // tuple elements, or the elements of a set (sorted by cmp, unique)
    std::vector<std::pair<V, V>> fn;         // function: (key, value) sorted by key; a record is a function over strings
    V a, b;                                  // lazy sets: element set / domain, range
    std::vector<std::pair<std::string, V>> fields;  // L_RECSET
};
std::string fmt(const V &v);

int cmp(const V &x, const V &y);
int cmp_vec(const std::vector<V> &a, const std::vector<V> &b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = 0; i < a.size(); i++) { const int c = cmp(a[i], b[i]); if (c) return c; }
    return 0;
}
// a total order on values (CHOOSE takes the first satisfying element in it): booleans, integers, strings, model values (cfg
// order), sequences (by length, then elementwise), functions, sets
int cmp(const V &x, const V &y) {
    if (x.get() == y.get()) return 0;
    if (x->k != y->k) return x->k < y->k ? -1 : 1;
    switch (x->k) {
        case K_BOOL: case K_INT: case K_MV: return x->i < y->i ? -1 : x->i > y->i ? 1 : 0;
        case K_STR: { const int c = x->s.compare(y->s); return c < 0 ? -1 : c > 0 ? 1 : 0; }
        case K_TUPLE: case K_SET: return cmp_vec(x->items, y->items);
        case K_FN:
            if (x->fn.size() != y->fn.size()) return x->fn.size() < y->fn.size() ? -1 : 1;
            for (size_t i = 0; i < x->fn.size(); i++) {
                int c = cmp(x->fn[i].first, y->fn[i].first);
                if (c) return c;
                if ((c = cmp(x->fn[i].second, y->fn[i].second))) return c;
            }
            return 0;
        default: {  // sets that are not enumerated (Nat, Seq(S), ...): structurally
            if (x->i != y->i) return x->i < y->i ? -1 : 1;
            if ((bool)x->a != (bool)y->a) return x->a ? 1 : -1;
            if (x->a) { const int c = cmp(x->a, y->a); if (c) return c; }
            if ((bool)x->b != (bool)y->b) return x->b ? 1 : -1;
            if (x->b) { const int c = cmp(x->b, y->b); if (c) return c; }
            if (x->fields.size() != y->fields.size()) return x->fields.size() < y->fields.size() ? -1 : 1;
            for (size_t i = 0; i < x->fields.size(); i++) {
                int c = x->fields[i].first.compare(y->fields[i].first);
                if (c) return c < 0 ? -1 : 1;
                if ((c = cmp(x->fields[i].second, y->fields[i].second))) return c;
            }
            return 0;
        }
    }
}
bool eq(const V &x, const V &y) {
    if (x->k == K_LAZY || y->k == K_LAZY) fail("equality of sets that are not enumerated");
    return x->k == y->k && cmp(x, y) == 0;
}
struct VLess { bool operator()(const V &a, const V &b) const { return cmp(a, b) < 0; } };

V g_true, g_false, g_empty_tuple, g_empty_set;
V mk_bool(bool b) { return b ? g_true : g_false; }
V g_small_ints[1 + 16 + 1024];  // -16 .. 1023: most integers a specification computes are counters and indices
V mk_int(long i) {
    if (i >= -16 && i < 1024 && g_small_ints[i + 16]) return g_small_ints[i + 16];
    auto v = std::make_shared<Val>(); v->k = K_INT; v->i = i; return v;
}
V mk_str(const std::string &s) { auto v = std::make_shared<Val>(); v->k = K_STR; v->s = s; return v; }
V mk_tuple(std::vector<V> items) { if (items.empty()) return g_empty_tuple; auto v = std::make_shared<Val>(); v->k = K_TUPLE; v->items = std::move(items); return v; }
V normal_set(const V &s);
V mk_set(std::vector<V> items) {  // sorts and removes duplicates; an element that is a lazily represented finite set is enumerated
    if (items.empty()) return g_empty_set;
    for (auto &x : items) if (x->k == K_LAZY) x = normal_set(x);
    std::sort(items.begin(), items.end(), VLess());
    items.erase(std::unique(items.begin(), items.end(), [](const V &a, const V &b) { return cmp(a, b) == 0; }), items.end());
    auto v = std::make_shared<Val>(); v->k = K_SET; v->items = std::move(items); return v;
}
V mk_set_sorted(std::vector<V> items) { if (items.empty()) return g_empty_set; auto v = std::make_shared<Val>(); v->k = K_SET; v->items = std::move(items); return v; }
// a function from its (key, value) pairs: domain 1..n makes it a sequence (the empty function is <<>>)
V mk_fn(std::vector<std::pair<V, V>> kv) {
    if (kv.empty()) return g_empty_tuple;
    std::sort(kv.begin(), kv.end(), [](const std::pair<V, V> &a, const std::pair<V, V> &b) { return cmp(a.first, b.first) < 0; });
    bool seq = true;
    for (size_t i = 0; i < kv.size() && seq; i++) seq = kv[i].first->k == K_INT && kv[i].first->i == (long)i + 1;
    if (seq) { std::vector<V> it; it.reserve(kv.size()); for (auto &p : kv) it.push_back(p.second); return mk_tuple(std::move(it)); }
    auto v = std::make_shared<Val>(); v->k = K_FN; v->fn = std::move(kv); return v;
}
std::vector<std::string> g_mv_names;
std::map<std::string, V> g_mvs;
V mk_mv(const std::string &name) {
    auto it = g_mvs.find(name);
    if (it != g_mvs.end()) return it->second;
    auto v = std::make_shared<Val>(); v->k = K_MV; v->s = name; v->i = (long)g_mvs.size();
    g_mvs[name] = v;
    return v;
}
V mk_lazy(LazyKind lk, V a = nullptr, V b = nullptr) { auto v = std::make_shared<Val>(); v->k = K_LAZY; v->i = lk; v->a = a; v->b = b; return v; }

bool is_fn(const V &v) { return v->k == K_TUPLE || v->k == K_FN; }
V fn_domain(const V &f) {
    std::vector<V> d;
    if (f->k == K_TUPLE) { for (size_t i = 0; i < f->items.size(); i++) d.push_back(mk_int((long)i + 1)); return mk_set_sorted(d); }
    if (f->k == K_FN) { for (auto &p : f->fn) d.push_back(p.first); return mk_set_sorted(d); }
    fail("DOMAIN of a non-function " + fmt(f));
}
const V *fn_find(const V &f, const V &a) {
    if (f->k == K_TUPLE) { if (a->k == K_INT && a->i >= 1 && a->i <= (long)f->items.size()) return &f->items[(size_t)a->i - 1]; return nullptr; }
    if (f->k == K_FN) {
        auto it = std::lower_bound(f->fn.begin(), f->fn.end(), a, [](const std::pair<V, V> &p, const V &k) { return cmp(p.first, k) < 0; });
        if (it != f->fn.end() && cmp(it->first, a) == 0) return &it->second;
        return nullptr;
    }
    fail("applying a non-function " + fmt(f) + " to " + fmt(a));
}
V fn_apply(const V &f, const V &a) {
    const V *r = fn_find(f, a);
    if (!r) fail("function applied outside its domain: " + fmt(a) + " not in DOMAIN " + fmt(f));
    return *r;
}
std::vector<std::pair<V, V>> fn_items(const V &f) {
    if (f->k == K_FN) return f->fn;
    std::vector<std::pair<V, V>> out;
    if (f->k == K_TUPLE) { for (size_t i = 0; i < f->items.size(); i++) out.emplace_back(mk_int((long)i + 1), f->items[i]); return out; }
    fail("not a function: " + fmt(f));
}

bool set_in(const V &v, const V &s);
void enumerate(const V &s, const std::function<bool(const V &)> &each);  // each returns false to stop
std::vector<V> elements(const V &s) {  // in the order the set enumerates (sorted for explicit sets)
    if (s->k == K_SET) return s->items;
    std::vector<V> out;
    enumerate(s, [&](const V &x) { out.push_back(x); return true; });
    return out;
}
V to_set(const V &s) { return s->k == K_SET ? s : mk_set(elements(s)); }
V normal_set(const V &s) {
    switch ((LazyKind)s->i) { case L_POWERSET: case L_FNSET: case L_RECSET: return to_set(s); default: return s; }
}
bool set_in(const V &v, const V &s) {
    if (s->k == K_SET) return std::binary_search(s->items.begin(), s->items.end(), v, VLess());
    if (s->k != K_LAZY) fail("\\in applied to a non-set " + fmt(s));
    switch ((LazyKind)s->i) {
        case L_NAT: return v->k == K_INT && v->i >= 0;
        case L_INT: return v->k == K_INT;
        case L_STRING: return v->k == K_STR;
        case L_SEQ: if (v->k != K_TUPLE) return false; for (auto &x : v->items) if (!set_in(x, s->a)) return false; return true;
        case L_POWERSET: if (v->k != K_SET) return false; for (auto &x : v->items) if (!set_in(x, s->a)) return false; return true;
        case L_FNSET: {
            if (!is_fn(v)) return false;
            if (!eq(fn_domain(v), to_set(s->a))) return false;
            for (auto &p : fn_items(v)) if (!set_in(p.second, s->b)) return false;
            return true;
        }
        case L_RECSET: {
            if (v->k != K_FN || v->fn.size() != s->fields.size()) return false;
            for (auto &f : s->fields) {
                const V *x = fn_find(v, mk_str(f.first));
                if (!x || !set_in(*x, f.second)) return false;
            }
            return true;
        }
    }
    return false;
}
void enumerate(const V &s, const std::function<bool(const V &)> &each) {
    if (s->k == K_SET) { for (auto &x : s->items) if (!each(x)) return; return; }
    if (s->k != K_LAZY) fail("enumerating a non-set " + fmt(s));
    switch ((LazyKind)s->i) {
        case L_POWERSET: {  // by size, then combinations of the sorted elements
            const std::vector<V> el = to_set(s->a)->items;
            const size_t n = el.size();
            for (size_t r = 0; r <= n; r++) {
                std::vector<size_t> idx(r);
                for (size_t i = 0; i < r; i++) idx[i] = i;
                for (;;) {
                    std::vector<V> sub;
                    for (size_t i : idx) sub.push_back(el[i]);
                    if (!each(mk_set_sorted(sub))) return;
                    size_t i = r;
                    while (i > 0 && idx[i - 1] == n - r + i - 1) i--;
                    if (i == 0) break;
                    idx[i - 1]++;
                    for (size_t j = i; j < r; j++) idx[j] = idx[j - 1] + 1;
                }
            }
            return;
        }
        case L_FNSET: {
            const std::vector<V> dom = to_set(s->a)->items, rng = to_set(s->b)->items;
            if (dom.empty()) { each(g_empty_tuple); return; }
            if (rng.empty()) return;
            std::vector<size_t> c(dom.size(), 0);
            for (;;) {
                std::vector<std::pair<V, V>> kv;
                for (size_t i = 0; i < dom.size(); i++) kv.emplace_back(dom[i], rng[c[i]]);
                if (!each(mk_fn(kv))) return;
                size_t i = dom.size();
                while (i > 0) { if (++c[i - 1] < rng.size()) break; c[i - 1] = 0; i--; }
                if (i == 0) return;
            }
        }
        case L_RECSET: {
            std::vector<std::vector<V>> sets;
            for (auto &f : s->fields) { sets.push_back(to_set(f.second)->items); if (sets.back().empty()) return; }
            std::vector<size_t> c(sets.size(), 0);
            for (;;) {
                std::vector<std::pair<V, V>> kv;
                for (size_t i = 0; i < sets.size(); i++) kv.emplace_back(mk_str(s->fields[i].first), sets[i][c[i]]);
                if (!each(mk_fn(kv))) return;
                size_t i = sets.size();
                while (i > 0) { if (++c[i - 1] < sets[i - 1].size()) break; c[i - 1] = 0; i--; }
                if (i == 0) return;
            }
        }
        default: fail("cannot enumerate " + fmt(s));
    }
}

// canonical TLA+ text of a value: records with fields in alphabetical order, functions as (k :> v @@ ...) and sets sorted by
// text, sequences <<...>>, the empty function <<>> (the format of mc_state_format and of the oracle's printers)
std::string fmt(const V &v) {
    switch (v->k) {
        case K_BOOL: return v->i ? "TRUE" : "FALSE";
        case K_INT: return std::to_string(v->i);
        case K_STR: return "\"" + v->s + "\"";
        case K_MV: return v->s;
        case K_TUPLE: { std::string o = "<<"; for (size_t i = 0; i < v->items.size(); i++) o += (i ? ", " : "") + fmt(v->items[i]); return o + ">>"; }
        case K_FN: {
            bool rec = true;
            for (auto &p : v->fn) rec = rec && p.first->k == K_STR;
            if (rec) { std::string o = "["; for (size_t i = 0; i < v->fn.size(); i++) o += (i ? ", " : "") + v->fn[i].first->s + " |-> " + fmt(v->fn[i].second); return o + "]"; }
            std::vector<std::string> parts;
            for (auto &p : v->fn) parts.push_back(fmt(p.first) + " :> " + fmt(p.second));
            std::sort(parts.begin(), parts.end());
            std::string o = "(";
            for (size_t i = 0; i < parts.size(); i++) o += (i ? " @@ " : "") + parts[i];
            return o + ")";
        }
        case K_SET: {
            std::vector<std::string> parts;
            for (auto &x : v->items) parts.push_back(fmt(x));
            std::sort(parts.begin(), parts.end());
            std::string o = "{";
            for (size_t i = 0; i < parts.size(); i++) o += (i ? ", " : "") + parts[i];
            return o + "}";
        }
        case K_LAZY:
            switch ((LazyKind)v->i) { case L_NAT: return "Nat"; case L_INT: return "Int"; case L_STRING: return "STRING"; case L_SEQ: return "Seq(" + fmt(v->a) + ")";
                case L_POWERSET: return "SUBSET " + fmt(v->a); case L_FNSET: return "[" + fmt(v->a) + " -> " + fmt(v->b) + "]"; default: return "[record set]"; }
    }
    return "?";
}
// compact structural key of a value (the seen-set compares these: exact de-duplication on whole states)
void ser(const V &v, std::string &o) {
    switch (v->k) {
        case K_BOOL: o += v->i ? 'T' : 'F'; break;
        case K_INT: o += 'i'; o += std::to_string(v->i); o += ';'; break;
        case K_STR: o += 's'; o += v->s; o += '\0'; break;
        case K_MV: o += 'm'; o += std::to_string(v->i); o += ';'; break;
        case K_TUPLE: o += '<'; for (auto &x : v->items) ser(x, o); o += '>'; break;
        case K_SET: o += '{'; for (auto &x : v->items) ser(x, o); o += '}'; break;
        case K_FN: o += '('; for (auto &p : v->fn) { ser(p.first, o); ser(p.second, o); } o += ')'; break;
        case K_LAZY: ser(to_set(v), o); break;
    }
}

// =============================================================================================== lexer
struct Tok { enum T { ID, NUM, STR, SYM, SEP, END } k = END; std::string s; int line = 0, col = 0; };
const char *SYMS[] = {"<=>", "|->", "-+->", "::=", ":=", "==", "=>", "=<", "<=", ">=", "/=", "/\\", "\\/", "..", "->", "<-", "<<", ">>", ":>",
                      "@@", "[]", "<>", "~>", "||", "(", ")", "[", "]", "{", "}", ",", ";", ":", "+", "-", "*", "/", "%", "=", "<", ">", "#",
                      "~", "'", "!", "@", ".", "^", "|", "&", "\\"};
std::vector<Tok> lex(const std::string &t) {
    std::vector<Tok> out;
    size_t i = 0, n = t.size();
    int line = 1, col = 1;
    auto adv = [&](size_t k) { for (size_t q = 0; q < k && i < n; q++) { if (t[i] == '\n') { line++; col = 1; } else col++; i++; } };
    auto starts = [&](const char *p) { return t.compare(i, strlen(p), p) == 0; };
    while (i < n) {
        const char c = t[i];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { adv(1); continue; }
        if (starts("\\*")) { while (i < n && t[i] != '\n') adv(1); continue; }
        if (starts("(*")) {
            int depth = 1;
            adv(2);
            while (i < n && depth) { if (starts("(*")) { depth++; adv(2); } else if (starts("*)")) { depth--; adv(2); } else adv(1); }
            continue;
        }
        Tok tk; tk.line = line; tk.col = col;
        if (starts("----") || starts("====")) {
            size_t j = i; while (j < n && t[j] == c) j++;
            tk.k = Tok::SEP; tk.s = t.substr(i, j - i); out.push_back(tk); adv(j - i); continue;
        }
        if (isalnum((unsigned char)c) || c == '_') {
            size_t j = i; while (j < n && (isalnum((unsigned char)t[j]) || t[j] == '_')) j++;
            tk.s = t.substr(i, j - i);
            bool num = true; for (char ch : tk.s) num = num && isdigit((unsigned char)ch);
            tk.k = num ? Tok::NUM : Tok::ID; out.push_back(tk); adv(j - i); continue;
        }
        if (c == '"') {
            size_t j = i + 1; std::string buf;
            while (j < n && t[j] != '"') { if (t[j] == '\\' && j + 1 < n) j++; buf += t[j]; j++; }
            tk.k = Tok::STR; tk.s = buf; out.push_back(tk); adv(j + 1 - i); continue;
        }
        if (c == '\\' && i + 1 < n && isalpha((unsigned char)t[i + 1])) {
            size_t j = i + 1; while (j < n && isalpha((unsigned char)t[j])) j++;
            tk.k = Tok::SYM; tk.s = t.substr(i, j - i); out.push_back(tk); adv(j - i); continue;
        }
        bool found = false;
        for (const char *s : SYMS) if (starts(s)) { tk.k = Tok::SYM; tk.s = s; out.push_back(tk); adv(strlen(s)); found = true; break; }
        if (!found) throw SyntaxErr{"unexpected character '" + std::string(1, c) + "' at line " + std::to_string(line) + ", column " + std::to_string(col)};
    }
    Tok e; e.k = Tok::END; e.line = line; out.push_back(e);
    return out;
}

// =============================================================================================== syntax tree and parser
enum NK { N_NUM, N_STR, N_BOOL, N_ID, N_CALL, N_PAREN, N_AT, N_CONJ, N_DISJ, N_OP, N_NOT, N_NEG, N_QUANT, N_CHOOSE, N_CHOOSE_UNB, N_IF, N_CASE,
          N_LET, N_LAMBDA, N_UNCHANGED, N_ENABLED, N_PRE, N_SETENUM, N_SETFILTER, N_SETMAP, N_TUPLE, N_RECORD, N_RECORDSET, N_FNDEF, N_FNSET,
          N_EXCEPT, N_IDX, N_PRIME, N_TEMPORAL, N_INSTANCE, N_NTH };
struct Node;
using NodeP = std::shared_ptr<Node>;
struct PInfo;
struct Bound { std::vector<std::string> names; bool is_tuple = false; NodeP dom; mutable std::vector<int> syms; };
struct Def { std::string name; std::vector<std::pair<std::string, int>> params; NodeP body; int line = 0; mutable std::vector<int> psyms; mutable int sym = -1;
             int l1 = 0, c1 = 0, l2 = 0, c2 = 0; };  // the body's first and last character (what TLC prints as the location of an action)
struct Node {
    NK k;
    std::string s;            // identifier / operator / keyword
    long num = 0;
    int line = 0;
    std::vector<NodeP> kids;
    std::vector<Bound> bounds;
    std::vector<Def> defs;                                          // LET
    std::vector<std::string> names;                                 // LAMBDA parameters
    std::vector<std::pair<std::vector<NodeP>, NodeP>> ups;          // EXCEPT: (path, value)
    std::vector<std::pair<std::string, NodeP>> fields;              // record / record set / INSTANCE substitutions
    // caches filled by the evaluator
    mutable int sym = -1, opc = -1, uv_state = 0;
    mutable V lit;
    mutable std::vector<int> uvars, psyms;
    mutable std::vector<V> keys;   // N_RECORD: the field names as values
    mutable std::shared_ptr<PInfo> pinfo;
};
NodeP node(NK k, int line = 0) { auto n = std::make_shared<Node>(); n->k = k; n->line = line; return n; }
NodeP node_id(const std::string &s) { auto n = node(N_ID); n->s = s; return n; }

const std::map<std::string, int> PREC = {{"=>", 1}, {"<=>", 2}, {"\\equiv", 2}, {"~>", 2}, {"-+->", 2}, {"\\/", 3}, {"\\lor", 3}, {"/\\", 3}, {"\\land", 3},
    {"=", 5}, {"#", 5}, {"/=", 5}, {"<", 5}, {">", 5}, {"<=", 5}, {"=<", 5}, {">=", 5}, {"\\leq", 5}, {"\\geq", 5}, {"\\in", 5}, {"\\notin", 5},
    {"\\subseteq", 5}, {"\\subset", 5}, {"@@", 6}, {":>", 7}, {"\\cup", 8}, {"\\union", 8}, {"\\cap", 8}, {"\\intersect", 8}, {"\\", 8},
    {"..", 9}, {"+", 10}, {"-", 10}, {"%", 11}, {"\\X", 11}, {"\\times", 11}, {"*", 13}, {"/", 13}, {"\\div", 13}, {"\\o", 13}, {"\\circ", 13},
    {"\\cdot", 5}, {"^", 14}};
const std::map<std::string, std::string> CANON = {{"=<", "<="}, {"\\leq", "<="}, {"\\geq", ">="}, {"/=", "#"}, {"\\union", "\\cup"}, {"\\intersect", "\\cap"},
    {"\\lor", "\\/"}, {"\\land", "/\\"}, {"\\equiv", "<=>"}, {"\\times", "\\X"}, {"\\circ", "\\o"}};
const std::set<std::string> KEYWORDS = {"MODULE", "EXTENDS", "CONSTANT", "CONSTANTS", "VARIABLE", "VARIABLES", "ASSUME", "ASSUMPTION", "AXIOM", "THEOREM",
    "LEMMA", "PROPOSITION", "COROLLARY", "RECURSIVE", "INSTANCE", "WITH", "LOCAL", "IF", "THEN", "ELSE", "CASE", "OTHER",
    "LET", "IN", "CHOOSE", "LAMBDA", "EXCEPT", "UNCHANGED", "ENABLED", "SUBSET", "UNION", "DOMAIN", "PROOF", "BY", "OBVIOUS", "OMITTED", "QED"};

struct Parser {
    std::vector<Tok> t;
    size_t i = 0;
    std::vector<int> jstack{0};  // columns of the enclosing junction-list bullets: a token at or left of the top ends the item
    const Tok &cur() const { return t[i]; }
    const Tok &peek(size_t k = 1) const { return t[std::min(i + k, t.size() - 1)]; }
    bool is_sym(const char *s) const { return t[i].k == Tok::SYM && t[i].s == s; }
    bool is_id(const char *s) const { return t[i].k == Tok::ID && t[i].s == s; }
    [[noreturn]] void failp(const std::string &what) const {
        throw SyntaxErr{what + " at line " + std::to_string(cur().line) + ", column " + std::to_string(cur().col) + " (near '" + cur().s + "')"};
    }
    void expect(const char *s) { if (t[i].s != s || (t[i].k != Tok::SYM && t[i].k != Tok::ID)) failp(std::string("expected '") + s + "'"); i++; }
    bool ended() const { return t[i].k == Tok::END || t[i].k == Tok::SEP || t[i].col <= jstack.back(); }
    std::string ident() { if (t[i].k != Tok::ID || KEYWORDS.count(t[i].s)) failp("expected an identifier"); return t[i++].s; }

    NodeP expr(int minprec = 0) {
        NodeP lhs = prefix();
        while (!ended()) {
            const Tok &c = cur();
            if (c.k != Tok::SYM) break;
            auto pi = PREC.find(c.s);
            if (pi == PREC.end() || pi->second < minprec) break;
            const int p = pi->second;
            const std::string raw = c.s;
            const int line = c.line;
            i++;
            NodeP rhs = expr(raw == "=>" ? p : p + 1);
            auto ci = CANON.find(raw);
            const std::string op = ci == CANON.end() ? raw : ci->second;
            NodeP n;
            if (op == "/\\" || op == "\\/") { n = node(op == "/\\" ? N_CONJ : N_DISJ, line); n->kids = {lhs, rhs}; }
            else { n = node(N_OP, line); n->s = op; n->kids = {lhs, rhs}; }
            lhs = n;
        }
        return lhs;
    }
    NodeP junction(const std::string &bullet) {
        const int col = cur().col;
        NodeP n = node(bullet == "/\\" ? N_CONJ : N_DISJ, cur().line);
        while (cur().k == Tok::SYM && cur().s == bullet && cur().col == col) {
            i++;
            jstack.push_back(col);
            n->kids.push_back(expr(0));
            jstack.pop_back();
        }
        return n;
    }
    // x \in S | x, y \in S | x \in S, y \in T | <<a, b>> \in S
    std::vector<Bound> bounds() {
        std::vector<Bound> out;
        for (;;) {
            std::vector<Bound> pats;
            for (;;) {
                Bound b;
                if (is_sym("<<")) {
                    i++;
                    b.is_tuple = true;
                    b.names.push_back(ident());
                    while (is_sym(",")) { i++; b.names.push_back(ident()); }
                    expect(">>");
                } else b.names.push_back(ident());
                pats.push_back(b);
                if (is_sym(",")) { i++; continue; }
                break;
            }
            expect("\\in");
            NodeP dom = expr(6);
            for (auto &b : pats) { b.dom = dom; out.push_back(b); }
            if (is_sym(",")) { i++; continue; }
            return out;
        }
    }
    std::vector<Def> definitions(bool stop_in) {
        std::vector<Def> defs;
        for (;;) {
            if (stop_in && is_id("IN")) return defs;
            if (is_id("RECURSIVE")) {
                i++;
                for (;;) {
                    ident();
                    if (is_sym("(")) { while (!is_sym(")")) i++; i++; }
                    if (is_sym(",")) { i++; continue; }
                    break;
                }
                continue;
            }
            if (cur().k != Tok::ID || KEYWORDS.count(cur().s)) return defs;
            defs.push_back(definition());
        }
    }
    Def definition() {
        Def d;
        d.line = cur().line;
        d.name = ident();
        if (is_sym("(")) {
            i++;
            for (;;) {
                std::string p = ident();
                int arity = 0;
                if (is_sym("(")) { i++; while (!is_sym(")")) { if (is_id("_")) arity++; i++; } i++; }
                d.params.emplace_back(p, arity);
                if (is_sym(",")) { i++; continue; }
                break;
            }
            expect(")");
            expect("==");
            body_with_span(d);
        } else if (is_sym("[")) {  // f[x \in S] == e  is  f == [x \in S |-> e]
            i++;
            NodeP n = node(N_FNDEF, d.line);
            n->bounds = bounds();
            expect("]");
            expect("==");
            n->kids = {expr(0)};
            n->s = d.name;
            d.body = n;
        } else {
            expect("==");
            if (is_id("INSTANCE")) {
                i++;
                NodeP n = node(N_INSTANCE, d.line);
                n->s = ident();
                if (is_id("WITH")) {
                    i++;
                    for (;;) {
                        std::string a = ident();
                        expect("<-");
                        n->fields.emplace_back(a, expr(0));
                        if (is_sym(",")) { i++; continue; }
                        break;
                    }
                }
                d.body = n;
            } else body_with_span(d);
        }
        return d;
    }
    void body_with_span(Def &d) {
        const size_t first = i;
        d.body = expr(0);
        const Tok &a = t[first], &z = t[i - 1];
        d.l1 = a.line; d.c1 = a.col; d.l2 = z.line; d.c2 = z.col + (int)z.s.size() - 1 + (z.k == Tok::STR ? 2 : 0);
    }
    NodeP prefix() {
        const Tok c = cur();
        if (c.k == Tok::SYM) {
            if (c.s == "/\\" || c.s == "\\/") return junction(c.s);
            if (c.s == "~" || c.s == "\\lnot" || c.s == "\\neg") { i++; NodeP n = node(N_NOT, c.line); n->kids = {expr(4)}; return n; }
            if (c.s == "-") { i++; NodeP n = node(N_NEG, c.line); n->kids = {expr(12)}; return n; }
            if (c.s == "\\A" || c.s == "\\E") {
                i++;
                NodeP n = node(N_QUANT, c.line);
                n->s = c.s.substr(1);
                n->bounds = bounds();
                expect(":");
                n->kids = {expr(0)};
                return n;
            }
            if (c.s == "[]" || c.s == "<>") { i++; NodeP n = node(N_TEMPORAL, c.line); n->s = c.s; n->kids = {expr(4)}; return n; }
        }
        if (c.k == Tok::ID) {
            if ((c.s == "WF_" || c.s == "SF_") && t[i + 1].s == "<<") {  // WF_<<v1, v2>>(A): fairness, parsed and never evaluated
                i++;
                NodeP sub = expr(16);
                expect("(");
                NodeP act = expr(0);
                expect(")");
                NodeP n = node(N_TEMPORAL, c.line); n->s = c.s; n->kids = {sub, act}; return n;
            }
            if (c.s.size() > 3 && (c.s.compare(0, 3, "WF_") == 0 || c.s.compare(0, 3, "SF_") == 0) && t[i + 1].s == "(") {  // WF_vars(A)
                i += 2;
                NodeP act = expr(0);
                expect(")");
                NodeP n = node(N_TEMPORAL, c.line); n->s = c.s.substr(0, 3); n->kids = {node_id(c.s.substr(3)), act}; return n;
            }
            if (c.s == "CHOOSE") {
                i++;
                if (cur().k == Tok::ID && t[i + 1].s == ":") {  // unbounded CHOOSE x : P — the cfg replaces the symbol by a model value
                    NodeP n = node(N_CHOOSE_UNB, c.line);
                    n->s = ident();
                    expect(":");
                    n->kids = {expr(0)};
                    return n;
                }
                NodeP n = node(N_CHOOSE, c.line);
                n->bounds = bounds();
                expect(":");
                n->kids = {expr(0)};
                return n;
            }
            if (c.s == "IF") {
                i++;
                NodeP n = node(N_IF, c.line);
                NodeP cond = expr(0);
                expect("THEN");
                NodeP a = expr(0);
                expect("ELSE");
                n->kids = {cond, a, expr(0)};
                return n;
            }
            if (c.s == "CASE") {
                i++;
                NodeP n = node(N_CASE, c.line);
                NodeP other;
                for (;;) {
                    if (is_id("OTHER")) { i++; expect("->"); other = expr(0); }
                    else { NodeP g = expr(0); expect("->"); n->kids.push_back(g); n->kids.push_back(expr(0)); }
                    if (is_sym("[]") && !ended()) { i++; continue; }
                    break;
                }
                if (other) { n->num = 1; n->kids.push_back(other); }
                return n;
            }
            if (c.s == "LET") {
                i++;
                NodeP n = node(N_LET, c.line);
                n->defs = definitions(true);
                expect("IN");
                n->kids = {expr(0)};
                return n;
            }
            if (c.s == "LAMBDA") {
                i++;
                NodeP n = node(N_LAMBDA, c.line);
                n->names.push_back(ident());
                while (is_sym(",")) { i++; n->names.push_back(ident()); }
                expect(":");
                n->kids = {expr(0)};
                return n;
            }
            if (c.s == "UNCHANGED") { i++; NodeP n = node(N_UNCHANGED, c.line); n->kids = {expr(14)}; return n; }
            if (c.s == "ENABLED") { i++; NodeP n = node(N_ENABLED, c.line); n->kids = {expr(14)}; return n; }
            if (c.s == "SUBSET" || c.s == "UNION" || c.s == "DOMAIN") { i++; NodeP n = node(N_PRE, c.line); n->s = c.s; n->kids = {expr(9)}; return n; }
        }
        return postfix(atom());
    }
    std::vector<NodeP> exprlist(const char *close) {
        std::vector<NodeP> items;
        if (!is_sym(close)) {
            items.push_back(expr(0));
            while (is_sym(",")) { i++; items.push_back(expr(0)); }
        }
        expect(close);
        return items;
    }
    NodeP tuple_of(std::vector<NodeP> idx, int line) { if (idx.size() == 1) return idx[0]; NodeP n = node(N_TUPLE, line); n->kids = idx; return n; }
    NodeP atom() {
        const Tok c = cur();
        i++;
        if (c.k == Tok::NUM) { NodeP n = node(N_NUM, c.line); n->num = atol(c.s.c_str()); return n; }
        if (c.k == Tok::STR) { NodeP n = node(N_STR, c.line); n->s = c.s; return n; }
        if (c.k == Tok::ID && !KEYWORDS.count(c.s)) {
            if (c.s == "TRUE" || c.s == "FALSE") { NodeP n = node(N_BOOL, c.line); n->num = c.s == "TRUE"; return n; }
            std::string name = c.s;
            while (is_sym("!") && peek().k == Tok::ID) { i++; name += "!" + ident(); }  // Inst!Op
            if (is_sym("!") && peek().k == Tok::SYM && peek().s == ":") { i += 2; NodeP n = node(N_ID, c.line); n->s = name; return n; }  // Thm!:
            if (is_sym("!") && peek().k == Tok::NUM) { i++; NodeP n = node(N_NTH, c.line); n->s = name; n->num = atol(cur().s.c_str()); i++; return n; }
            if (is_sym("(") && !ended()) { i++; NodeP n = node(N_CALL, c.line); n->s = name; n->kids = exprlist(")"); return n; }
            NodeP n = node(N_ID, c.line); n->s = name; return n;
        }
        if (c.k == Tok::SYM) {
            if (c.s == "(") { NodeP e = expr(0); expect(")"); NodeP n = node(N_PAREN, c.line); n->kids = {e}; return n; }
            if (c.s == "@") return node(N_AT, c.line);
            if (c.s == "{") {
                if (is_sym("}")) { i++; return node(N_SETENUM, c.line); }
                const size_t save = i;
                if ((cur().k == Tok::ID && peek().k == Tok::SYM && peek().s == "\\in") || is_sym("<<")) {  // {x \in S : P}
                    try {
                        auto bs = bounds();
                        if (bs.size() == 1 && is_sym(":")) {
                            i++;
                            NodeP n = node(N_SETFILTER, c.line);
                            n->bounds = bs;
                            n->kids = {expr(0)};
                            expect("}");
                            return n;
                        }
                    } catch (SyntaxErr &) {}
                    i = save;
                }
                NodeP first = expr(0);
                if (is_sym(":")) {  // {e : x \in S, y \in T}
                    i++;
                    NodeP n = node(N_SETMAP, c.line);
                    n->kids = {first};
                    n->bounds = bounds();
                    expect("}");
                    return n;
                }
                NodeP n = node(N_SETENUM, c.line);
                n->kids.push_back(first);
                while (is_sym(",")) { i++; n->kids.push_back(expr(0)); }
                expect("}");
                return n;
            }
            if (c.s == "<<") {
                auto items = exprlist(">>");
                if (cur().k == Tok::ID && cur().s[0] == '_' && !ended()) {  // <<A>>_v  =  A /\ (v' # v)
                    const bool bare = cur().s == "_";
                    NodeP sub = node_id(cur().s.substr(1));
                    i++;
                    if (bare) sub = expr(16);
                    NodeP un = node(N_UNCHANGED, c.line); un->kids = {sub};
                    NodeP act = items.size() == 1 ? items[0] : nullptr;
                    if (!act) failp("<<A>>_v takes one action");
                    NodeP n = node(N_TEMPORAL, c.line); n->s = "<<>>_"; n->kids = {act, sub, un}; return n;
                }
                NodeP n = node(N_TUPLE, c.line); n->kids = items; return n;
            }
            if (c.s == "[") {
                const Tok n1 = peek();
                if (cur().k == Tok::ID && n1.k == Tok::SYM && (n1.s == "|->" || n1.s == ":") && !KEYWORDS.count(cur().s)) {
                    NodeP n = node(n1.s == "|->" ? N_RECORD : N_RECORDSET, c.line);
                    for (;;) {
                        std::string f = ident();
                        expect(n1.s.c_str());
                        n->fields.emplace_back(f, expr(0));
                        if (is_sym(",")) { i++; continue; }
                        break;
                    }
                    expect("]");
                    return n;
                }
                const size_t save = i;
                if ((cur().k == Tok::ID && n1.k == Tok::SYM && (n1.s == "\\in" || n1.s == ",")) || is_sym("<<")) {  // [x \in S |-> e]
                    try {
                        auto bs = bounds();
                        if (is_sym("|->")) {
                            i++;
                            NodeP n = node(N_FNDEF, c.line);
                            n->bounds = bs;
                            n->kids = {expr(0)};
                            expect("]");
                            return n;
                        }
                    } catch (SyntaxErr &) {}
                    i = save;
                }
                NodeP first = expr(0);
                if (is_id("EXCEPT")) {
                    i++;
                    NodeP n = node(N_EXCEPT, c.line);
                    n->kids = {first};
                    for (;;) {
                        expect("!");
                        std::vector<NodeP> path;
                        for (;;) {
                            if (is_sym("[")) { i++; path.push_back(tuple_of(exprlist("]"), c.line)); }
                            else if (is_sym(".")) { i++; NodeP f = node(N_STR, c.line); f->s = ident(); path.push_back(f); }
                            else break;
                        }
                        expect("=");
                        n->ups.emplace_back(path, expr(0));
                        if (is_sym(",")) { i++; continue; }
                        break;
                    }
                    expect("]");
                    return n;
                }
                if (is_sym("->")) { i++; NodeP n = node(N_FNSET, c.line); n->kids = {first, expr(0)}; expect("]"); return n; }
                expect("]");
                if (cur().k == Tok::ID && cur().s[0] == '_') {  // [A]_v
                    const bool bare = cur().s == "_";
                    NodeP sub = node_id(cur().s.substr(1));
                    i++;
                    if (bare) sub = expr(16);
                    NodeP un = node(N_UNCHANGED, c.line); un->kids = {sub};  // [A]_v  =  A \/ UNCHANGED v
                    NodeP n = node(N_TEMPORAL, c.line); n->s = "[]_"; n->kids = {first, sub, un}; return n;
                }
                failp("unsupported bracket expression");
            }
        }
        i--;
        failp("expected an expression");
    }
    NodeP postfix(NodeP e) {
        while (!ended()) {
            if (is_sym("[")) { const int line = cur().line; i++; NodeP n = node(N_IDX, line); n->kids = {e, tuple_of(exprlist("]"), line)}; e = n; }
            else if (is_sym(".")) { const int line = cur().line; i++; NodeP f = node(N_STR, line); f->s = ident(); NodeP n = node(N_IDX, line); n->kids = {e, f}; e = n; }
            else if (is_sym("'")) { const int line = cur().line; i++; NodeP n = node(N_PRIME, line); n->kids = {e}; e = n; }
            else break;
        }
        return e;
    }
};

// parsed MODULE text: name, extends, constants (name -> arity), variables (ordered), definitions
struct Module {
    std::string name;
    std::vector<std::string> extends, variables, def_order, instances;
    std::vector<std::pair<std::string, int>> assumes;  // (definition name, line) of every ASSUME / ASSUMPTION / AXIOM, in order
    std::map<std::string, int> constants;
    std::map<std::string, Def> defs;
    explicit Module(const std::string &text_in) {
        // structured proofs (examples/Paxos/Voting.tla:187-196) are not evaluated: their lines are blanked
        std::string text;
        {
            std::istringstream in(text_in);
            std::string ln;
            while (std::getline(in, ln)) {
                size_t p = ln.find_first_not_of(" \t");
                bool proof = false;
                if (p != std::string::npos) {
                    const std::string r = ln.substr(p);
                    auto kw = [&](const char *w) { const size_t L = strlen(w); return r.compare(0, L, w) == 0 && (r.size() == L || !(isalnum((unsigned char)r[L]) || r[L] == '_')); };
                    if (r[0] == '<' && r.size() > 2 && isdigit((unsigned char)r[1])) { size_t q = 1; while (q < r.size() && isdigit((unsigned char)r[q])) q++; proof = q < r.size() && r[q] == '>'; }
                    proof = proof || kw("BY") || kw("QED") || kw("OBVIOUS") || kw("OMITTED") || kw("PROOF");
                }
                text += proof ? "" : ln;
                text += "\n";
            }
        }
        Parser p;
        p.t = lex(text);
        while (!(p.cur().k == Tok::ID && p.cur().s == "MODULE")) { if (p.cur().k == Tok::END) throw SyntaxErr{"no MODULE header"}; p.i++; }
        p.i++;
        name = p.cur().s;
        p.i++;
        if (p.cur().k == Tok::SEP) p.i++;
        for (;;) {
            const Tok c = p.cur();
            if (c.k == Tok::END || (c.k == Tok::SEP && c.s[0] == '=')) break;
            if (c.k == Tok::SEP) { p.i++; continue; }
            if (c.k != Tok::ID) p.failp("expected a module unit");
            if (c.s == "LOCAL") { p.i++; continue; }
            if (c.s == "EXTENDS") {
                p.i++;
                extends.push_back(p.ident());
                while (p.is_sym(",")) { p.i++; extends.push_back(p.ident()); }
            } else if (c.s == "CONSTANT" || c.s == "CONSTANTS") {
                p.i++;
                for (;;) {
                    std::string nm = p.ident();
                    int arity = 0;
                    if (p.is_sym("(")) { p.i++; while (!p.is_sym(")")) { if (p.is_id("_")) arity++; p.i++; } p.i++; }
                    constants[nm] = arity;
                    if (p.is_sym(",")) { p.i++; continue; }
                    break;
                }
            } else if (c.s == "VARIABLE" || c.s == "VARIABLES") {
                p.i++;
                variables.push_back(p.ident());
                while (p.is_sym(",")) { p.i++; variables.push_back(p.ident()); }
            } else if (c.s == "ASSUME" || c.s == "ASSUMPTION" || c.s == "AXIOM" || c.s == "THEOREM" || c.s == "LEMMA" || c.s == "PROPOSITION" || c.s == "COROLLARY") {
                const bool assumption = c.s == "ASSUME" || c.s == "ASSUMPTION" || c.s == "AXIOM";
                p.i++;
                std::string nm;
                if (p.cur().k == Tok::ID && p.peek().k == Tok::SYM && p.peek().s == "==") { nm = p.cur().s; p.i += 2; }
                NodeP e = p.expr(0);  // evaluated in TLC's "No Behavior Spec" mode only (Checker::run), or when a model refers to it as Name
                // (an unnamed assumption gets a name nobody can type, so that it is loaded like every other definition)
                if (nm.empty() && assumption) nm = "ASSUME@" + name + ":" + std::to_string(c.line);   // (with the module: definitions of EXTENDed modules are merged by name — ADVICE round 5)
                if (!nm.empty()) { Def d; d.name = nm; d.body = e; d.line = c.line; defs[nm] = d; def_order.push_back(nm); }
                if (assumption) assumes.emplace_back(nm, c.line);
                while (p.cur().k == Tok::ID && (p.cur().s == "PROOF" || p.cur().s == "BY" || p.cur().s == "OBVIOUS" || p.cur().s == "OMITTED" || p.cur().s == "QED")) p.i++;
            } else if (c.s == "INSTANCE") {
                p.i++;
                instances.push_back(p.ident());  // its definitions join this module's (those of a standard module are built in)
                if (p.is_id("WITH")) p.failp("unnamed INSTANCE ... WITH is not supported");
            } else {
                const size_t before = p.i;
                for (auto &d : p.definitions(false)) { defs[d.name] = d; def_order.push_back(d.name); }
                if (p.i == before) p.failp("cannot parse module unit");
            }
        }
    }
};

// =============================================================================================== evaluator: environments
std::unordered_map<std::string, int> g_symtab;
std::vector<std::string> g_symnames;
int intern(const std::string &s) {
    auto it = g_symtab.find(s);
    if (it != g_symtab.end()) return it->second;
    const int id = (int)g_symnames.size();
    g_symtab.emplace(s, id);
    g_symnames.push_back(s);
    return id;
}
inline int nsym(const Node *n) { if (n->sym < 0) n->sym = intern(n->s); return n->sym; }
const std::vector<int> &bsyms(const Bound &b) {
    if (b.syms.size() != b.names.size()) { b.syms.clear(); for (auto &s : b.names) b.syms.push_back(intern(s)); }
    return b.syms;
}
const std::vector<int> &dpsyms(const Def &d) {
    if (d.psyms.size() != d.params.size()) { d.psyms.clear(); for (auto &p : d.params) d.psyms.push_back(intern(p.first)); }
    return d.psyms;
}

typedef std::vector<V> State;  // one value per VARIABLE, in declaration order; a null entry = not assigned yet

// Environment entries, thunks and operator values live until the current source state has been expanded (they refer to each
// other cyclically: a LET definition sees itself and its siblings); values (V) are the only things that outlive the arena.
// (pooled: an expansion creates tens of millions of environment entries; they are constructed in place in blocks that are
//  reused from one source state to the next)
template <class T>
struct Pool {
    static constexpr size_t B = 4096;
    std::vector<T *> blocks;
    size_t used = 0;
    T *get() {
        if (used == blocks.size() * B) blocks.push_back(static_cast<T *>(malloc(sizeof(T) * B)));
        T *p = &blocks[used / B][used % B];
        used++;
        return new (p) T();
    }
    void reset() { for (size_t i = 0; i < used; i++) blocks[i / B][i % B].~T(); used = 0; }
    ~Pool() { reset(); for (T *b : blocks) free(b); }
};
template <class T> Pool<T> &pool() { static Pool<T> p; return p; }
template <class T> T *anew() { return pool<T>().get(); }
void arena_reset();
struct Thunk;
struct OpVal;
struct RecFn;
struct Env { int sym = -1; V val; Thunk *th = nullptr; OpVal *op = nullptr; RecFn *rec = nullptr; Env *next = nullptr; };
// a lazily evaluated operator argument / LET definition without parameters: memoised unless it looks at primed variables
struct Thunk { const Node *n = nullptr; Env *env = nullptr; const State *st = nullptr; bool memo = true, done = false, is_def = false; V val; };
// an operator as a value: LAMBDA, an operator passed by name, a LET operator with parameters
struct OpVal { std::vector<int> params; const Node *body = nullptr; Env *env = nullptr; bool primed = false, is_let = false; int builtin = -1; std::string name; };

// f[x \in S] == e with f inside e (WriteThroughCache.tla:55-60): while the function is being built, f[a] evaluates e for x = a
struct RecFn { const Node *n = nullptr; Env *env = nullptr; const State *st = nullptr; V dom; std::vector<std::pair<V, V>> cache; };
void arena_reset() { pool<Env>().reset(); pool<Thunk>().reset(); pool<OpVal>().reset(); pool<RecFn>().reset(); }
inline Env *env_find(Env *e, int sym) { for (; e; e = e->next) if (e->sym == sym) return e; return nullptr; }
inline Env *bind_val(int sym, const V &v, Env *next) { Env *e = anew<Env>(); e->sym = sym; e->val = v; e->next = next; return e; }
inline bool entry_primed(const Env *e);

struct GDef {  // a module-level definition
    std::string name, module;
    int sym = -1, line = 0, l1 = 0, c1 = 0, l2 = 0, c2 = 0;
    std::vector<std::pair<int, int>> params;  // (symbol, arity)
    NodeP body;
    int primed = -1, is_const = -1;
    V const_val;
    // a state-level definition without parameters is a function of the variables it (transitively) mentions: its last few
    // values are kept by the identity of those variables' values (totalOpOrder of InnerSerial.tla:89-95 depends on opQ only and
    // is asked for once per candidate opOrder')
    int deps_known = 0, memo_next = 0;
    std::vector<int> deps;
    struct Memo { std::vector<V> key; V val; } memo[4];
};

// which identifiers decide whether evaluating a node looks at primed variables (decides whether a thunk may be memoised, and
// whether a LET definition used as an action conjunct is expanded as an action)
struct PInfo { bool stat = false; std::vector<std::pair<int, bool>> free; };  // free identifier -> primed when it means the global definition
inline bool entry_primed(const Env *e) { return e->th ? !e->th->memo : e->op ? e->op->primed : false; }

enum { OP_EQ, OP_NE, OP_IN, OP_NOTIN, OP_IMP, OP_EQUIV, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MOD, OP_POW, OP_LT, OP_GT, OP_LE, OP_GE, OP_RANGE,
       OP_CUP, OP_CAP, OP_SETMINUS, OP_SUBSETEQ, OP_CONCAT, OP_MAPSTO, OP_ATAT, OP_TIMES, OP_PSUBSET, OP_UNSUPPORTED };
const std::map<std::string, int> OPCODES = {{"=", OP_EQ}, {"#", OP_NE}, {"\\in", OP_IN}, {"\\notin", OP_NOTIN}, {"=>", OP_IMP}, {"<=>", OP_EQUIV},
    {"+", OP_ADD}, {"-", OP_SUB}, {"*", OP_MUL}, {"\\div", OP_DIV}, {"%", OP_MOD}, {"^", OP_POW}, {"<", OP_LT}, {">", OP_GT}, {"<=", OP_LE}, {">=", OP_GE},
    {"..", OP_RANGE}, {"\\cup", OP_CUP}, {"\\cap", OP_CAP}, {"\\", OP_SETMINUS}, {"\\subseteq", OP_SUBSETEQ}, {"\\o", OP_CONCAT}, {":>", OP_MAPSTO},
    {"@@", OP_ATAT}, {"\\X", OP_TIMES}, {"\\subset", OP_PSUBSET}};
std::vector<std::string> *g_print_sink = nullptr;
enum { B_CARD, B_ISFINITE, B_LEN, B_APPEND, B_HEAD, B_TAIL, B_SUBSEQ, B_SEQ, B_SELECTSEQ, B_ASSERT, B_PERMUTATIONS, B_PRINT, B_PRINTT, B_TOSTRING };
const std::map<std::string, std::pair<int, int>> BUILTIN_OPS = {{"Cardinality", {B_CARD, 1}}, {"IsFiniteSet", {B_ISFINITE, 1}}, {"Len", {B_LEN, 1}},
    {"Append", {B_APPEND, 2}}, {"Head", {B_HEAD, 1}}, {"Tail", {B_TAIL, 1}}, {"SubSeq", {B_SUBSEQ, 3}}, {"Seq", {B_SEQ, 1}}, {"SelectSeq", {B_SELECTSEQ, 2}},
    {"Assert", {B_ASSERT, 2}}, {"Permutations", {B_PERMUTATIONS, 1}}, {"Print", {B_PRINT, 2}}, {"PrintT", {B_PRINTT, 1}}, {"ToString", {B_TOSTRING, 1}}};
const std::set<std::string> BUILTIN_MODULES = {"Naturals", "Integers", "Reals", "FiniteSets", "Sequences", "TLC", "Bags", "RealTime", "TLAPS"};

long as_int(const V &v, const char *what) { if (v->k != K_INT) fail(std::string(what) + " applied to the non-integer " + fmt(v)); return v->i; }
const std::vector<V> &as_seq(const V &v, const char *what) { if (v->k != K_TUPLE) fail(std::string(what) + " applied to the non-sequence " + fmt(v)); return v->items; }
bool as_bool(const V &v, const char *what) { if (v->k != K_BOOL) fail(std::string(what) + " the non-boolean " + fmt(v)); return v->i != 0; }
bool veq(const V &a, const V &b) {
    if (a->k == K_LAZY || b->k == K_LAZY) return eq(to_set(a), to_set(b));
    return eq(a, b);
}

bool read_text(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

// a copy of a syntax tree in which identifiers are rewritten (INSTANCE import, Init as an action)
NodeP clone_tree(const NodeP &x, const std::function<NodeP(const Node &)> &leaf) {
    if (!x) return x;
    if (NodeP r = leaf(*x)) return r;
    auto n = std::make_shared<Node>(*x);
    n->sym = n->opc = -1; n->uv_state = 0; n->uvars.clear(); n->psyms.clear(); n->pinfo.reset();
    for (auto &k : n->kids) k = clone_tree(k, leaf);
    for (auto &b : n->bounds) { b.dom = clone_tree(b.dom, leaf); b.syms.clear(); }
    for (auto &d : n->defs) { d.body = clone_tree(d.body, leaf); d.psyms.clear(); d.sym = -1; }
    for (auto &u : n->ups) { for (auto &p : u.first) p = clone_tree(p, leaf); u.second = clone_tree(u.second, leaf); }
    for (auto &f : n->fields) f.second = clone_tree(f.second, leaf);
    return n;
}
// every child expression of a node, for the generic tree walks
template <class F> void each_child(const Node &n, F f) {
    for (auto &k : n.kids) if (k) f(*k);
    for (auto &b : n.bounds) if (b.dom) f(*b.dom);
    for (auto &d : n.defs) if (d.body) f(*d.body);
    for (auto &u : n.ups) { for (auto &p : u.first) f(*p); f(*u.second); }
    for (auto &fl : n.fields) if (fl.second) f(*fl.second);
}

using Cont = std::function<void()>;

// a root module + everything it EXTENDS, constants bound by a cfg: evaluates expressions (ev) and actions (act)
struct Spec {
    std::vector<std::string> search, loaded, variables;
    std::unordered_map<int, int> varidx, overrides, scoped_overrides;
    std::map<std::string, int> constants;
    std::unordered_map<int, V> const_vals;
    std::unordered_map<int, GDef *> defs;
    std::vector<std::unique_ptr<GDef>> def_store;
    std::map<std::pair<std::string, std::string>, std::string> scoped;  // (module, name) -> name: the cfg's `Id <-[Module] Id`
    std::unordered_map<int, V> builtin_consts;
    std::unordered_map<int, std::pair<int, int>> builtin_ops;
    int sym_at = -1;

    Spec() {
        sym_at = intern("@");
        builtin_consts[intern("Nat")] = mk_lazy(L_NAT);
        builtin_consts[intern("Int")] = mk_lazy(L_INT);
        builtin_consts[intern("STRING")] = mk_lazy(L_STRING);
        builtin_consts[intern("BOOLEAN")] = mk_set({g_false, g_true});
        for (auto &b : BUILTIN_OPS) builtin_ops[intern(b.first)] = b.second;
    }
    std::string find_module(const std::string &name) const {
        for (auto &d : search) { std::string p = d + "/" + name + ".tla"; std::ifstream f(p); if (f) return p; }
        std::string sp;
        for (auto &d : search) sp += (sp.empty() ? "" : ", ") + d;
        fail("module " + name + " not found (search path: " + sp + ")");
    }
    GDef *add_def(const std::string &name, const Def &d, const NodeP &body, const std::string &module) {
        auto g = std::make_unique<GDef>();
        g->name = name; g->module = module; g->sym = intern(name); g->line = d.line; g->body = body;
        g->l1 = d.l1; g->c1 = d.c1; g->l2 = d.l2; g->c2 = d.c2;
        for (auto &p : d.params) g->params.emplace_back(intern(p.first), p.second);
        GDef *r = g.get();
        def_store.push_back(std::move(g));
        defs[r->sym] = r;
        return r;
    }
    struct Assume { std::string def; int line; std::string module; };
    std::vector<Assume> assumes;  // of the root module and of what it EXTENDS, in load order (TLC checks them all)
    void load(const std::string &path) {
        std::string text;
        if (!read_text(path, text)) fail("cannot read " + path);
        Module m(text);
        for (auto &e : m.extends) {
            if (BUILTIN_MODULES.count(e) || std::find(loaded.begin(), loaded.end(), e) != loaded.end()) continue;
            load(find_module(e));
        }
        loaded.push_back(m.name);
        for (auto &a : m.assumes) assumes.push_back({a.first, a.second, m.name});
        for (auto &c : m.constants) constants[c.first] = c.second;
        for (auto &v : m.variables) if (std::find(variables.begin(), variables.end(), v) == variables.end()) variables.push_back(v);
        for (auto &nm : m.def_order) {
            const Def &d = m.defs.at(nm);
            if (d.body->k == N_INSTANCE) import_instance(nm + "!", d.body->s, d.body->fields);
            else add_def(nm, d, d.body, m.name);
        }
        for (auto &i : m.instances)  // INSTANCE M without a name (TLC/MCAlternatingBit.tla:4): M's definitions under their own names
            if (!BUILTIN_MODULES.count(i) && std::find(loaded.begin(), loaded.end(), i) == loaded.end()) import_instance("", i, {});
    }
    // I == INSTANCE M WITH c <- e, ...: every definition d of M (and of what M EXTENDS) becomes the global definition I!d, in
    // which M's own definition names are prefixed and the substituted constants / variables are replaced by their expressions; a
    // constant or variable of M without a WITH clause stands for the instantiating module's identifier of the same name
    // (examples/Paxos/Voting.tla:185 `C == INSTANCE Consensus`).  Priming such an identifier is priming a state function.
    void import_instance(const std::string &prefix, const std::string &modname, const std::vector<std::pair<std::string, NodeP>> &subst) {
        std::vector<std::unique_ptr<Module>> mods;
        std::function<void(const std::string &)> gather = [&](const std::string &nm) {
            std::string text;
            const std::string p = find_module(nm);
            if (!read_text(p, text)) fail("cannot read " + p);
            auto m = std::make_unique<Module>(text);
            for (auto &e : m->extends) if (!BUILTIN_MODULES.count(e)) gather(e);
            mods.push_back(std::move(m));
        };
        gather(modname);
        std::set<std::string> names;
        for (auto &m : mods) for (auto &d : m->defs) names.insert(d.first);
        std::map<std::string, NodeP> sub(subst.begin(), subst.end());
        std::function<NodeP(const Node &)> leaf = [&](const Node &x) -> NodeP {
            if (x.k != N_ID && x.k != N_CALL && x.k != N_NTH) return nullptr;
            if (x.k == N_ID) { auto it = sub.find(x.s); if (it != sub.end()) return it->second; }
            const std::string head = x.s.substr(0, x.s.find('!'));
            if (!names.count(head)) return nullptr;
            auto n = std::make_shared<Node>(x);
            n->s = prefix + x.s;
            n->sym = -1; n->pinfo.reset();
            for (auto &k : n->kids) k = clone_tree(k, leaf);
            return n;
        };
        for (auto &m : mods)
            for (auto &nm : m->def_order) {
                const Def &d = m->defs.at(nm);
                if (d.body->k == N_INSTANCE) {
                    std::vector<std::pair<std::string, NodeP>> s2;
                    for (auto &f : d.body->fields) s2.emplace_back(f.first, clone_tree(f.second, leaf));
                    import_instance(prefix + nm + "!", d.body->s, s2);
                    continue;
                }
                if (prefix.empty() && defs.count(intern(nm))) continue;  // already there through EXTENDS
                add_def(prefix + nm, d, clone_tree(d.body, leaf), m->name);
                auto sc = scoped.find({m->name, nm});
                if (sc != scoped.end()) scoped_overrides[intern(prefix + nm)] = intern(sc->second);
            }
    }
    void finish_load() {
        for (size_t i = 0; i < variables.size(); i++) varidx[intern(variables[i])] = (int)i;
        for (auto &s : scoped_overrides) overrides[s.first] = s.second;
        for (auto &c : constants) {
            const int s = intern(c.first);
            if (!const_vals.count(s) && !overrides.count(s)) fail("CONSTANT " + c.first + " has no value in the configuration");
        }
    }
    int resolve(int sym) const { auto it = overrides.find(sym); return it == overrides.end() ? sym : it->second; }

    // ------------------------------------------------------------------ static analysis
    const Node *nth_node(const Node *n) {
        const int name = resolve(nsym(n));
        auto it = defs.find(name);
        if (it == defs.end()) fail(n->s + "!" + std::to_string(n->num) + ": " + n->s + " is not a definition");
        const Node *b = it->second->body.get();
        while (b->k == N_PAREN) b = b->kids[0].get();
        if ((b->k != N_CONJ && b->k != N_DISJ) || n->num < 1 || n->num > (long)b->kids.size()) fail(n->s + "!" + std::to_string(n->num) + ": the definition has no such conjunct");
        return b->kids[(size_t)n->num - 1].get();
    }
    typedef std::vector<std::pair<int, std::shared_ptr<PInfo>>> PScope;  // local name -> what using it contributes (null: nothing)
    static void pmerge(PInfo &into, const PInfo &x) { into.stat = into.stat || x.stat; into.free.insert(into.free.end(), x.free.begin(), x.free.end()); }
    bool def_primed(GDef *d) {
        if (d->primed < 0) {
            d->primed = 0;  // recursion guard
            PScope sc;
            for (auto &p : d->params) sc.emplace_back(p.first, nullptr);
            PInfo r = pwalk(d->body.get(), sc);
            bool p = r.stat;
            for (auto &f : r.free) p = p || f.second;
            d->primed = p;
        }
        return d->primed != 0;
    }
    PInfo pwalk(const Node *n, PScope &sc) {
        PInfo out;
        auto push_names = [&](const std::vector<int> &syms) { for (int s : syms) sc.emplace_back(s, nullptr); };
        switch (n->k) {
            case N_PRIME: case N_UNCHANGED: case N_ENABLED: out.stat = true; return out;
            case N_NUM: case N_STR: case N_BOOL: case N_AT: return out;
            case N_NTH: { PScope e; return pwalk(nth_node(n), e); }
            case N_ID: case N_CALL: {
                const int s = nsym(n);
                bool local = false;
                for (size_t i = sc.size(); i-- > 0;) if (sc[i].first == s) { local = true; if (sc[i].second) pmerge(out, *sc[i].second); break; }
                if (!local) {
                    bool gp = false;
                    auto it = defs.find(s);
                    if (it != defs.end() && !overrides.count(s)) gp = def_primed(it->second);
                    out.free.emplace_back(s, gp);
                }
                if (n->k == N_CALL) for (auto &a : n->kids) pmerge(out, pwalk(a.get(), sc));
                return out;
            }
            case N_LET: {
                const size_t mark = sc.size();
                for (auto &d : n->defs) {
                    const size_t m2 = sc.size();
                    sc.emplace_back(intern(d.name), nullptr);  // visible to itself, not primed while its own body is analysed
                    const size_t self = sc.size() - 1;
                    push_names(dpsyms(d));
                    auto info = std::make_shared<PInfo>(pwalk(d.body.get(), sc));
                    sc.resize(m2 + 1);
                    sc[self].second = info;  // a primed definition matters only where it is used
                }
                out = pwalk(n->kids[0].get(), sc);
                sc.resize(mark);
                return out;
            }
            case N_QUANT: case N_CHOOSE: case N_SETFILTER: case N_SETMAP: case N_FNDEF: {
                for (auto &b : n->bounds) pmerge(out, pwalk(b.dom.get(), sc));
                const size_t mark = sc.size();
                for (auto &b : n->bounds) push_names(bsyms(b));
                pmerge(out, pwalk(n->kids[0].get(), sc));
                sc.resize(mark);
                return out;
            }
            case N_LAMBDA: {
                const size_t mark = sc.size();
                for (auto &p : n->names) sc.emplace_back(intern(p), nullptr);
                out = pwalk(n->kids[0].get(), sc);
                sc.resize(mark);
                return out;
            }
            default:
                each_child(*n, [&](const Node &c) { pmerge(out, pwalk(&c, sc)); });
                return out;
        }
    }
    bool primed_rt(const Node *n, Env *env) {
        if (!n->pinfo) { PScope sc; n->pinfo = std::make_shared<PInfo>(pwalk(n, sc)); }
        if (n->pinfo->stat) return true;
        for (auto &f : n->pinfo->free) {
            const Env *e = env_find(env, f.first);
            if (e ? entry_primed(e) : f.second) return true;
        }
        return false;
    }
    bool mentions_state(const Node *n, std::set<int> &seen) {  // does the expression (transitively) mention a VARIABLE?
        if (n->k == N_NTH) return mentions_state(nth_node(n), seen);
        if (n->k == N_PRIME || n->k == N_UNCHANGED || n->k == N_ENABLED) return true;
        if (n->k == N_ID || n->k == N_CALL) {
            const int nm = resolve(nsym(n));
            if (varidx.count(nm)) return true;
            auto it = defs.find(nm);
            if (it != defs.end() && !seen.count(nm)) { seen.insert(nm); if (mentions_state(it->second->body.get(), seen)) return true; }
        }
        bool r = false;
        each_child(*n, [&](const Node &c) { r = r || mentions_state(&c, seen); });
        return r;
    }
    void state_deps(const Node *n, std::set<int> &seen, std::set<int> &vars) {  // the variables an expression (transitively) mentions
        if (n->k == N_NTH) { state_deps(nth_node(n), seen, vars); return; }
        if (n->k == N_ID || n->k == N_CALL) {
            const int nm = resolve(nsym(n));
            auto v = varidx.find(nm);
            if (v != varidx.end()) vars.insert(v->second);
            auto it = defs.find(nm);
            if (it != defs.end() && !seen.count(nm)) { seen.insert(nm); state_deps(it->second->body.get(), seen, vars); }
        }
        each_child(*n, [&](const Node &c) { state_deps(&c, seen, vars); });
    }
    // UNCHANGED <<a, b, vars>> flattened to variable indices (definitions that are tuples of variables are expanded)
    bool unchanged_vars(const Node *n, std::vector<int> &out) {
        if (n->k == N_PAREN) return unchanged_vars(n->kids[0].get(), out);
        if (n->k == N_TUPLE) { for (auto &x : n->kids) if (!unchanged_vars(x.get(), out)) return false; return true; }
        if (n->k == N_ID) {
            const int name = resolve(nsym(n));
            auto v = varidx.find(name);
            if (v != varidx.end()) { out.push_back(v->second); return true; }
            auto d = defs.find(name);
            if (d != defs.end()) return unchanged_vars(d->second->body.get(), out);
        }
        return false;
    }
    const std::vector<int> *uvars_of(const Node *n) {  // of an N_UNCHANGED node; null when it is UNCHANGED of a state function
        if (!n->uv_state) { n->uvars.clear(); n->uv_state = unchanged_vars(n->kids[0].get(), n->uvars) ? 1 : 2; }
        return n->uv_state == 1 ? &n->uvars : nullptr;
    }

    // ------------------------------------------------------------------ VALUE mode
    // st: the current state; nx: the (partial) next state inside an action, null outside
    V force(Thunk *t, State *nx) {
        if (t->done) return t->val;
        V v = ev(t->n, t->env, t->st, nx);
        if (t->memo) { t->val = v; t->done = true; }
        return v;
    }
    // an argument as an environment entry: a thunk, or an operator value for LAMBDA / an operator passed by name
    Env *bind_arg(int psym, const Node *a, int arity, Env *env, const State *st, Env *next) {
        Env *e = anew<Env>();
        e->sym = psym; e->next = next;
        if (a->k == N_LAMBDA) {
            OpVal *op = anew<OpVal>();
            for (auto &p : a->names) op->params.push_back(intern(p));
            op->body = a->kids[0].get(); op->env = env; op->name = "LAMBDA";
            op->primed = primed_rt(a, env);
            e->op = op;
            return e;
        }
        if (a->k == N_NUM || a->k == N_STR || a->k == N_BOOL) { e->val = ev(a, env, st, nullptr); return e; }
        if (a->k == N_ID) {
            const int name = nsym(a);
            if (Env *l = env_find(env, name)) { e->val = l->val; e->th = l->th; e->op = l->op; e->rec = l->rec; return e; }  // pass the entry on
            if (arity > 0) {
                const int nm = resolve(name);
                auto d = defs.find(nm);
                if (d != defs.end()) {
                    OpVal *op = anew<OpVal>();
                    for (auto &p : d->second->params) op->params.push_back(p.first);
                    op->body = d->second->body.get(); op->name = d->second->name; op->primed = def_primed(d->second);
                    e->op = op;
                    return e;
                }
                auto b = builtin_ops.find(nm);
                if (b != builtin_ops.end()) { OpVal *op = anew<OpVal>(); op->builtin = b->second.first; op->params.resize((size_t)arity); op->name = a->s; e->op = op; return e; }
            }
        }
        Thunk *t = anew<Thunk>();
        t->n = a; t->env = env; t->st = st; t->memo = !primed_rt(a, env);
        e->th = t;
        return e;
    }
    V call_builtin(int code, const std::vector<V> &a) {
        switch (code) {
            case B_CARD: return mk_int((long)to_set(a[0])->items.size());
            case B_ISFINITE: return mk_bool(a[0]->k == K_SET);
            case B_LEN: return mk_int((long)as_seq(a[0], "Len").size());
            case B_APPEND: { std::vector<V> it = as_seq(a[0], "Append"); it.push_back(a[1]); return mk_tuple(std::move(it)); }
            case B_HEAD: { auto &s = as_seq(a[0], "Head"); if (s.empty()) fail("Head of the empty sequence"); return s[0]; }
            case B_TAIL: { auto &s = as_seq(a[0], "Tail"); if (s.empty()) fail("Tail of the empty sequence"); return mk_tuple(std::vector<V>(s.begin() + 1, s.end())); }
            case B_SUBSEQ: {
                auto &s = as_seq(a[0], "SubSeq");
                const long m = as_int(a[1], "SubSeq"), n = as_int(a[2], "SubSeq");
                if (m > n) return g_empty_tuple;
                if (m < 1 || n > (long)s.size()) fail("SubSeq(" + fmt(a[0]) + ", " + std::to_string(m) + ", " + std::to_string(n) + ") out of range");
                return mk_tuple(std::vector<V>(s.begin() + (m - 1), s.begin() + n));
            }
            case B_SEQ: return mk_lazy(L_SEQ, a[0]);
            case B_PERMUTATIONS: {
                std::vector<V> el = to_set(a[0])->items, out;
                std::vector<size_t> p(el.size());
                for (size_t i = 0; i < p.size(); i++) p[i] = i;
                do {
                    std::vector<std::pair<V, V>> kv;
                    for (size_t i = 0; i < el.size(); i++) kv.emplace_back(el[i], el[p[i]]);
                    out.push_back(mk_fn(kv));
                } while (std::next_permutation(p.begin(), p.end()));
                return mk_set(out);
            }
            // TLC prints when it EVALUATES Print / PrintT; here only where the caller asked for it (the ASSUMEs of a model without a
            // behaviour spec: AsynchronousInterface/PrintValues.tla) — not once per state of a search
            case B_PRINT: if (g_print_sink) g_print_sink->push_back(fmt(a[0]) + "  " + fmt(a[1])); return a[1];
            case B_PRINTT: if (g_print_sink) g_print_sink->push_back(fmt(a[0])); return g_true;
            case B_TOSTRING: return mk_str(fmt(a[0]));
            default: fail("built-in operator used in an unsupported position");
        }
    }
    V call_op_vals(OpVal *op, const std::vector<V> &vals, const State *st, State *nx) {
        if (op->builtin >= 0) return call_builtin(op->builtin, vals);
        if (vals.size() != op->params.size()) fail("operator " + op->name + " takes " + std::to_string(op->params.size()) + " arguments");
        Env *e2 = op->env;
        for (size_t i = 0; i < vals.size(); i++) e2 = bind_val(op->params[i], vals[i], e2);
        return ev(op->body, e2, st, nx);
    }
    Env *bind_args(const std::vector<int> &params, const std::vector<NodeP> &args, const std::vector<std::pair<int, int>> *arities, Env *env,
                   const State *st, Env *base, const std::string &name) {
        if (args.size() != params.size()) fail("operator " + name + " takes " + std::to_string(params.size()) + " arguments, " + std::to_string(args.size()) + " given");
        Env *e2 = base;
        for (size_t i = 0; i < args.size(); i++) e2 = bind_arg(params[i], args[i].get(), arities ? (*arities)[i].second : 0, env, st, e2);
        return e2;
    }
    V global_apply(GDef *d, const std::vector<NodeP> &args, Env *env, const State *st, State *nx) {
        if (d->params.empty()) {
            if (!args.empty()) fail("operator " + d->name + " takes 0 arguments, " + std::to_string(args.size()) + " given");
            if (d->const_val) return d->const_val;
            if (d->is_const < 0) { std::set<int> seen; d->is_const = !mentions_state(d->body.get(), seen); }
            if (d->is_const) { d->const_val = ev(d->body.get(), nullptr, st, nx); return d->const_val; }  // constant-level: evaluated once
            if (!st || def_primed(d)) return ev(d->body.get(), nullptr, st, nx);
            if (!d->deps_known) { std::set<int> seen, vars; state_deps(d->body.get(), seen, vars); d->deps.assign(vars.begin(), vars.end()); d->deps_known = 1; }
            for (int i : d->deps) if (!(*st)[(size_t)i]) return ev(d->body.get(), nullptr, st, nx);  // inside Init
            for (auto &m : d->memo) {
                if (!m.val) continue;
                bool hit = true;
                for (size_t k = 0; k < d->deps.size() && hit; k++) hit = m.key[k].get() == (*st)[(size_t)d->deps[k]].get();
                if (hit) return m.val;
            }
            V v = ev(d->body.get(), nullptr, st, nx);
            GDef::Memo &m = d->memo[d->memo_next++ & 3];
            m.key.clear();
            for (int i : d->deps) m.key.push_back((*st)[(size_t)i]);
            m.val = v;
            return v;
        }
        std::vector<int> ps;
        for (auto &p : d->params) ps.push_back(p.first);
        Env *e2 = bind_args(ps, args, &d->params, env, st, nullptr, d->name);
        return ev(d->body.get(), e2, st, nx);
    }
    Env *bind_pattern(const Bound &b, const V &val, Env *env) {
        const std::vector<int> &syms = bsyms(b);
        if (!b.is_tuple) return bind_val(syms[0], val, env);
        if (val->k != K_TUPLE || val->items.size() != syms.size()) fail("tuple pattern bound to " + fmt(val));
        for (size_t i = 0; i < syms.size(); i++) env = bind_val(syms[i], val->items[i], env);
        return env;
    }
    // the environments of `x \in S, <<a, b>> \in T, ...` (every set is evaluated in the outer environment); f returns false to stop
    void for_bounds(const std::vector<Bound> &bs, Env *env, const State *st, State *nx, const std::function<bool(Env *, const std::vector<V> &)> &f) {
        if (bs.size() == 1) {
            std::vector<V> combo(1);
            enumerate(ev(bs[0].dom.get(), env, st, nx), [&](const V &x) { combo[0] = x; return f(bind_pattern(bs[0], x, env), combo); });
            return;
        }
        std::vector<std::vector<V>> sets;
        for (auto &b : bs) { sets.push_back(elements(ev(b.dom.get(), env, st, nx))); if (sets.back().empty()) return; }
        std::vector<size_t> c(bs.size(), 0);
        std::vector<V> combo(bs.size());
        for (;;) {
            Env *e2 = env;
            for (size_t i = 0; i < bs.size(); i++) { combo[i] = sets[i][c[i]]; e2 = bind_pattern(bs[i], combo[i], e2); }
            if (!f(e2, combo)) return;
            size_t i = bs.size();
            while (i > 0) { if (++c[i - 1] < sets[i - 1].size()) break; c[i - 1] = 0; i--; }
            if (i == 0) return;
        }
    }
    Env *bind_let(const Node *n, Env *env, const State *st) {
        Env *e2 = env;
        std::vector<Env *> entries;
        for (auto &d : n->defs) { Env *e = anew<Env>(); e->sym = d.sym < 0 ? (d.sym = intern(d.name)) : d.sym; e->next = e2; e2 = e; entries.push_back(e); }
        // every definition sees itself (RECURSIVE) and its siblings: the closures take the complete environment
        for (size_t i = 0; i < n->defs.size(); i++) {
            const Def &d = n->defs[i];
            const bool pr = primed_rt(d.body.get(), e2);
            if (!d.params.empty()) {
                OpVal *op = anew<OpVal>();
                op->params = dpsyms(d); op->body = d.body.get(); op->env = e2; op->primed = pr; op->is_let = true; op->name = d.name;
                entries[i]->op = op;
            } else {
                Thunk *t = anew<Thunk>();
                t->n = d.body.get(); t->env = e2; t->st = st; t->memo = !pr; t->is_def = true;
                entries[i]->th = t;
            }
        }
        return e2;
    }
    V except_update(const V &f, const std::vector<NodeP> &path, size_t k, const Node *val, Env *env, const State *st, State *nx) {
        V key = ev(path[k].get(), env, st, nx);
        V old = fn_apply(f, key);
        V nw = k + 1 == path.size() ? ev(val, bind_val(sym_at, old, env), st, nx) : except_update(old, path, k + 1, val, env, st, nx);
        if (f->k == K_TUPLE) { std::vector<V> it = f->items; it[(size_t)key->i - 1] = nw; return mk_tuple(std::move(it)); }
        auto r = std::make_shared<Val>(*f);
        auto it = std::lower_bound(r->fn.begin(), r->fn.end(), key, [](const std::pair<V, V> &p, const V &kk) { return cmp(p.first, kk) < 0; });
        it->second = nw;
        return r;
    }
    V rec_apply(RecFn *r, const V &a, State *nx) {
        for (auto &p : r->cache) if (cmp(p.first, a) == 0) return p.second;
        if (!set_in(a, r->dom)) fail("function applied outside its domain: " + fmt(a) + " not in the domain of " + r->n->s);
        const Bound &b = r->n->bounds[0];
        V v = ev(r->n->kids[0].get(), bind_pattern(b, a, r->env), r->st, nx);
        r->cache.emplace_back(a, v);
        return v;
    }
    V rec_function(const Node *n, Env *env, const State *st, State *nx) {
        if (n->bounds.size() != 1) fail("recursive function definitions with several bounds are not supported");
        RecFn *r = anew<RecFn>();
        Env *self = anew<Env>();
        self->sym = nsym(n); self->rec = r; self->next = env;
        r->n = n; r->env = self; r->st = st; r->dom = ev(n->bounds[0].dom.get(), env, st, nx);
        std::vector<std::pair<V, V>> kv;
        for (auto &x : elements(r->dom)) kv.emplace_back(x, rec_apply(r, x, nx));
        return mk_fn(std::move(kv));
    }
    V binop(int opc, const Node *n, Env *env, const State *st, State *nx) {
        const Node *an = n->kids[0].get(), *bn = n->kids[1].get();
        if (opc == OP_IMP) {
            V a = ev(an, env, st, nx);
            if (a->k == K_BOOL && !a->i) return g_true;
            return ev(bn, env, st, nx);
        }
        V a = ev(an, env, st, nx), b = ev(bn, env, st, nx);
        switch (opc) {
            case OP_EQ: return mk_bool(veq(a, b));
            case OP_NE: return mk_bool(!veq(a, b));
            case OP_IN: return mk_bool(set_in(a, b));
            case OP_NOTIN: return mk_bool(!set_in(a, b));
            case OP_EQUIV: return mk_bool(as_bool(a, "<=> applied to") == as_bool(b, "<=> applied to"));
            case OP_ADD: return mk_int(as_int(a, "+") + as_int(b, "+"));
            case OP_SUB: return mk_int(as_int(a, "-") - as_int(b, "-"));
            case OP_MUL: return mk_int(as_int(a, "*") * as_int(b, "*"));
            case OP_DIV: { const long x = as_int(a, "\\div"), y = as_int(b, "\\div"); if (!y) fail("division by zero"); long q = x / y; if ((x % y) && ((x < 0) != (y < 0))) q--; return mk_int(q); }
            case OP_MOD: { const long x = as_int(a, "%"), y = as_int(b, "%"); if (y <= 0) fail("% with a non-positive modulus"); long r = x % y; if (r < 0) r += y; return mk_int(r); }
            case OP_POW: { long x = as_int(a, "^"), y = as_int(b, "^"), r = 1; if (y < 0) fail("^ with a negative exponent"); while (y-- > 0) r *= x; return mk_int(r); }
            case OP_LT: return mk_bool(as_int(a, "<") < as_int(b, "<"));
            case OP_GT: return mk_bool(as_int(a, ">") > as_int(b, ">"));
            case OP_LE: return mk_bool(as_int(a, "<=") <= as_int(b, "<="));
            case OP_GE: return mk_bool(as_int(a, ">=") >= as_int(b, ">="));
            case OP_RANGE: { const long lo = as_int(a, ".."), hi = as_int(b, ".."); std::vector<V> it; for (long i = lo; i <= hi; i++) it.push_back(mk_int(i)); return mk_set_sorted(std::move(it)); }
            case OP_CUP: { std::vector<V> it = to_set(a)->items; const V bs = to_set(b); it.insert(it.end(), bs->items.begin(), bs->items.end()); return mk_set(std::move(it)); }
            case OP_CAP: { std::vector<V> it; enumerate(a, [&](const V &x) { if (set_in(x, b)) it.push_back(x); return true; }); return mk_set(std::move(it)); }
            case OP_SETMINUS: { std::vector<V> it; enumerate(a, [&](const V &x) { if (!set_in(x, b)) it.push_back(x); return true; }); return mk_set(std::move(it)); }
            case OP_SUBSETEQ: { bool r = true; enumerate(a, [&](const V &x) { r = set_in(x, b); return r; }); return mk_bool(r); }
            case OP_PSUBSET: { bool r = true; enumerate(a, [&](const V &x) { r = set_in(x, b); return r; }); return mk_bool(r && !veq(a, b)); }
            case OP_CONCAT: { std::vector<V> it = as_seq(a, "\\o"); auto &s2 = as_seq(b, "\\o"); it.insert(it.end(), s2.begin(), s2.end()); return mk_tuple(std::move(it)); }
            case OP_MAPSTO: return mk_fn({{a, b}});
            case OP_ATAT: {  // f @@ g: union of the domains, f wins (TLC.tla:11-12)
                std::vector<std::pair<V, V>> kv = fn_items(a);
                for (auto &p : fn_items(b)) if (!fn_find(a, p.first)) kv.push_back(p);
                return mk_fn(kv);
            }
            case OP_TIMES: { std::vector<V> it; const std::vector<V> ys = elements(b); enumerate(a, [&](const V &x) { for (auto &y : ys) it.push_back(mk_tuple({x, y})); return true; }); return mk_set(std::move(it)); }
            default: fail("operator " + n->s + " is not supported");
        }
    }
    V read_var(int i, const std::string &name, const State *st, State *nx) {
        V v = st ? (*st)[(size_t)i] : nullptr;
        if (!v && nx) v = (*nx)[(size_t)i];  // inside Init a later conjunct reads what an earlier one assigned (MCConsensus.tla:19-20)
        if (!v) fail(name + " is read before the initial predicate gives it a value");
        return v;
    }
    V ev_id(const Node *n, Env *env, const State *st, State *nx) {
        int name = nsym(n);
        if (Env *e = env_find(env, name)) {
            if (e->th) return force(e->th, nx);
            if (e->rec) fail("recursive function " + n->s + " used as a value inside its own definition");
            if (e->op) { if (!e->op->params.empty() || e->op->builtin >= 0) fail("operator " + e->op->name + " used as a value"); return ev(e->op->body, e->op->env, st, nx); }
            return e->val;
        }
        name = resolve(name);
        auto vi = varidx.find(name);
        if (vi != varidx.end()) return read_var(vi->second, n->s, st, nx);
        auto cv = const_vals.find(name);
        if (cv != const_vals.end()) return cv->second;
        auto d = defs.find(name);
        if (d != defs.end()) return global_apply(d->second, {}, env, st, nx);
        auto b = builtin_consts.find(name);
        if (b != builtin_consts.end()) return b->second;
        fail("unknown identifier " + n->s);
    }
    V ev_call(const Node *n, Env *env, const State *st, State *nx) {
        int name = nsym(n);
        if (Env *e = env_find(env, name)) {  // operator parameter / LET operator
            OpVal *op = e->op;
            if (!op) fail(n->s + " is not an operator");
            if (op->builtin >= 0) { std::vector<V> vals; for (auto &a : n->kids) vals.push_back(ev(a.get(), env, st, nx)); return call_builtin(op->builtin, vals); }
            return ev(op->body, bind_args(op->params, n->kids, nullptr, env, st, op->env, op->name), st, nx);
        }
        name = resolve(name);
        auto d = defs.find(name);
        if (d != defs.end()) return global_apply(d->second, n->kids, env, st, nx);
        auto b = builtin_ops.find(name);
        if (b == builtin_ops.end()) fail("unknown operator " + n->s);
        if ((int)n->kids.size() != b->second.second && !(b->second.first == B_PRINT && n->kids.size() == 1)) fail("operator " + n->s + " takes " + std::to_string(b->second.second) + " arguments");
        if (b->second.first == B_SELECTSEQ) {
            const V seq = ev(n->kids[0].get(), env, st, nx);
            Env *t = bind_arg(-1, n->kids[1].get(), 1, env, st, nullptr);
            if (!t->op) fail("SelectSeq: the second argument is not an operator");
            std::vector<V> out;
            for (auto &x : as_seq(seq, "SelectSeq")) if (as_bool(call_op_vals(t->op, {x}, st, nx), "SelectSeq test is")) out.push_back(x);
            return mk_tuple(std::move(out));
        }
        if (b->second.first == B_ASSERT) {
            V c = ev(n->kids[0].get(), env, st, nx);
            if (!(c->k == K_BOOL && c->i)) { V m = ev(n->kids[1].get(), env, st, nx); throw TlaError{m->k == K_STR ? m->s : fmt(m), true}; }
            return g_true;
        }
        std::vector<V> vals;
        for (auto &a : n->kids) vals.push_back(ev(a.get(), env, st, nx));
        if (b->second.first == B_PRINT && vals.size() == 1) return g_true;
        return call_builtin(b->second.first, vals);
    }
    V ev(const Node *n, Env *env, const State *st, State *nx) {
        switch (n->k) {
            case N_NUM: if (!n->lit) n->lit = mk_int(n->num); return n->lit;
            case N_STR: if (!n->lit) n->lit = mk_str(n->s); return n->lit;
            case N_BOOL: return mk_bool(n->num != 0);
            case N_PAREN: return ev(n->kids[0].get(), env, st, nx);
            case N_AT: { Env *e = env_find(env, sym_at); if (!e) fail("@ outside EXCEPT"); return e->val; }
            case N_NTH: return ev(nth_node(n), nullptr, st, nx);
            case N_TEMPORAL:
                if (n->s == "[]_") return mk_bool(as_bool(ev(n->kids[0].get(), env, st, nx), "[A]_v: A is") || as_bool(ev(n->kids[2].get(), env, st, nx), "UNCHANGED is"));
                if (n->s == "<<>>_") return mk_bool(as_bool(ev(n->kids[0].get(), env, st, nx), "<<A>>_v: A is") && !as_bool(ev(n->kids[2].get(), env, st, nx), "UNCHANGED is"));
                fail("temporal formula evaluated");
            case N_ID: return ev_id(n, env, st, nx);
            case N_CALL: return ev_call(n, env, st, nx);
            case N_PRIME: {
                const Node *in = n->kids[0].get();
                if (in->k == N_ID && !env_find(env, nsym(in))) {
                    auto vi = varidx.find(resolve(nsym(in)));
                    if (vi != varidx.end()) {
                        if (!nx) fail(in->s + "' evaluated outside an action");
                        const V &v = (*nx)[(size_t)vi->second];
                        if (!v) fail(in->s + "' is read before it is assigned");
                        return v;
                    }
                }
                if (!nx) fail("priming outside an action");
                const State st2 = *nx;  // e' for a state function e: e in the next state (all its variables must be assigned)
                return ev(in, env, &st2, nullptr);
            }
            case N_NOT: return mk_bool(!as_bool(ev(n->kids[0].get(), env, st, nx), "~ applied to"));
            case N_NEG: return mk_int(-as_int(ev(n->kids[0].get(), env, st, nx), "-"));
            case N_CONJ: for (auto &x : n->kids) if (!as_bool(ev(x.get(), env, st, nx), "/\\ applied to")) return g_false; return g_true;
            case N_DISJ: for (auto &x : n->kids) if (as_bool(ev(x.get(), env, st, nx), "\\/ applied to")) return g_true; return g_false;
            case N_IF: return ev(n->kids[as_bool(ev(n->kids[0].get(), env, st, nx), "IF condition is") ? 1 : 2].get(), env, st, nx);
            case N_CASE: {
                const size_t arms = (n->kids.size() - (size_t)n->num) / 2;
                for (size_t i = 0; i < arms; i++) { V g = ev(n->kids[2 * i].get(), env, st, nx); if (g->k == K_BOOL && g->i) return ev(n->kids[2 * i + 1].get(), env, st, nx); }
                if (!n->num) fail("CASE: no arm is true and there is no OTHER");
                return ev(n->kids.back().get(), env, st, nx);
            }
            case N_QUANT: {
                const bool ex = n->s == "E";
                bool r = !ex;
                for_bounds(n->bounds, env, st, nx, [&](Env *e2, const std::vector<V> &) {
                    const bool b = as_bool(ev(n->kids[0].get(), e2, st, nx), ex ? "\\E body is" : "\\A body is");
                    if (b == ex) { r = ex; return false; }
                    return true;
                });
                return mk_bool(r);
            }
            case N_CHOOSE_UNB: fail("TLC cannot evaluate the unbounded CHOOSE " + n->s + " : ... (give the defined symbol a model value in the cfg)");
            case N_CHOOSE: {
                if (n->bounds.size() != 1) fail("CHOOSE with several bounds");
                std::vector<V> el = elements(ev(n->bounds[0].dom.get(), env, st, nx));
                std::sort(el.begin(), el.end(), VLess());
                for (auto &v : el) { V r = ev(n->kids[0].get(), bind_pattern(n->bounds[0], v, env), st, nx); if (r->k == K_BOOL && r->i) return v; }
                fail("CHOOSE: no element satisfies the predicate");
            }
            case N_SETENUM: { std::vector<V> it; for (auto &x : n->kids) it.push_back(ev(x.get(), env, st, nx)); return mk_set(std::move(it)); }
            case N_TUPLE: { std::vector<V> it; for (auto &x : n->kids) it.push_back(ev(x.get(), env, st, nx)); return mk_tuple(std::move(it)); }
            case N_SETFILTER: {
                std::vector<V> out;
                enumerate(ev(n->bounds[0].dom.get(), env, st, nx), [&](const V &v) {
                    if (as_bool(ev(n->kids[0].get(), bind_pattern(n->bounds[0], v, env), st, nx), "set filter predicate is")) out.push_back(v);
                    return true;
                });
                return mk_set(std::move(out));
            }
            case N_SETMAP: {
                std::vector<V> out;
                for_bounds(n->bounds, env, st, nx, [&](Env *e2, const std::vector<V> &) { out.push_back(ev(n->kids[0].get(), e2, st, nx)); return true; });
                return mk_set(std::move(out));
            }
            case N_FNDEF: {
                if (!n->s.empty()) return rec_function(n, env, st, nx);
                std::vector<std::pair<V, V>> kv;
                for_bounds(n->bounds, env, st, nx, [&](Env *e2, const std::vector<V> &combo) {
                    kv.emplace_back(combo.size() > 1 ? mk_tuple(combo) : combo[0], ev(n->kids[0].get(), e2, st, nx));
                    return true;
                });
                return mk_fn(std::move(kv));
            }
            case N_RECORD: {
                if (n->keys.size() != n->fields.size()) { n->keys.clear(); for (auto &f : n->fields) n->keys.push_back(mk_str(f.first)); }
                std::vector<std::pair<V, V>> kv;
                for (size_t i = 0; i < n->fields.size(); i++) kv.emplace_back(n->keys[i], ev(n->fields[i].second.get(), env, st, nx));
                return mk_fn(std::move(kv));
            }
            case N_RECORDSET: { auto r = std::make_shared<Val>(); r->k = K_LAZY; r->i = L_RECSET; for (auto &f : n->fields) r->fields.emplace_back(f.first, ev(f.second.get(), env, st, nx)); return r; }
            case N_FNSET: { V a = ev(n->kids[0].get(), env, st, nx), b = ev(n->kids[1].get(), env, st, nx); return mk_lazy(L_FNSET, a, b); }
            case N_EXCEPT: {
                V f = ev(n->kids[0].get(), env, st, nx);
                for (auto &u : n->ups) f = except_update(f, u.first, 0, u.second.get(), env, st, nx);
                return f;
            }
            case N_IDX: {
                if (n->kids[0]->k == N_ID) { Env *e = env_find(env, nsym(n->kids[0].get())); if (e && e->rec) return rec_apply(e->rec, ev(n->kids[1].get(), env, st, nx), nx); }
                V f = ev(n->kids[0].get(), env, st, nx);
                return fn_apply(f, ev(n->kids[1].get(), env, st, nx));
            }
            case N_PRE: {
                V x = ev(n->kids[0].get(), env, st, nx);
                if (n->s == "DOMAIN") return fn_domain(x);
                if (n->s == "SUBSET") return mk_lazy(L_POWERSET, x);
                std::vector<V> out;
                enumerate(x, [&](const V &s) { enumerate(s, [&](const V &y) { out.push_back(y); return true; }); return true; });
                return mk_set(std::move(out));
            }
            case N_LET: return ev(n->kids[0].get(), bind_let(n, env, st), st, nx);
            case N_LAMBDA: fail("LAMBDA outside an argument position");
            case N_UNCHANGED: {
                const std::vector<int> *idx = uvars_of(n);
                if (!idx) {  // UNCHANGED e for a state function e (an instantiated module's variable, Paxos.tla:201): e' = e
                    if (!nx) fail("UNCHANGED outside an action");
                    const State st2 = *nx;
                    return mk_bool(veq(ev(n->kids[0].get(), env, &st2, nullptr), ev(n->kids[0].get(), env, st, nx)));
                }
                if (!nx) fail("UNCHANGED outside an action");
                for (int i : *idx) {
                    const V &v = (*nx)[(size_t)i];
                    if (!v) fail("UNCHANGED " + variables[(size_t)i] + " read before it is assigned");
                    if (!veq(v, (*st)[(size_t)i])) return g_false;
                }
                return g_true;
            }
            case N_OP: {
                if (n->opc < 0) { auto it = OPCODES.find(n->s); n->opc = it == OPCODES.end() ? OP_UNSUPPORTED : it->second; }
                return binop(n->opc, n, env, st, nx);
            }
            case N_ENABLED: fail("cannot evaluate an ENABLED expression");
            case N_INSTANCE: fail("cannot evaluate an INSTANCE expression");
        }
        fail("cannot evaluate this expression");
    }

    // ------------------------------------------------------------------ ACTION mode
    // act(n, ..., nx, k): k() is called once per way of satisfying n, with nx extended by the assignments n makes (and
    // restored afterwards)
    void act_test(const Node *n, Env *env, const State *st, State &nx, const Cont &k) {
        if (as_bool(ev(n, env, st, &nx), "action conjunct is")) k();
    }
    void act_conj(const Node *n, size_t i, Env *env, const State *st, State &nx, const Cont &k) {
        if (i + 1 == n->kids.size()) { act(n->kids[i].get(), env, st, nx, k); return; }
        act(n->kids[i].get(), env, st, nx, [&]() { act_conj(n, i + 1, env, st, nx, k); });
    }
    // Which ACTION produced a successor, the way TLC names it in a counterexample (README.md:278 "<Action line 35, col 19 to line
    // 40, col 42 of module pcal_intro>"): Next is taken apart through disjunctions, \E and the definitions that are nothing but
    // those; the first definition whose body is something else is the action, and its body's span is its location.
    const GDef *cur_action = nullptr;
    void act_global(GDef *d, const std::vector<NodeP> &args, Env *env, const State *st, State &nx, const Cont &k) {
        std::vector<int> ps;
        for (auto &p : d->params) ps.push_back(p.first);
        const Node *b = d->body.get();
        while (b->k == N_PAREN) b = b->kids[0].get();
        const bool transparent = b->k == N_DISJ || (b->k == N_QUANT && b->s == "E") || ((b->k == N_ID || b->k == N_CALL) && defs.count(resolve(nsym(b))));
        const GDef *saved = cur_action;
        if (!cur_action && !transparent) cur_action = d;
        try { act(d->body.get(), bind_args(ps, args, &d->params, env, st, nullptr, d->name), st, nx, k); }
        catch (...) { cur_action = saved; throw; }
        cur_action = saved;
    }
    void act_call(const Node *n, const std::vector<NodeP> &args, Env *env, const State *st, State &nx, const Cont &k) {
        int name = nsym(n);
        if (Env *e = env_find(env, name)) {
            if (e->op && e->op->is_let) { act(e->op->body, bind_args(e->op->params, args, nullptr, env, st, e->op->env, e->op->name), st, nx, k); return; }
            if (e->th && e->th->is_def && !e->th->memo && args.empty()) { act(e->th->n, e->th->env, e->th->st, nx, k); return; }
            act_test(n, env, st, nx, k);
            return;
        }
        name = resolve(name);
        auto d = defs.find(name);
        if (d != defs.end()) { act_global(d->second, args, env, st, nx, k); return; }
        act_test(n, env, st, nx, k);
    }
    // is the left side of `=` / `\in` a primed variable — directly, or an operator parameter whose argument is one
    // (CachingMemory/InternalMemory.tla:17 Send(p, req, memInt, memInt') with MCSend(p, d, old, new) == new = <<p, d>>)
    int primed_var(const Node *lhs, Env *env) {
        for (int hops = 0; hops < 32; hops++) {
            while (lhs->k == N_PAREN) lhs = lhs->kids[0].get();
            if (lhs->k == N_PRIME) {
                const Node *in = lhs->kids[0].get();
                if (in->k != N_ID || env_find(env, nsym(in))) return -1;
                auto vi = varidx.find(resolve(nsym(in)));
                return vi == varidx.end() ? -1 : vi->second;
            }
            if (lhs->k != N_ID) return -1;
            Env *e = env_find(env, nsym(lhs));
            if (!e || !e->th || e->th->is_def) return -1;
            env = e->th->env;
            lhs = e->th->n;
        }
        return -1;
    }
    // One branch of a disjunction / of \E: while the successors of a state are generated (branch_errors), an evaluation error or a
    // failed Assert inside one branch — one action instance — does not lose the successors of the other branches: the first error
    // is kept and reported for the state (the GPU engine evaluates a state's action slots independently in the same way).
    bool branch_errors = false;
    uint64_t failed_branches = 0;
    std::unique_ptr<TlaError> first_error;
    void act_branch(const Node *n, Env *env, const State *st, State &nx, const Cont &k) {
        if (!branch_errors) { act(n, env, st, nx, k); return; }
        const State saved = nx;
        try { act(n, env, st, nx, k); }
        catch (TlaError &e) {
            nx = saved;  // (assignments of the abandoned branch)
            failed_branches++;  // an enabled action instance whose evaluation failed: counted as generated, like the engine does
            if (!first_error) first_error.reset(new TlaError(e));
        }
    }
    void act(const Node *n, Env *env, const State *st, State &nx, const Cont &k) {
        switch (n->k) {
            case N_PAREN: act(n->kids[0].get(), env, st, nx, k); return;
            case N_CONJ: act_conj(n, 0, env, st, nx, k); return;
            case N_DISJ: for (auto &x : n->kids) act_branch(x.get(), env, st, nx, k); return;
            case N_IF: act(n->kids[as_bool(ev(n->kids[0].get(), env, st, &nx), "IF condition is") ? 1 : 2].get(), env, st, nx, k); return;
            case N_CASE: {
                const size_t arms = (n->kids.size() - (size_t)n->num) / 2;
                for (size_t i = 0; i < arms; i++) { V g = ev(n->kids[2 * i].get(), env, st, &nx); if (g->k == K_BOOL && g->i) { act(n->kids[2 * i + 1].get(), env, st, nx, k); return; } }
                if (!n->num) fail("CASE: no arm is true and there is no OTHER");
                act(n->kids.back().get(), env, st, nx, k);
                return;
            }
            case N_QUANT:
                if (n->s != "E") { act_test(n, env, st, nx, k); return; }
                for_bounds(n->bounds, env, st, &nx, [&](Env *e2, const std::vector<V> &) { act_branch(n->kids[0].get(), e2, st, nx, k); return true; });
                return;
            case N_LET: act(n->kids[0].get(), bind_let(n, env, st), st, nx, k); return;
            case N_UNCHANGED: {
                const std::vector<int> *idx = uvars_of(n);
                if (!idx) { act_test(n, env, st, nx, k); return; }
                std::vector<int> set;
                bool ok = true;
                for (int i : *idx) {
                    V &v = nx[(size_t)i];
                    if (!v) { v = (*st)[(size_t)i]; set.push_back(i); }
                    else if (!veq(v, (*st)[(size_t)i])) { ok = false; break; }
                }
                if (ok) k();
                for (int i : set) nx[(size_t)i] = nullptr;
                return;
            }
            case N_OP: {
                if (n->s == "=" || n->s == "\\in") {
                    const int pv = primed_var(n->kids[0].get(), env);
                    {
                        if (pv >= 0) {
                            const size_t i = (size_t)pv;
                            V rhs = ev(n->kids[1].get(), env, st, &nx);
                            if (n->s == "=") {  // x' = e assigns when x' has no value yet and is an equality test otherwise
                                if (!nx[i]) { nx[i] = rhs; k(); nx[i] = nullptr; }
                                else if (veq(nx[i], rhs)) k();
                            } else {
                                if (!nx[i]) { for (auto &v : elements(rhs)) { nx[i] = v; k(); } nx[i] = nullptr; }
                                else if (set_in(nx[i], rhs)) k();
                            }
                            return;
                        }
                    }
                }
                act_test(n, env, st, nx, k);
                return;
            }
            case N_TEMPORAL:
                if (n->s == "[]_") { act(n->kids[0].get(), env, st, nx, k); act(n->kids[2].get(), env, st, nx, k); return; }
                if (n->s == "<<>>_") { act(n->kids[0].get(), env, st, nx, [&]() { if (!as_bool(ev(n->kids[2].get(), env, st, &nx), "UNCHANGED is")) k(); }); return; }
                act_test(n, env, st, nx, k);
                return;
            case N_ID: act_call(n, {}, env, st, nx, k); return;
            case N_CALL: act_call(n, n->kids, env, st, nx, k); return;
            default: act_test(n, env, st, nx, k); return;
        }
    }

    // ------------------------------------------------------------------ entry points
    // Init is evaluated like an action in which every UNPRIMED variable is assigned: the action evaluator runs on a copy of the
    // tree where `x = e` / `x \in S` for a variable x stands for x' = e / x' \in S
    NodeP init_tree(const NodeP &n) {
        if (n->k == N_OP && (n->s == "=" || n->s == "\\in")) {
            const Node *l = n->kids[0].get();
            while (l->k == N_PAREN) l = l->kids[0].get();
            if (l->k == N_ID && varidx.count(resolve(nsym(l)))) {
                auto r = std::make_shared<Node>(*n);
                auto p = node(N_PRIME, n->line);
                p->kids = {n->kids[0]};
                while (p->kids[0]->k == N_PAREN) p->kids[0] = p->kids[0]->kids[0];
                r->kids[0] = p; r->pinfo.reset();
                return r;
            }
            return n;
        }
        if (n->k == N_CONJ || n->k == N_DISJ || n->k == N_PAREN || (n->k == N_QUANT && n->s == "E")) {
            auto r = std::make_shared<Node>(*n);
            r->pinfo.reset();
            for (auto &x : r->kids) x = init_tree(x);
            return r;
        }
        if (n->k == N_ID) { auto d = defs.find(nsym(n.get())); if (d != defs.end() && d->second->params.empty()) return init_tree(d->second->body); }
        return n;
    }
    std::string state_text(const State &st, const char *sep = "\n") const {
        std::string o;
        for (size_t i = 0; i < variables.size(); i++) o += (i ? sep : "") + std::string("/\\ ") + variables[i] + " = " + fmt(st[i]);
        return o;
    }
};

// =============================================================================================== cfg
// TLC configuration file (grammar: examples/SpecifyingSystems/TLC/ConfigFileGrammar.tla:4-32)
struct Cfg {
    std::string spec, init, next, symmetry, view;
    std::vector<std::string> invariants, constraints, properties, action_constraints;
    std::vector<std::pair<std::string, V>> constants;
    std::vector<std::pair<std::string, std::string>> overrides;
    std::map<std::pair<std::string, std::string>, std::string> scoped;
    int check_deadlock = -1;   // CHECK_DEADLOCK TRUE | FALSE (-1: not given)
};
Cfg parse_cfg(const std::string &text) {
    static const std::set<std::string> KW = {"SPECIFICATION", "INIT", "NEXT", "INVARIANT", "INVARIANTS", "CONSTRAINT", "CONSTRAINTS", "CONSTANT", "CONSTANTS",
        "SYMMETRY", "PROPERTY", "PROPERTIES", "ACTION_CONSTRAINT", "ACTION_CONSTRAINTS", "VIEW", "CHECK_DEADLOCK"};
    std::vector<Tok> toks = lex(text);
    Cfg out;
    size_t i = 0;
    auto is_kw = [&](size_t j) { return toks[j].k == Tok::ID && KW.count(toks[j].s); };
    std::function<V(size_t &)> value = [&](size_t &j) -> V {
        const Tok &t = toks[j];
        if (t.k == Tok::NUM) { j++; return mk_int(atol(t.s.c_str())); }
        if (t.k == Tok::STR) { j++; return mk_str(t.s); }
        if (t.k == Tok::SYM && t.s == "-" && toks[j + 1].k == Tok::NUM) { j += 2; return mk_int(-atol(toks[j - 1].s.c_str())); }
        if (t.k == Tok::SYM && t.s == "{") {
            j++;
            std::vector<V> items;
            while (!(toks[j].k == Tok::SYM && toks[j].s == "}")) {
                if (toks[j].k == Tok::END) throw SyntaxErr{"cfg: unterminated set at line " + std::to_string(t.line)};
                if (toks[j].k == Tok::SYM && toks[j].s == ",") { j++; continue; }
                items.push_back(value(j));
            }
            j++;
            return mk_set(std::move(items));
        }
        if (t.k == Tok::ID) { j++; if (t.s == "TRUE" || t.s == "FALSE") return mk_bool(t.s == "TRUE"); return mk_mv(t.s); }
        throw SyntaxErr{"cfg: unexpected '" + t.s + "' at line " + std::to_string(t.line)};
    };
    while (toks[i].k != Tok::END) {
        if (!is_kw(i)) throw SyntaxErr{"cfg: expected a statement keyword at line " + std::to_string(toks[i].line) + ", found '" + toks[i].s + "'"};
        const std::string kw = toks[i++].s;
        if (kw == "CHECK_DEADLOCK") {   // TLC2's statement: TRUE | FALSE
            if (toks[i].k != Tok::ID || (toks[i].s != "TRUE" && toks[i].s != "FALSE")) throw SyntaxErr{"cfg: CHECK_DEADLOCK needs TRUE or FALSE at line " + std::to_string(toks[i].line)};
            out.check_deadlock = toks[i++].s == "TRUE";
        } else if (kw == "SPECIFICATION" || kw == "INIT" || kw == "NEXT" || kw == "SYMMETRY" || kw == "VIEW") {
            if (toks[i].k != Tok::ID) throw SyntaxErr{"cfg: " + kw + " needs a name at line " + std::to_string(toks[i].line)};
            (kw == "SPECIFICATION" ? out.spec : kw == "INIT" ? out.init : kw == "NEXT" ? out.next : kw == "SYMMETRY" ? out.symmetry : out.view) = toks[i++].s;
        } else if (kw.compare(0, 3, "INV") == 0 || kw.compare(0, 10, "CONSTRAINT") == 0 || kw.compare(0, 4, "PROP") == 0 || kw.compare(0, 6, "ACTION") == 0) {
            auto &dst = kw.compare(0, 3, "INV") == 0 ? out.invariants : kw.compare(0, 10, "CONSTRAINT") == 0 ? out.constraints
                      : kw.compare(0, 6, "ACTION") == 0 ? out.action_constraints : out.properties;
            while (toks[i].k != Tok::END && !is_kw(i)) dst.push_back(toks[i++].s);
        } else {
            while (toks[i].k != Tok::END && !is_kw(i)) {
                if (toks[i].k != Tok::ID) throw SyntaxErr{"cfg: expected a constant name at line " + std::to_string(toks[i].line) + ", found '" + toks[i].s + "'"};
                const std::string name = toks[i++].s;
                if (toks[i].k == Tok::SYM && toks[i].s == "=") { i++; out.constants.emplace_back(name, value(i)); }
                else if (toks[i].k == Tok::SYM && toks[i].s == "<-") {
                    i++;
                    if (toks[i].k == Tok::SYM && toks[i].s == "[") {  // <-[Module] Id (MCPaxos.cfg:9)
                        if (i + 3 >= toks.size() || toks[i + 1].k != Tok::ID || toks[i + 2].s != "]" || toks[i + 3].k != Tok::ID)
                            throw SyntaxErr{"cfg: expected <-[Module] Id at line " + std::to_string(toks[i].line)};
                        out.scoped[{toks[i + 1].s, name}] = toks[i + 3].s;
                        i += 4;
                        continue;
                    }
                    if (toks[i].k != Tok::ID) throw SyntaxErr{"cfg: expected a definition name after <- at line " + std::to_string(toks[i].line)};
                    out.overrides.emplace_back(name, toks[i++].s);
                } else throw SyntaxErr{"cfg: expected = or <- after " + name + " at line " + std::to_string(toks[i].line)};
            }
        }
    }
    return out;
}

// the value with every model value m replaced by g[m] (TLC's symmetry reduction applies a permutation to a whole state)
V permute(const V &v, const std::map<const Val *, V> &g) {
    switch (v->k) {
        case K_MV: { auto it = g.find(v.get()); return it == g.end() ? v : it->second; }
        case K_TUPLE: { std::vector<V> it; for (auto &x : v->items) it.push_back(permute(x, g)); return mk_tuple(std::move(it)); }
        case K_SET: { std::vector<V> it; for (auto &x : v->items) it.push_back(permute(x, g)); return mk_set(std::move(it)); }
        case K_FN: { std::vector<std::pair<V, V>> kv; for (auto &p : v->fn) kv.emplace_back(permute(p.first, g), permute(p.second, g)); return mk_fn(std::move(kv)); }
        default: return v;
    }
}

// =============================================================================================== checker
// TLC's breadth-first search over a Spec
struct Checker {
    Spec sp;
    Cfg cfg;
    NodeP init_node, next_node, init_act;
    std::vector<std::pair<std::string, NodeP>> invs, cons, acons;
    NodeP view;  // VIEW: two states with the same value of this expression are the same state (p-manual section 4.5 / TLC's -view)
    struct Prop { std::string name; std::vector<NodeP> inits; std::vector<std::pair<NodeP, NodeP>> steps; };
    std::vector<Prop> props;
    size_t n_cfg_invariants = 0;
    std::vector<std::pair<std::string, NodeP>> always;  // []P conjuncts of the PROPERTIES: checked like invariants, under the property's name
    std::vector<std::string> unchecked;                 // PROPERTIES with a liveness part (never checked here; the report names them)
    std::vector<std::map<const Val *, V>> group;
    bool has_group = false;
    bool no_behavior = false;   // the cfg names neither SPECIFICATION nor INIT / NEXT: evaluate the ASSUMEs (setup)

    bool is_temporal(const Node *n, std::set<int> &seen) {
        if (n->k == N_TEMPORAL || (n->k == N_OP && n->s == "~>")) return true;
        if (n->k == N_ID) {
            const int nm = sp.resolve(nsym(n));
            auto d = sp.defs.find(nm);
            if (d != sp.defs.end() && !seen.count(nm)) { seen.insert(nm); return is_temporal(d->second->body.get(), seen); }
            return false;
        }
        bool r = false;
        each_child(*n, [&](const Node &c) { r = r || is_temporal(&c, seen); });
        return r;
    }
    static bool is_box_action(const Node *n) { return n->k == N_TEMPORAL && n->s == "[]" && n->kids[0]->k == N_TEMPORAL && n->kids[0]->s == "[]_"; }
    static bool is_fairness(const Node *n) { return n->k == N_TEMPORAL && (n->s == "WF_" || n->s == "SF_"); }
    // Spec == Init /\ [][Next]_vars (/\ fairness): the first non-temporal conjunct is Init, [][N]_v gives Next
    void split_spec(const std::string &name) {
        auto d = sp.defs.find(intern(name));
        if (d == sp.defs.end()) fail("SPECIFICATION " + name + " is not defined");
        std::vector<NodeP> flat;
        std::function<void(const NodeP &)> walk = [&](const NodeP &n) {
            if (n->k == N_CONJ) { for (auto &x : n->kids) walk(x); return; }
            if (n->k == N_PAREN) { walk(n->kids[0]); return; }
            if (n->k == N_ID && !sp.varidx.count(nsym(n.get()))) {  // LSpec == HC /\ WF_hr(HCnxt) with HC == HCini /\ [][HCnxt]_hr (Liveness/LiveHourClock.tla)
                auto dd = sp.defs.find(sp.resolve(nsym(n.get())));
                std::set<int> seen;
                if (dd != sp.defs.end() && dd->second->params.empty() && is_temporal(dd->second->body.get(), seen)) { walk(dd->second->body); return; }
            }
            flat.push_back(n);
        };
        walk(d->second->body);
        for (auto &n : flat) {
            if (is_box_action(n.get())) next_node = n->kids[0]->kids[0];
            else if (!init_node) { std::set<int> seen; if (!is_temporal(n.get(), seen)) init_node = n; }  // not fairness, possibly quantified (InnerSerial.tla:147-156)
        }
        if (!init_node || !next_node) fail("cannot split " + name + " into Init and Next");
    }
    // PROPERTY P with P == I /\ [][A]_v (/\ fairness): what TLC checks of it without liveness — I on the initial states,
    // A \/ v' = v on every transition it generates (MCVoting.cfg:9 ConsensusSpecBar == C!Spec, MCPaxos.cfg:12)
    Prop compile_property(const std::string &name) {
        Prop p;
        p.name = name;
        std::vector<NodeP> flat;
        std::function<void(const NodeP &)> walk = [&](const NodeP &n) {
            if (n->k == N_CONJ) { for (auto &x : n->kids) walk(x); return; }
            if (n->k == N_PAREN) { walk(n->kids[0]); return; }
            if (n->k == N_ID && !sp.varidx.count(nsym(n.get()))) {
                auto d = sp.defs.find(sp.resolve(nsym(n.get())));
                std::set<int> seen;
                if (d != sp.defs.end() && d->second->params.empty() && is_temporal(d->second->body.get(), seen)) { walk(d->second->body); return; }
            }
            flat.push_back(n);
        };
        walk(node_id(name));
        bool live = false;
        for (auto &n : flat) {
            std::set<int> seen, seen2;
            if (is_box_action(n.get())) p.steps.emplace_back(n->kids[0]->kids[0], n->kids[0]->kids[1]);
            else if (n->k == N_TEMPORAL && n->s == "[]" && !is_temporal(n->kids[0].get(), seen2)) {
                // []P with a state predicate P is an invariance property: TLC checks it as an invariant (Liveness/LiveHourClock.cfg:10
                // TypeInvariance == []HCini, LiveHourClock.tla:23)
                always.emplace_back(name, n->kids[0]);
            } else if (is_temporal(n.get(), seen) || is_fairness(n.get())) live = true;  // liveness (<>, ~>, WF_ / SF_): NOT checked — and said so
            else p.inits.push_back(n);
        }
        if (live) unchecked.push_back(name);
        return p;
    }
    void setup(const std::string &tla_path, const std::string &cfg_text, const std::vector<std::string> &search, bool symmetry) {
        cfg = parse_cfg(cfg_text);
        if (!symmetry) cfg.symmetry.clear();
        const size_t slash = tla_path.rfind('/');
        sp.search.push_back(slash == std::string::npos ? "." : tla_path.substr(0, slash));
        for (auto &s : search) sp.search.push_back(s);
        sp.scoped = cfg.scoped;
        for (auto &c : cfg.constants) sp.const_vals[intern(c.first)] = c.second;
        for (auto &o : cfg.overrides) sp.overrides[intern(o.first)] = intern(o.second);
        sp.load(tla_path);
        sp.finish_load();
        if (!cfg.spec.empty()) split_spec(cfg.spec);
        else if (cfg.init.empty() && cfg.next.empty()) {
            // TLC's "No Behavior Spec" mode (SpecifyingSystems/SimpleMath/SimpleMath.cfg, AsynchronousInterface/PrintValues.cfg; the
            // run-book of serializableSnapshotIsolation.tla:1062-1066 for the in-spec unit tests): no states, the ASSUMEs are evaluated
            no_behavior = true;
            return;
        } else {
            if (cfg.init.empty() || cfg.next.empty()) fail("the configuration names neither a SPECIFICATION nor INIT and NEXT");
            init_node = node_id(cfg.init);
            next_node = node_id(cfg.next);
        }
        init_act = sp.init_tree(init_node);
        for (auto &n : cfg.invariants) invs.emplace_back(n, node_id(n));
        for (auto &n : cfg.constraints) cons.emplace_back(n, node_id(n));
        for (auto &n : cfg.action_constraints) acons.emplace_back(n, node_id(n));
        if (!cfg.view.empty()) view = node_id(cfg.view);
        for (auto &n : cfg.properties) props.push_back(compile_property(n));
        n_cfg_invariants = invs.size();
        for (auto &a : always) invs.push_back(a);  // (after the cfg's INVARIANTs: the indices of those stay what they were)
        if (!cfg.symmetry.empty()) symmetry_group(cfg.symmetry);
    }
    // the group generated by the cfg's SYMMETRY set of permutations (functions on model values)
    void symmetry_group(const std::string &name) {
        V gens_v = sp.ev(node_id(name).get(), nullptr, nullptr, nullptr);
        std::vector<std::vector<std::pair<V, V>>> gens;
        std::vector<V> dom;
        enumerate(gens_v, [&](const V &f) { gens.push_back(fn_items(f)); for (auto &p : gens.back()) dom.push_back(p.first); return true; });
        dom = mk_set(dom)->items;
        auto image = [&](const std::vector<std::pair<V, V>> &g, const V &x) { for (auto &p : g) if (cmp(p.first, x) == 0) return p.second; return x; };
        std::vector<std::vector<V>> grp{dom}, todo{dom};
        auto known = [&](const std::vector<V> &c) { for (auto &g : grp) if (cmp_vec(g, c) == 0) return true; return false; };
        while (!todo.empty()) {
            std::vector<V> a = todo.back();
            todo.pop_back();
            for (auto &g : gens) {
                std::vector<V> c;
                for (auto &x : a) c.push_back(image(g, x));
                if (!known(c)) { grp.push_back(c); todo.push_back(c); }
            }
        }
        std::sort(grp.begin(), grp.end(), [](const std::vector<V> &a, const std::vector<V> &b) { return cmp_vec(a, b) < 0; });
        for (auto &g : grp) { std::map<const Val *, V> m; for (size_t i = 0; i < dom.size(); i++) m[dom[i].get()] = g[i]; group.push_back(m); }
        has_group = true;
    }
    std::string key_of(const State &st) {  // SYMMETRY: the orbit's key is the least image of the state under the group
        std::string best;
        if (view) { ser(sp.ev(view.get(), nullptr, &st, nullptr), best); return best; }
        if (!has_group) { for (auto &v : st) ser(v, best); return best; }
        V bestv;
        for (auto &g : group) {
            std::vector<V> it;
            for (auto &v : st) it.push_back(permute(v, g));
            V t = mk_tuple(it);
            if (!bestv || cmp(t, bestv) < 0) bestv = t;
        }
        ser(bestv, best);
        return best;
    }
    bool in_model(const State &st) { for (auto &c : cons) { V v = sp.ev(c.second.get(), nullptr, &st, nullptr); if (!(v->k == K_BOOL && v->i)) return false; } return true; }
    // ACTION_CONSTRAINT: a transition that does not satisfy it is generated, and its successor checked, but the successor is not stored
    bool in_actions(const State &st, const State &s2) {
        for (auto &c : acons) {
            State nx = s2;
            V v = sp.ev(c.second.get(), nullptr, &st, &nx);
            if (!(v->k == K_BOOL && v->i)) return false;
        }
        return true;
    }
    int violated(const State &st) { for (size_t k = 0; k < invs.size(); k++) { V v = sp.ev(invs[k].second.get(), nullptr, &st, nullptr); if (!(v->k == K_BOOL && v->i)) return (int)k; } return -1; }
    int property_violated_init(const State &st) {
        for (size_t k = 0; k < props.size(); k++)
            for (auto &f : props[k].inits) { V v = sp.ev(f.get(), nullptr, &st, nullptr); if (!(v->k == K_BOOL && v->i)) return (int)k; }
        return -1;
    }
    int property_violated_step(const State &st, const State &s2) {
        for (size_t k = 0; k < props.size(); k++)
            for (auto &s : props[k].steps) {
                if (veq(sp.ev(s.second.get(), nullptr, &st, nullptr), sp.ev(s.second.get(), nullptr, &s2, nullptr))) continue;
                State nx = s2;
                bool any = false;
                sp.act(s.first.get(), nullptr, &st, nx, [&]() { any = true; });
                if (!any) return (int)k;
            }
        return -1;
    }

    void run_assumes(Result &R) {
        const auto t0 = std::chrono::steady_clock::now();
        R.no_behavior = true;
        R.verdict = MC_V_OK;
        g_print_sink = &R.printed;
        struct Unsink { ~Unsink() { g_print_sink = nullptr; } } unsink;
        for (auto &a : sp.assumes) {
            bool ok = false;
            try {
                V v = sp.ev(node_id(a.def).get(), nullptr, nullptr, nullptr);
                ok = v->k == K_BOOL && v->i;
                if (!ok && v->k != K_BOOL) { R.verdict = MC_V_SPECERR; R.error_message = "Assumption line " + std::to_string(a.line) + " of module " + a.module + " is not a Boolean: " + fmt(v); break; }
            } catch (TlaError &e) {
                R.verdict = e.is_assert ? MC_V_ASSERT : MC_V_SPECERR;
                R.error_message = e.is_assert ? e.msg : "evaluating the assumption of line " + std::to_string(a.line) + " of module " + a.module + ": " + e.msg;
                break;
            }
            if (!ok) { R.verdict = MC_V_ASSUME; R.error_message = "Assumption line " + std::to_string(a.line) + " of module " + a.module + " is false."; break; }
            R.assumes_checked++;
        }
        R.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void run(const Options &opt, Result &R) {
        if (no_behavior) { run_assumes(R); return; }
        const auto t0 = std::chrono::steady_clock::now();
        const size_t nv = sp.variables.size();
        std::unordered_map<std::string, int64_t> seen;
        std::vector<State> states;      // every stored state, in discovery order
        std::vector<int64_t> parent;
        std::vector<const GDef *> via;  // the action that produced it (null: an initial state)
        std::vector<int64_t> frontier;
        uint64_t generated = 0;
        int verdict = MC_V_OK, viol = -1;
        std::string viol_name, err_msg;
        std::vector<State> trace;
        std::vector<const GDef *> trace_via;
        const GDef *last_via = nullptr;  // the action that produced the successor being checked
        auto chain = [&](int64_t from, const State *last) {
            std::vector<State> t;
            trace_via.clear();
            for (int64_t i = from; i >= 0; i = parent[(size_t)i]) { t.push_back(states[(size_t)i]); trace_via.push_back(via[(size_t)i]); }
            std::reverse(t.begin(), t.end());
            std::reverse(trace_via.begin(), trace_via.end());
            if (last) { t.push_back(*last); trace_via.push_back(last_via); }
            return t;
        };
        auto note = [&](int kind, int inv, const std::string &name, int64_t from, const State *last) {
            if (verdict != MC_V_OK) return;
            verdict = kind; viol = inv; viol_name = name; trace = chain(from, last);
        };
        auto missing = [&](const State &nx) { std::string m; for (size_t i = 0; i < nv; i++) if (!nx[i]) m += (m.empty() ? "" : ", ") + sp.variables[i]; return m; };
        // ---- initial states
        {
            std::vector<State> inits;
            const State none(nv);
            State nx(nv);
            try {
                sp.act(init_act.get(), nullptr, &none, nx, [&]() {
                    const std::string m = missing(nx);
                    if (!m.empty()) fail("initial state leaves " + m + " unassigned");
                    inits.push_back(nx);
                });
            } catch (TlaError &e) { arena_reset(); throw; }
            arena_reset();
            for (auto &st : inits) {
                generated++;
                const std::string key = key_of(st);
                if (seen.count(key)) continue;
                int k = violated(st);
                if (k >= 0) note(MC_V_INVARIANT, k, invs[(size_t)k].first, -1, &st);
                if (!props.empty() && (k = property_violated_init(st)) >= 0) note(MC_V_INVARIANT, (int)invs.size() + k, props[(size_t)k].name, -1, &st);
                if (!in_model(st)) continue;
                seen.emplace(key, (int64_t)states.size());
                frontier.push_back((int64_t)states.size());
                states.push_back(st);
                parent.push_back(-1);
                via.push_back(nullptr);
                arena_reset();
            }
        }
        R.init_states = frontier.size();
        R.levels.push_back(frontier.size());
        uint32_t depth = 1;
        bool budget = false;
        auto last_report = t0;
        while (!frontier.empty() && verdict == MC_V_OK) {
            if (opt.max_levels && depth >= opt.max_levels) { budget = true; break; }
            if (opt.max_distinct && seen.size() >= opt.max_distinct) { budget = true; break; }
            std::vector<int64_t> fresh;
            size_t done = 0;
            for (int64_t si : frontier) {
                const State st = states[(size_t)si];
                uint64_t nsucc = 0;
                try {
                    std::vector<State> succ;
                    std::vector<const GDef *> succ_via;
                    State nx(nv);
                    sp.branch_errors = true;
                    sp.cur_action = nullptr;
                    sp.first_error.reset();
                    sp.failed_branches = 0;
                    try {
                        sp.act(next_node.get(), nullptr, &st, nx, [&]() {
                            const std::string m = missing(nx);
                            if (!m.empty()) fail("a successor leaves " + m + " unassigned");
                            succ.push_back(nx);
                            succ_via.push_back(sp.cur_action);
                        });
                    } catch (TlaError &e) { if (!sp.first_error) sp.first_error.reset(new TlaError(e)); }
                    sp.cur_action = nullptr;
                    sp.branch_errors = false;
                    arena_reset();
                    generated += sp.failed_branches;
                    if (sp.first_error) {
                        note(sp.first_error->is_assert ? MC_V_ASSERT : MC_V_SPECERR, -1, "", si, nullptr);
                        if (err_msg.empty()) err_msg = sp.first_error->msg;
                    }
                    for (size_t q = 0; q < succ.size(); q++) {
                        const State &s2 = succ[q];
                        last_via = succ_via[q];
                        nsucc++;
                        generated++;
                        if (!props.empty()) { const int k = property_violated_step(st, s2); if (k >= 0) note(MC_V_INVARIANT, (int)invs.size() + k, props[(size_t)k].name, si, &s2); }
                        const std::string key = key_of(s2);
                        if (seen.count(key)) { arena_reset(); continue; }
                        const bool inm = in_model(s2) && in_actions(st, s2);
                        const int k = violated(s2);
                        if (k >= 0) note(MC_V_INVARIANT, k, invs[(size_t)k].first, si, &s2);
                        if (inm) {
                            seen.emplace(key, (int64_t)states.size());
                            fresh.push_back((int64_t)states.size());
                            states.push_back(s2);
                            parent.push_back(si);
                            via.push_back(last_via);
                        }
                        arena_reset();
                    }
                } catch (TlaError &e) {  // (an error while a successor is checked: invariants, constraints, properties)
                    sp.branch_errors = false;
                    arena_reset();
                    note(e.is_assert ? MC_V_ASSERT : MC_V_SPECERR, -1, "", si, nullptr);
                    if (err_msg.empty()) err_msg = e.msg;
                }
                if (!nsucc && opt.check_deadlock && cfg.check_deadlock != 0 && !sp.first_error) note(MC_V_DEADLOCK, -1, "", si, nullptr);
                done++;
                // (an error does not stop the level: like the GPU engine — and like the TLC run of README.md:319-321 — the search ends
                //  when the level the first error was found on has been expanded: counters, queue and depth do not depend on the order
                //  the states of a level are expanded in)
                if (opt.progress_seconds > 0) {
                    const auto now = std::chrono::steady_clock::now();
                    if (std::chrono::duration<double>(now - last_report).count() >= opt.progress_seconds) {
                        last_report = now;
                        printf("Progress(%u): %llu states generated, %llu distinct states found, %llu states left on queue.\n", depth, (unsigned long long)generated,
                               (unsigned long long)seen.size(), (unsigned long long)(frontier.size() - done + fresh.size()));
                        fflush(stdout);
                    }
                }
            }
            frontier = fresh;
            if (!fresh.empty()) { R.levels.push_back(fresh.size()); depth++; }
        }
        if (verdict == MC_V_OK && budget) verdict = MC_V_BUDGET;
        if (!opt.dump_path.empty()) {
            FILE *f = fopen(opt.dump_path.c_str(), "w");
            if (!f) fail("cannot write " + opt.dump_path);
            std::vector<size_t> order;
            for (auto &nm : opt.dump_order) { auto it = sp.varidx.find(intern(nm)); if (it == sp.varidx.end()) fail("dump order: no variable " + nm); order.push_back((size_t)it->second); }
            if (order.empty()) for (size_t i = 0; i < nv; i++) order.push_back(i);
            std::vector<uint32_t> level(states.size(), 0);
            for (size_t i = 0; i < states.size(); i++) {
                level[i] = parent[i] < 0 ? 0 : level[(size_t)parent[i]] + 1;
                std::string line = "L" + std::to_string(level[i]);
                for (size_t k : order) line += " /\\ " + sp.variables[k] + " = " + fmt(states[i][k]);
                fprintf(f, "%s\n", line.c_str());
            }
            fclose(f);
        }
        R.distinct = seen.size();
        R.n_invariants = invs.size();
        R.unchecked_properties = unchecked;
        R.generated = generated;
        R.depth = depth;
        R.verdict = verdict;
        R.violated_invariant = viol;
        R.violated_name = viol_name;
        R.error_message = err_msg;
        R.queue_left = verdict != MC_V_OK ? frontier.size() : 0;
        for (size_t i = 0; i < trace.size(); i++) {
            const GDef *a = i < trace_via.size() ? trace_via[i] : nullptr;
            std::string label = i ? "" : "Initial predicate";
            if (a && a->l1) label = "Action line " + std::to_string(a->l1) + ", col " + std::to_string(a->c1) + " to line " + std::to_string(a->l2) + ", col " +
                                    std::to_string(a->c2) + " of module " + a->module;
            else if (a) label = "Action " + a->name + " of module " + a->module;
            R.trace.emplace_back(label, sp.state_text(trace[i]));
        }
        R.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};

void init_globals() {
    auto t = std::make_shared<Val>(); t->k = K_BOOL; t->i = 1; g_true = t;
    auto f = std::make_shared<Val>(); f->k = K_BOOL; f->i = 0; g_false = f;
    auto e = std::make_shared<Val>(); e->k = K_TUPLE; g_empty_tuple = e;
    auto s = std::make_shared<Val>(); s->k = K_SET; g_empty_set = s;
    g_mvs.clear();
    for (long i = -16; i < 1024; i++) { auto v = std::make_shared<Val>(); v->k = K_INT; v->i = i; g_small_ints[i + 16] = v; }
}

struct Job { const std::string *tla, *cfg; const Options *opt; Result *out; std::string *error; int rc = 0; };
void *job_main(void *p) {
    Job &j = *(Job *)p;
    try {
        init_globals();
        std::string cfg_text;
        if (!read_text(*j.cfg, cfg_text)) { *j.error = "cannot read configuration file " + *j.cfg; j.rc = MC_EPARSE; return nullptr; }
        Checker c;
        c.setup(*j.tla, cfg_text, j.opt->search, j.opt->symmetry);
        c.run(*j.opt, *j.out);
    } catch (SyntaxErr &e) { *j.error = e.msg; j.rc = MC_EPARSE; }
    catch (TlaError &e) { *j.error = e.msg; j.rc = MC_ENOSPEC; }
    catch (std::exception &e) { *j.error = e.what(); j.rc = MC_EBADCFG; }
    arena_reset();
    return nullptr;
}

}  // namespace

int check_files(const std::string &tla_path, const std::string &cfg_path, const Options &opt, Result &out, std::string &error) {
    static std::mutex mu;  // the evaluator keeps its symbol tables and the arena in globals: one run at a time
    std::lock_guard<std::mutex> lock(mu);
    Job j{&tla_path, &cfg_path, &opt, &out, &error};
    if (getenv("TLAEVAL_INLINE")) { job_main(&j); return j.rc; }  // profiling: on the caller's own stack
    // a tree-walking evaluator recurses as deep as the specification nests: run on a thread with a large stack
    for (int shift = 30; shift >= 24; shift -= 2) {  // 1 GB of address space, less where the process may not reserve that much
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setstacksize(&attr, (size_t)1 << shift);
        pthread_t th;
        const int rc = pthread_create(&th, &attr, job_main, &j);
        pthread_attr_destroy(&attr);
        if (rc) continue;
        pthread_join(th, nullptr);
        return j.rc;
    }
    job_main(&j);  // no thread to be had: on the caller's stack
    return j.rc;
}

}  // namespace tlaeval
