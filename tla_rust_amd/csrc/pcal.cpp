// pcal.cpp — PlusCal (p-syntax) parser and pcal2tla-style translator.  See pcal.h.
// Grammar: examples/p-manual.pdf App. A (p-syntax); translation: §3.8 pp.31-32 and App. B pp.60-64.
// Supported: variables (= / \in), multiprocess and uniprocess algorithms, labels, assignment (x := e,
// x[i] := e, a := e || b := e), if / elsif / else, while, either / or, with (\in / =), await / when, assert,
// skip, goto, print, define blocks, macros; both the p-syntax (begin ... end) and the c-syntax ({ ... }).
// procedure / call / return: parsed here and expanded into the calling processes (expand_procedures below; pcal.h says what
// that keeps of pcal2tla's translation and what not).
#include "pcal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <set>

namespace pcal {
namespace {

// ------------------------------------------------------------------------------------------ lexer
struct Tok {
    enum T { END, IDENT, NUM, STR, SYM, SEP } t = END;  // SEP: a ---- or ==== line
    std::string s;
    int line = 0, col = 0;
};

const char *kSyms[] = {"|->", ":=", "||", "==", "<=>", "=>", "<=", ">=", "=<", "/=", "/\\", "\\/", "..", "->", "<<", ">>", "(", ")", "[",
                       "]",   "{",  "}",  ",",  ";",  ":",  "+",  "-",  "*",  "%",   "=",   "<",  ">",  "#",  "~",  "'", "!", "@",
                       ".",   "^",  "\\"};

struct LexError { std::string msg; };

// Tokens of text[b, e) (comments skipped); lines and columns are 1-based positions in the whole file.
std::vector<Tok> lex(const std::string &text, size_t b, size_t e) {
    std::vector<Tok> out;
    int line = 1, col = 1;
    for (size_t i = 0; i < b; i++) { if (text[i] == '\n') { line++; col = 1; } else col++; }
    size_t p = b;
    auto adv = [&](size_t n) { for (size_t i = 0; i < n && p < e; i++, p++) { if (text[p] == '\n') { line++; col = 1; } else col++; } };
    auto idch = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; };
    while (p < e) {
        const char c = text[p];
        if (c == ' ' || c == '\t' || c == '\r' || c == '\n') { adv(1); continue; }
        if (c == '\\' && p + 1 < e && text[p + 1] == '*') { while (p < e && text[p] != '\n') adv(1); continue; }
        if (c == '(' && p + 1 < e && text[p + 1] == '*') {
            int depth = 1;
            adv(2);
            while (p < e && depth) {
                if (p + 1 < e && text[p] == '(' && text[p + 1] == '*') { depth++; adv(2); }
                else if (p + 1 < e && text[p] == '*' && text[p + 1] == ')') { depth--; adv(2); }
                else adv(1);
            }
            continue;
        }
        Tok t;
        t.line = line;
        t.col = col;
        if ((c == '-' || c == '=') && p + 3 < e && text[p + 1] == c && text[p + 2] == c && text[p + 3] == c) {
            size_t q = p;
            while (q < e && text[q] == c) q++;
            t.t = Tok::SEP;
            t.s.assign(text, p, q - p);
            adv(q - p);
            out.push_back(t);
            continue;
        }
        if (idch(c)) {
            size_t q = p;
            while (q < e && idch(text[q])) q++;
            t.s.assign(text, p, q - p);
            bool digits = true;
            for (char ch : t.s) digits &= ch >= '0' && ch <= '9';
            t.t = digits ? Tok::NUM : Tok::IDENT;
            adv(q - p);
            out.push_back(t);
            continue;
        }
        if (c == '"') {
            size_t q = p + 1;
            while (q < e && text[q] != '"' && text[q] != '\n') q++;
            if (q >= e || text[q] != '"') throw LexError{"unterminated string at line " + std::to_string(line)};
            t.t = Tok::STR;
            t.s.assign(text, p + 1, q - p - 1);
            adv(q + 1 - p);
            out.push_back(t);
            continue;
        }
        if (c == '\\') {  // \in \notin \div \A \E \cup ...
            size_t q = p + 1;
            while (q < e && ((text[q] >= 'a' && text[q] <= 'z') || (text[q] >= 'A' && text[q] <= 'Z'))) q++;
            if (q > p + 1) {
                t.t = Tok::SYM;
                t.s.assign(text, p, q - p);
                adv(q - p);
                out.push_back(t);
                continue;
            }
        }
        bool found = false;
        for (const char *sy : kSyms) {
            const size_t n = strlen(sy);
            if (p + n <= e && text.compare(p, n, sy) == 0) {
                t.t = Tok::SYM;
                t.s = sy;
                adv(n);
                out.push_back(t);
                found = true;
                break;
            }
        }
        if (!found) throw LexError{std::string("unexpected character '") + c + "' at line " + std::to_string(line) + ", column " + std::to_string(col)};
    }
    Tok end;
    end.line = line;
    end.col = col;
    out.push_back(end);
    return out;
}

// two field paths of one record variable ("" = the whole record, "f", "f.g") name overlapping parts when one is a prefix of the other
static bool paths_overlap(const std::string &a, const std::string &b) {
    if (a.empty() || b.empty() || a == b) return true;
    const std::string &sh = a.size() < b.size() ? a : b, &lg = a.size() < b.size() ? b : a;
    return lg.compare(0, sh.size(), sh) == 0 && lg[sh.size()] == '.';
}

// ------------------------------------------------------------------------------------------ parser
struct ParseError { std::string msg; };

struct Parser {
    std::vector<Tok> t;
    size_t i = 0;
    explicit Parser(std::vector<Tok> toks) : t(std::move(toks)) { fresh_counter() = 0; }
    const Tok &cur() const { return t[i]; }
    const Tok &peek(size_t k = 1) const { return t[std::min(i + k, t.size() - 1)]; }
    bool is_sym(const char *s) const { return cur().t == Tok::SYM && cur().s == s; }
    bool is_id(const char *s) const { return cur().t == Tok::IDENT && cur().s == s; }
    [[noreturn]] void fail(const std::string &what) const {
        throw ParseError{what + " at line " + std::to_string(cur().line) + ", column " + std::to_string(cur().col) +
                         (cur().t == Tok::END ? " (end of text)" : " (near `" + cur().s + "`)")};
    }
    void expect_sym(const char *s) { if (!is_sym(s)) fail(std::string("expected `") + s + "`"); i++; }
    void expect_id(const char *s) { if (!is_id(s)) fail(std::string("expected `") + s + "`"); i++; }
    std::string ident(const char *what) {
        if (cur().t != Tok::IDENT) fail(std::string("expected ") + what);
        return t[i++].s;
    }

    // ---- expressions (TLA+ operator precedences, Specifying Systems table 6)
    static int infix_prec(const Tok &k, bool &right) {
        right = false;
        if (k.t != Tok::SYM) return -1;
        const std::string &s = k.s;
        if (s == "=>" || s == "<=>" || s == "\\equiv") return 1;
        if (s == "\\/" || s == "\\lor") return 3;
        if (s == "/\\" || s == "\\land") return 3;
        if (s == "=" || s == "#" || s == "/=" || s == "<" || s == ">" || s == "<=" || s == "=<" || s == ">=" || s == "\\leq" || s == "\\geq" ||
            s == "\\in" || s == "\\notin" || s == "\\subseteq")
            return 5;
        if (s == "\\cup" || s == "\\union" || s == "\\cap" || s == "\\intersect" || s == "\\") return 8;
        if (s == "..") return 9;
        if (s == "+" || s == "-") return 10;
        if (s == "%") return 11;
        if (s == "*" || s == "\\div" || s == "\\o" || s == "\\circ") return 13;
        return -1;
    }
    static EP mk(Expr::K k, const Tok &at) {
        auto e = std::make_shared<Expr>();
        e->k = k;
        e->pos = {at.line, at.col};
        return e;
    }
    struct LetDef { std::string name; std::vector<std::string> params; EP body; };
    // e with the LET definitions substituted: a name without parameters by its body, f(args) by f's body with the parameters replaced;
    // `hidden` = names re-bound by a quantifier / function constructor on the way down
    // identifiers that occur FREE in e (operator names of calls included): what a binder at the place of substitution could capture
    static void free_ids(const EP &e, std::set<std::string> bound, std::set<std::string> &out) {
        if (!e) return;
        if ((e->k == Expr::ID || e->k == Expr::CALL) && !bound.count(e->s)) out.insert(e->s);
        if ((e->k == Expr::QUANT || e->k == Expr::FUNCDEF || e->k == Expr::SETOF) && !e->bound.empty()) {
            if (!e->a.empty()) free_ids(e->a[0], bound, out);   // the domain is outside the binding
            bound.insert(e->bound);
            for (size_t j = 1; j < e->a.size(); j++) free_ids(e->a[j], bound, out);
            return;
        }
        for (auto &x : e->a) free_ids(x, bound, out);
    }
    static int &fresh_counter() { static thread_local int n = 0; return n; }
    // e with the LET definitions substituted: a name without parameters by its body, f(args) by f's body with the parameters replaced;
    // `hidden` = names re-bound by a quantifier / function constructor on the way down.  CAPTURE-AVOIDING: a binder
    // whose variable occurs free in a body that is substituted below it — `LET f(a) == \E y \in S : y + 1 = a IN \E y \in T : f(y)`, or
    // `LET a == y + 1 IN \E y \in S : a` — is renamed (y_1, y_2, ...) in its own scope first, so that the substituted `y` keeps meaning
    // the outer one; the translation then prints the renamed binder, which SANY accepts.
    static EP let_subst(const EP &e, const std::vector<LetDef> &defs, std::set<std::string> hidden) {
        if (!e) return e;
        if (e->k == Expr::ID && !hidden.count(e->s))
            for (size_t k = defs.size(); k-- > 0;)
                if (defs[k].name == e->s && defs[k].params.empty()) return defs[k].body;
        if (e->k == Expr::CALL && !hidden.count(e->s))
            for (size_t k = defs.size(); k-- > 0;)
                if (defs[k].name == e->s && defs[k].params.size() == e->a.size() && !defs[k].params.empty()) {
                    std::vector<LetDef> args;
                    for (size_t j = 0; j < e->a.size(); j++) {
                        EP a = let_subst(e->a[j], defs, hidden);
                        a = std::make_shared<Expr>(*a);
                        a->paren = true;
                        args.push_back({defs[k].params[j], {}, a});
                    }
                    return let_subst(defs[k].body, args, {});
                }
        auto c = std::make_shared<Expr>(*e);
        if ((e->k == Expr::QUANT || e->k == Expr::FUNCDEF || e->k == Expr::SETOF) && !e->bound.empty()) {
            if (!c->a.empty()) c->a[0] = let_subst(e->a[0], defs, hidden);   // the domain is outside the binding
            bool clash = false;
            for (const auto &d : defs) {
                if (hidden.count(d.name) || d.name == e->bound) continue;   // (a definition the binder itself shadows is not substituted below)
                std::set<std::string> fr, bnd(d.params.begin(), d.params.end());
                free_ids(d.body, bnd, fr);
                if (fr.count(e->bound)) { clash = true; break; }
            }
            std::vector<EP> scope(e->a.begin() + (e->a.empty() ? 0 : 1), e->a.end());
            if (clash) {
                auto id = std::make_shared<Expr>();
                id->k = Expr::ID;
                id->pos = e->pos;
                id->s = e->bound + "_" + std::to_string(++fresh_counter());
                c->bound = id->s;
                const std::vector<LetDef> ren{{e->bound, {}, id}};
                for (auto &x : scope) x = let_subst(x, ren, {});
            } else {
                hidden.insert(e->bound);
            }
            for (size_t j = 1; j < c->a.size(); j++) c->a[j] = let_subst(scope[j - 1], defs, hidden);
            return c;
        }
        for (auto &x : c->a) x = let_subst(x, defs, hidden);
        return c;
    }
    EP primary() {
        const Tok k = cur();
        if (k.t == Tok::NUM) { i++; auto e = mk(Expr::NUM, k); e->num = atoll(k.s.c_str()); return e; }
        if (k.t == Tok::STR) { i++; auto e = mk(Expr::STR, k); e->s = k.s; return e; }
        if (k.t == Tok::IDENT) {
            if (k.s == "TRUE" || k.s == "FALSE") { i++; auto e = mk(Expr::BOOL, k); e->num = k.s == "TRUE"; return e; }
            if (k.s == "IF") {
                i++;
                auto e = mk(Expr::IF, k);
                e->a.push_back(expr(0));
                expect_id("THEN");
                e->a.push_back(expr(0));
                expect_id("ELSE");
                e->a.push_back(expr(0));
                return e;
            }
            if (k.s == "CASE") {   // CASE p1 -> e1 [] p2 -> e2 [] OTHER -> e: the first arm whose guard holds; without OTHER, no arm is an evaluation error
                i++;
                std::vector<std::pair<EP, EP>> arms;
                EP other;
                for (;;) {
                    if (is_id("OTHER")) { i++; expect_sym("->"); other = expr(2); break; }
                    EP g = expr(2);
                    expect_sym("->");
                    arms.push_back({g, expr(2)});
                    if (is_sym("[") && peek().t == Tok::SYM && peek().s == "]") { i += 2; continue; }
                    break;
                }
                if (!other) {   // (what TLC raises an error for: CHOOSE from the empty set does the same, in every back-end)
                    other = mk(Expr::QUANT, k);
                    other->s = "CHOOSE";
                    other->bound = "none_";
                    auto dom = mk(Expr::BINOP, k);
                    dom->s = "..";
                    auto one = mk(Expr::NUM, k), zero = mk(Expr::NUM, k);
                    one->num = 1;
                    dom->a = {one, zero};
                    auto yes = mk(Expr::BOOL, k);
                    yes->num = 1;
                    other->a = {dom, yes};
                    other->paren = true;
                }
                EP acc = other;
                for (size_t a = arms.size(); a-- > 0;) {
                    auto e = mk(Expr::IF, k);
                    e->a = {arms[a].first, arms[a].second, acc};
                    e->paren = true;
                    acc = e;
                }
                return acc;
            }
            if (k.s == "DOMAIN") { i++; auto e = mk(Expr::UNOP, k); e->s = "DOMAIN"; e->a.push_back(postfix()); return e; }
            if (k.s == "LET") {   // LET a == e  f(x, y) == g ... IN body: substituted where it is parsed (a definition sees the earlier ones;
                i++;              //  like TLC's, an unused or guarded definition is never evaluated); the translation prints the result
                std::vector<LetDef> defs;
                for (;;) {
                    LetDef d;
                    d.name = ident("a name after LET");
                    if (is_sym("(")) {
                        i++;
                        for (;;) {
                            d.params.push_back(ident("a parameter name"));
                            if (is_sym(",")) { i++; continue; }
                            break;
                        }
                        expect_sym(")");
                    }
                    expect_sym("==");
                    d.body = let_subst(expr(0), defs, {});
                    d.body = std::make_shared<Expr>(*d.body);
                    d.body->paren = true;
                    defs.push_back(d);
                    if (cur().t == Tok::IDENT && cur().s == "IN") break;
                    if (cur().t != Tok::IDENT) fail("expected another definition or IN");
                }
                expect_id("IN");
                EP body = let_subst(expr(0), defs, {});
                body = std::make_shared<Expr>(*body);
                body->paren = true;
                return body;
            }
            if (k.s == "CHOOSE") {  // bounded CHOOSE x \in S : P (examples/p-manual.pdf section 2.4: the definition of gcd)
                i++;
                auto e = mk(Expr::QUANT, k);
                e->s = "CHOOSE";
                e->bound = ident("a bound variable");
                expect_sym("\\in");
                e->a.push_back(expr(6));
                expect_sym(":");
                e->a.push_back(expr(0));
                return e;
            }
            i++;
            if (is_sym("(") && cur().line == k.line && cur().col == k.col + (int)k.s.size()) {  // Op(args): no space before (
                i++;
                auto e = mk(Expr::CALL, k);
                e->s = k.s;
                if (!is_sym(")")) for (;;) { e->a.push_back(expr(0)); if (is_sym(",")) { i++; continue; } break; }
                expect_sym(")");
                return e;
            }
            auto e = mk(Expr::ID, k);
            e->s = k.s;
            return e;
        }
        if (k.t == Tok::SYM) {
            if (k.s == "(") { i++; EP e = expr(0); expect_sym(")"); e->paren = true; return e; }
            if (k.s == "{") {
                i++;
                auto e = mk(Expr::SETENUM, k);
                if (!is_sym("}")) {
                    EP first = expr(0);
                    if (is_sym(":")) {   // {x \\in S : P} or {e : x \\in S}
                        i++;
                        auto so = mk(Expr::SETOF, k);
                        if (first->k == Expr::BINOP && first->s == "\\in" && first->a[0]->k == Expr::ID && !first->paren) {
                            so->s = "filter";
                            so->bound = first->a[0]->s;
                            so->a = {first->a[1], expr(0)};
                        } else {
                            so->s = "map";
                            so->bound = ident("a bound variable");
                            expect_sym("\\in");
                            so->a = {expr(6), first};
                        }
                        expect_sym("}");
                        return so;
                    }
                    e->a.push_back(first);
                    while (is_sym(",")) { i++; e->a.push_back(expr(0)); }
                }
                expect_sym("}");
                return e;
            }
            if (k.s == "<<") {
                i++;
                auto e = mk(Expr::TUPLE, k);
                if (!is_sym(">>")) for (;;) { e->a.push_back(expr(0)); if (is_sym(",")) { i++; continue; } break; }
                expect_sym(">>");
                return e;
            }
            if (k.s == "[" && peek().t == Tok::IDENT && i + 2 < t.size() && t[i + 2].t == Tok::SYM && t[i + 2].s == "|->") {  // [f |-> e, g |-> h]
                i++;
                auto e = mk(Expr::RECORD, k);
                for (;;) {
                    const std::string f = ident("a field name");
                    for (const auto &n : e->names) if (n == f) fail("field " + f + " appears twice in the record");
                    e->names.push_back(f);
                    expect_sym("|->");
                    e->a.push_back(expr(0));
                    if (is_sym(",")) { i++; continue; }
                    break;
                }
                expect_sym("]");
                return e;
            }
            if (k.s == "[") {  // [x \in S |-> e]
                i++;
                auto e = mk(Expr::FUNCDEF, k);
                e->bound = ident("a bound variable after `[`");
                expect_sym("\\in");
                e->a.push_back(expr(6));
                expect_sym("|->");
                e->a.push_back(expr(0));
                expect_sym("]");
                return e;
            }
            if (k.s == "\\A" || k.s == "\\E") {
                i++;
                auto e = mk(Expr::QUANT, k);
                e->s = k.s;
                e->bound = ident("a bound variable");
                expect_sym("\\in");
                e->a.push_back(expr(6));
                expect_sym(":");
                e->a.push_back(expr(0));
                return e;
            }
            if (k.s == "~" || k.s == "\\lnot" || k.s == "\\neg") { i++; auto e = mk(Expr::UNOP, k); e->s = "~"; e->a.push_back(expr(4)); return e; }
            if (k.s == "-") { i++; auto e = mk(Expr::UNOP, k); e->s = "-"; e->a.push_back(expr(12)); return e; }
            if (k.s == "/\\" || k.s == "\\/") {  // bulleted list: items separated by the same bullet
                const std::string op = k.s;
                i++;
                EP acc = expr(4);
                while (is_sym(op.c_str())) {
                    const Tok at = cur();
                    i++;
                    auto b = mk(Expr::BINOP, at);
                    b->s = op;
                    b->a = {acc, expr(4)};
                    acc = b;
                }
                acc->paren = true;
                return acc;
            }
        }
        fail("expected an expression");
    }
    EP postfix() {
        EP e = primary();
        for (;;) {
            if (is_sym("[") && !(peek().t == Tok::SYM && peek().s == "]")) {   // (`[]` separates the arms of a CASE)
                const Tok at = cur();
                i++;
                auto x = mk(Expr::INDEX, at);
                x->a = {e, expr(0)};
                expect_sym("]");
                e = x;
            } else if (is_sym(".") && peek().t == Tok::IDENT) {  // r.f
                const Tok at = cur();
                i++;
                auto x = mk(Expr::DOT, at);
                x->s = t[i++].s;
                x->a = {e};
                e = x;
            } else if (is_sym("'")) {
                const Tok at = cur();
                i++;
                auto x = mk(Expr::PRIME, at);
                x->a = {e};
                e = x;
            } else {
                return e;
            }
        }
    }
    EP expr(int minprec) {
        EP lhs = postfix();
        for (;;) {
            bool right = false;
            const int p = infix_prec(cur(), right);
            if (p < 0 || p < minprec) return lhs;
            const Tok op = cur();
            i++;
            EP rhs = expr(p + 1);
            auto b = mk(Expr::BINOP, op);
            b->s = op.s == "\\land" ? "/\\" : op.s == "\\lor" ? "\\/" : op.s == "=<" || op.s == "\\leq" ? "<=" : op.s == "\\geq" ? ">=" : op.s == "/=" ? "#" : op.s == "\\equiv" ? "<=>" : op.s;
            b->pos = lhs->pos;
            b->a = {lhs, rhs};
            lhs = b;
        }
    }

    // ---- PlusCal
    static bool decl_end_keyword(const std::string &s) {
        return s == "begin" || s == "process" || s == "fair" || s == "define" || s == "macro" || s == "procedure";
    }
    std::vector<VarDecl> vardecls() {
        std::vector<VarDecl> v;
        while (cur().t == Tok::IDENT && !decl_end_keyword(cur().s)) {  // (a `{` is not an IDENT: it ends the list in c-syntax)
            VarDecl d;
            d.pos = {cur().line, cur().col};
            d.name = t[i++].s;
            if (is_sym("=")) { i++; d.init = expr(0); }
            else if (is_sym("\\in")) { i++; d.in_set = true; d.init = expr(0); }
            else {  // `variable x;` — pcal2tla initialises it to the model value defaultInitValue (p-manual section 3.3)
                d.no_init = true;
                auto e = std::make_shared<Expr>();
                e->k = Expr::ID;
                e->s = "defaultInitValue";
                e->pos = d.pos;
                d.init = e;
            }
            v.push_back(d);
            if (is_sym(",") || is_sym(";")) i++;
        }
        return v;
    }
    bool at_block_end() const {
        return cur().t == Tok::END || is_id("end") || is_id("else") || is_id("elsif") || is_id("or");
    }
    std::vector<Macro> *macros = nullptr;
    bool in_macro = false;
    std::vector<SP> stmts() {
        std::vector<SP> v;
        while (!at_block_end()) {
            if (!macro_call(v)) v.push_back(stmt());
            if (is_sym(";")) i++;
            else if (!at_block_end()) fail("expected `;`");
        }
        return v;
    }
    // ---- macros: expanded where they are called (p-manual §3.4); arguments are substituted as expressions
    static EP subst(const EP &e, const std::map<std::string, EP> &m) {
        if (!e) return e;
        if (e->k == Expr::ID) {
            auto it = m.find(e->s);
            if (it == m.end()) return e;
            auto c = std::make_shared<Expr>(*it->second);
            if (c->k == Expr::BINOP || c->k == Expr::IF || c->k == Expr::QUANT) c->paren = true;
            return c;
        }
        auto c = std::make_shared<Expr>(*e);
        for (auto &x : c->a) x = subst(x, m);
        return c;
    }
    SP subst(const SP &s, const std::map<std::string, EP> &m) {
        auto c = std::make_shared<Stmt>(*s);
        if (!c->label.empty()) fail("labels are not allowed inside a macro");
        c->e = subst(s->e, m);
        c->idx = subst(s->idx, m);
        if (s->k == Stmt::ASSIGN) {
            auto it = m.find(s->var);
            if (it != m.end()) {
                const EP &a = it->second;
                if (a->k == Expr::ID) c->var = a->s;
                else if (a->k == Expr::INDEX && a->a[0]->k == Expr::ID && !c->idx) { c->var = a->a[0]->s; c->idx = a->a[1]; }
                else fail("a macro parameter that is assigned must be instantiated with a variable");
            }
        }
        for (auto &b : c->blocks) for (auto &x : b) x = subst(x, m);
        for (auto &x : c->more) x = subst(x, m);
        return c;
    }
    bool macro_call(std::vector<SP> &out) {
        size_t j = i;
        std::string label;
        if (t[j].t == Tok::IDENT && t[j + 1].t == Tok::SYM && t[j + 1].s == ":" && j + 2 < t.size()) { label = t[j].s; j += 2; if (t[j].t == Tok::SYM && (t[j].s == "+" || t[j].s == "-")) j++; }
        if (!macros || t[j].t != Tok::IDENT || !(t[j + 1].t == Tok::SYM && t[j + 1].s == "(")) return false;
        const Macro *mac = nullptr;
        for (const auto &m : *macros) if (m.name == t[j].s) mac = &m;
        if (!mac) return false;
        if (in_macro) fail("a macro cannot call a macro");
        const Pos at{t[j].line, t[j].col};
        i = j + 2;
        std::vector<EP> args;
        if (!is_sym(")")) for (;;) { args.push_back(expr(0)); if (is_sym(",")) { i++; continue; } break; }
        expect_sym(")");
        if (args.size() != mac->params.size()) fail("macro " + mac->name + " takes " + std::to_string(mac->params.size()) + " arguments");
        std::map<std::string, EP> m;
        for (size_t k = 0; k < args.size(); k++) m[mac->params[k]] = args[k];
        bool first = true;
        for (const auto &s : mac->body) {
            SP c = subst(s, m);
            c->pos = s->k == Stmt::ASSERT ? s->pos : at;
            if (first) c->label = label;
            first = false;
            out.push_back(c);
        }
        if (mac->body.empty() && !label.empty()) fail("a labeled call of an empty macro");
        return true;
    }
    // after `if` / `elsif`: condition, then-block and the else part (an elsif chain nests); `end if` is left to the caller
    void if_tail(Stmt &s) {
        s.e = expr(0);
        expect_id("then");
        s.blocks.push_back(stmts());
        if (is_id("elsif")) {
            auto inner = std::make_shared<Stmt>();
            inner->k = Stmt::IF;
            inner->pos = {cur().line, cur().col};
            i++;
            if_tail(*inner);
            s.blocks.push_back({inner});
        } else if (is_id("else")) {
            i++;
            s.blocks.push_back(stmts());
        } else {
            s.blocks.push_back({});
        }
    }
    SP stmt() {
        auto s = std::make_shared<Stmt>();
        if (cur().t == Tok::IDENT && peek().t == Tok::SYM && peek().s == ":") {
            s->label = t[i].s;
            i += 2;
            if (is_sym("+") || is_sym("-")) i++;  // fairness modifiers carry no meaning for safety checking
        }
        s->pos = {cur().line, cur().col};
        if (cur().t != Tok::IDENT) fail("expected a statement");
        const std::string kw = cur().s;
        if (kw == "if") {
            i++;
            s->k = Stmt::IF;
            if_tail(*s);
            expect_id("end");
            expect_id("if");
            return s;
        }
        if (kw == "while") {
            i++;
            s->k = Stmt::WHILE;
            s->e = expr(0);
            expect_id("do");
            s->blocks.push_back(stmts());
            expect_id("end");
            expect_id("while");
            return s;
        }
        if (kw == "either") {
            i++;
            s->k = Stmt::EITHER;
            s->blocks.push_back(stmts());
            while (is_id("or")) { i++; s->blocks.push_back(stmts()); }
            if (s->blocks.size() < 2) fail("`either` needs at least one `or`");
            expect_id("end");
            expect_id("either");
            return s;
        }
        if (kw == "with") {
            i++;
            s->k = Stmt::WITH;
            s->var = ident("a variable after `with`");
            if (is_sym("=")) { i++; s->with_eq = true; }
            else expect_sym("\\in");
            s->e = expr(0);
            if (is_sym(",") || is_sym(";")) fail("`with` over several variables is not supported: nest the statements");
            expect_id("do");
            s->blocks.push_back(stmts());
            expect_id("end");
            expect_id("with");
            return s;
        }
        simple(*s);
        return s;
    }
    // await / assert / skip / goto / print / assignment: the same in both syntaxes
    void simple(Stmt &s) {
        const std::string kw = cur().s;
        if (kw == "await" || kw == "when") { i++; s.k = Stmt::AWAIT; s.e = expr(0); return; }
        if (kw == "assert") { i++; s.k = Stmt::ASSERT; s.e = expr(0); return; }
        if (kw == "skip") { i++; s.k = Stmt::SKIP; return; }
        if (kw == "goto") { i++; s.k = Stmt::GOTO; s.var = ident("a label after `goto`"); return; }
        if (kw == "print") { i++; s.k = Stmt::PRINT; s.e = expr(0); return; }
        if (kw == "call") {  // call P(e1, ..., en)   (p-manual section 3.5)
            if (in_macro) fail("a macro cannot contain a `call`");
            i++;
            s.k = Stmt::CALL;
            s.var = ident("a procedure name after `call`");
            expect_sym("(");
            if (!is_sym(")")) for (;;) { s.args.push_back(expr(0)); if (is_sym(",")) { i++; continue; } break; }
            expect_sym(")");
            return;
        }
        if (kw == "return") {
            if (in_macro) fail("a macro cannot contain a `return`");
            i++;
            s.k = Stmt::RETURN;
            return;
        }
        s.k = Stmt::ASSIGN;
        s.var = t[i++].s;
        if (is_sym("[")) { i++; s.idx = expr(0); expect_sym("]"); if (is_sym("[")) fail("only one index level is supported on the left of `:=`"); }
        s.field = field_path();
        expect_sym(":=");
        s.e = expr(0);
        while (is_sym("||")) {  // a := e || b := f: simultaneous
            i++;
            auto o = std::make_shared<Stmt>();
            o->k = Stmt::ASSIGN;
            o->pos = {cur().line, cur().col};
            o->var = ident("a variable after `||`");
            if (is_sym("[")) { i++; o->idx = expr(0); expect_sym("]"); }
            o->field = field_path();
            expect_sym(":=");
            o->e = expr(0);
            // (two FIELDS of one record, `r.f := a || r.g := b`, are two variables here: pcal.h, RECORDS; with nested records two
            //  paths collide when one is a prefix of the other: `r.f := .. || r.f.g := ..`)
            auto same = [&](const Stmt &x) { return x.var == o->var && paths_overlap(x.field, o->field); };
            if (same(s)) fail("`||` with two assignments to " + s.var + " is not supported");
            for (const auto &x : s.more) if (same(*x)) fail("`||` with two assignments to " + o->var + " is not supported");
            s.more.push_back(o);
        }
    }

    // the field path on the left of `:=`: "" | "f" | "f.g" ... (r.f.g := e, r[i].f.g := e — nested records, pcal.h RECORDS)
    std::string field_path() {
        std::string path;
        while (is_sym(".")) {
            i++;
            path += (path.empty() ? "" : ".") + ident("a field name after `.`");
            if (is_sym("[")) fail("only `r.f...` and `r[i].f...` are supported on the left of `:=`");
        }
        return path;
    }
    // ---- c-syntax (p-manual App. A): braces instead of begin/end, tests in parentheses; same AST
    std::vector<SP> c_block() {  // { stmt; stmt; ... }
        expect_sym("{");
        std::vector<SP> v;
        while (!is_sym("}")) {
            if (cur().t == Tok::END) fail("missing `}`");
            c_stmt(v);
        }
        i++;
        if (is_sym(";")) i++;
        return v;
    }
    std::vector<SP> c_body() {  // a statement used as the body of if / while / either / with
        if (is_sym("{")) return c_block();
        std::vector<SP> v;
        c_stmt(v);
        return v;
    }
    void c_if_tail(Stmt &s) {
        expect_sym("(");
        s.e = expr(0);
        expect_sym(")");
        s.blocks.push_back(c_body());
        if (is_id("else")) {
            i++;
            if (is_id("if")) {  // else if: nests like elsif
                auto inner = std::make_shared<Stmt>();
                inner->k = Stmt::IF;
                inner->pos = {cur().line, cur().col};
                i++;
                c_if_tail(*inner);
                s.blocks.push_back({inner});
            } else {
                s.blocks.push_back(c_body());
            }
        } else {
            s.blocks.push_back({});
        }
    }
    void c_stmt(std::vector<SP> &out) {
        if (macro_call(out)) { if (is_sym(";")) i++; return; }
        auto s = std::make_shared<Stmt>();
        if (cur().t == Tok::IDENT && peek().t == Tok::SYM && peek().s == ":") {
            s->label = t[i].s;
            i += 2;
            if (is_sym("+") || is_sym("-")) i++;
        }
        s->pos = {cur().line, cur().col};
        if (is_sym("{")) {  // a compound statement in a sequence: its statements join the sequence
            std::vector<SP> inner = c_block();
            if (inner.empty()) fail("empty compound statement");
            if (!s->label.empty()) { if (!inner[0]->label.empty()) fail("two labels on one statement"); inner[0]->label = s->label; }
            for (auto &x : inner) out.push_back(x);
            return;
        }
        if (cur().t != Tok::IDENT) fail("expected a statement");
        const std::string kw = cur().s;
        if (kw == "if") { i++; s->k = Stmt::IF; c_if_tail(*s); out.push_back(s); return; }
        if (kw == "while") {
            i++;
            s->k = Stmt::WHILE;
            expect_sym("(");
            s->e = expr(0);
            expect_sym(")");
            s->blocks.push_back(c_body());
            out.push_back(s);
            return;
        }
        if (kw == "either") {
            i++;
            s->k = Stmt::EITHER;
            s->blocks.push_back(c_body());
            while (is_id("or")) { i++; s->blocks.push_back(c_body()); }
            if (s->blocks.size() < 2) fail("`either` needs at least one `or`");
            out.push_back(s);
            return;
        }
        if (kw == "with") {
            i++;
            s->k = Stmt::WITH;
            expect_sym("(");
            s->var = ident("a variable after `with (`");
            if (is_sym("=")) { i++; s->with_eq = true; }
            else expect_sym("\\in");
            s->e = expr(0);
            if (is_sym(",") || is_sym(";")) fail("`with` over several variables is not supported: nest the statements");
            expect_sym(")");
            s->blocks.push_back(c_body());
            out.push_back(s);
            return;
        }
        simple(*s);
        if (is_sym(";")) i++;
        else if (!is_sym("}") && !is_id("else") && !is_id("or")) fail("expected `;`");
        out.push_back(s);
    }
    void c_algorithm(Module &m) {  // positioned at the `{` after the algorithm name
        expect_sym("{");
        if (is_id("variables") || is_id("variable")) { i++; m.globals = vardecls(); }
        if (is_id("define")) {
            i++;
            expect_sym("{");
            while (!is_sym("}")) define_one(m);
            i++;
            if (is_sym(";")) i++;
        }
        while (is_id("macro")) {
            i++;
            Macro mac;
            mac.name = ident("a macro name");
            expect_sym("(");
            if (!is_sym(")")) for (;;) { mac.params.push_back(ident("a parameter name")); if (is_sym(",")) { i++; continue; } break; }
            expect_sym(")");
            in_macro = true;
            mac.body = c_block();
            in_macro = false;
            m.macros.push_back(mac);
        }
        macros = &m.macros;
        procedures(m, true);
        if (is_sym("{")) {  // uniprocess
            Proc p;
            p.body = c_block();
            m.procs.push_back(p);
        } else {
            while (is_id("process") || is_id("fair")) {
                if (is_id("fair")) { i++; if (is_sym("+")) i++; }
                expect_id("process");
                expect_sym("(");
                Proc p;
                p.name = ident("a process name");
                if (is_sym("=")) { i++; }
                else { expect_sym("\\in"); p.is_set = true; }
                p.id = expr(0);
                expect_sym(")");
                if (is_id("variables") || is_id("variable")) { i++; p.locals = vardecls(); }
                p.body = c_block();
                m.procs.push_back(p);
            }
            if (m.procs.empty()) fail("expected `{` or `process`");
        }
        expect_sym("}");
    }
    // procedure P(a, b = e) [variables x = e, y;] begin ... end procedure [;]      /      ... { ... }   in the c-syntax
    void procedures(Module &m, bool c_syntax) {
        while (is_id("procedure")) {
            Procedure pr;
            pr.pos = {cur().line, cur().col};
            i++;
            pr.name = ident("a procedure name");
            expect_sym("(");
            while (!is_sym(")")) {
                VarDecl d;
                d.pos = {cur().line, cur().col};
                d.name = ident("a parameter name");
                if (is_sym("=")) { i++; d.init = expr(0); }
                else {
                    d.no_init = true;
                    auto e = std::make_shared<Expr>();
                    e->k = Expr::ID;
                    e->s = "defaultInitValue";
                    e->pos = d.pos;
                    d.init = e;
                }
                pr.params.push_back(d);
                if (is_sym(",")) i++;
                else if (!is_sym(")")) fail("expected `,` or `)` in the parameter list");
            }
            i++;
            if (is_id("variables") || is_id("variable")) {
                i++;
                pr.locals = vardecls();
                for (const auto &d : pr.locals) if (d.in_set) fail("a procedure variable is initialised with `=`, not `\\in` (p-manual section 3.5)");
            }
            if (c_syntax) {
                pr.body = c_block();
            } else {
                expect_id("begin");
                pr.body = stmts();
                expect_id("end");
                expect_id("procedure");
                if (is_sym(";")) i++;
            }
            for (const auto &q : m.procedures) if (q.name == pr.name) fail("procedure " + pr.name + " is declared twice");
            m.procedures.push_back(pr);
        }
    }
    void define_one(Module &m) {
        Definition d;
        d.in_define = true;
        d.line = cur().line;
        d.name = ident("a definition name");
        if (is_sym("(")) {
            i++;
            for (;;) { d.params.push_back(ident("a parameter name")); if (is_sym(",")) { i++; continue; } break; }
            expect_sym(")");
        }
        expect_sym("==");
        d.body = expr(0);
        m.defs.push_back(d);
    }
    void algorithm(Module &m) {
        // positioned after "--algorithm" / "--fair algorithm"
        m.algorithm = ident("the algorithm name");
        if (is_sym("{")) { c_algorithm(m); return; }
        if (is_id("variables") || is_id("variable")) { i++; m.globals = vardecls(); }
        if (is_id("define")) {
            i++;
            while (!is_id("end")) define_one(m);
            expect_id("end");
            expect_id("define");
            if (is_sym(";")) i++;
        }
        while (is_id("macro")) {
            i++;
            Macro mac;
            mac.name = ident("a macro name");
            expect_sym("(");
            if (!is_sym(")")) for (;;) { mac.params.push_back(ident("a parameter name")); if (is_sym(",")) { i++; continue; } break; }
            expect_sym(")");
            expect_id("begin");
            in_macro = true;
            mac.body = stmts();
            in_macro = false;
            expect_id("end");
            expect_id("macro");
            if (is_sym(";")) i++;
            m.macros.push_back(mac);
        }
        macros = &m.macros;
        procedures(m, false);
        if (is_id("begin")) {  // uniprocess
            i++;
            Proc p;
            p.body = stmts();
            m.procs.push_back(p);
        } else {
            while (is_id("process") || is_id("fair")) {
                if (is_id("fair")) { i++; if (is_sym("+")) i++; }
                expect_id("process");
                Proc p;
                p.name = ident("a process name");
                if (is_sym("=")) { i++; }
                else { expect_sym("\\in"); p.is_set = true; }
                p.id = expr(0);
                if (is_id("variables") || is_id("variable")) { i++; p.locals = vardecls(); }
                expect_id("begin");
                p.body = stmts();
                expect_id("end");
                expect_id("process");
                if (is_sym(";")) i++;
                m.procs.push_back(p);
            }
            if (m.procs.empty()) fail("expected `begin` or `process`");
        }
        expect_id("end");
        expect_id("algorithm");
    }
};

size_t find_line_start(const std::string &text, int line) {  // offset of 1-based line
    size_t p = 0;
    for (int l = 1; l < line && p != std::string::npos; l++) { p = text.find('\n', p); if (p != std::string::npos) p++; }
    return p == std::string::npos ? text.size() : p;
}
int line_of(const std::string &text, size_t off) { return 1 + (int)std::count(text.begin(), text.begin() + (long)off, '\n'); }

}  // namespace

// An algorithm written without any label gets the labels it needs, named Lbl_1, Lbl_2, ... like pcal2tla's (p-manual section 2.3
// p.9: "Because this is a uniprocess algorithm that contains no labels, the translator will automatically add the necessary
// labels"): on the first statement of a body, on every while, on a statement that assigns a variable already assigned in the
// step, and on the statement after an if / either that received a label inside.
namespace {
struct AutoLabel {
    int next = 1;
    static bool any_label(const std::vector<SP> &v) {
        for (const auto &s : v) {
            if (!s->label.empty()) return true;
            for (const auto &b : s->blocks) if (any_label(b)) return true;
        }
        return false;
    }
    static void assigned_in(const SP &s, std::set<std::string> &out) {
        if (s->k == Stmt::ASSIGN) {
            out.insert(s->var);
            for (const auto &o : s->more) out.insert(o->var);
        }
        for (const auto &b : s->blocks) for (const auto &x : b) assigned_in(x, out);
    }
    // returns true when a label was placed somewhere inside v (not counting v's own first statement)
    bool seq(std::vector<SP> &v, std::set<std::string> &assigned, bool first_needs_label) {
        bool need = first_needs_label, placed = false;
        for (size_t i = 0; i < v.size(); i++) {
            SP &s = v[i];
            std::set<std::string> mine;
            assigned_in(s, mine);
            bool clash = false;
            for (const auto &x : mine) clash |= assigned.count(x) != 0;
            if (need || clash || s->k == Stmt::WHILE) {
                s->label = "Lbl_" + std::to_string(next++);
                assigned.clear();
                if (i) placed = true;
            }
            need = false;
            if (s->k == Stmt::WHILE) {
                std::set<std::string> in_body;
                seq(s->blocks[0], in_body, false);
                assigned.clear();  // the exit path of the test assigns nothing
            } else if (s->k == Stmt::IF || s->k == Stmt::EITHER) {
                std::set<std::string> uni;
                bool inner = false;
                for (auto &b : s->blocks) {
                    std::set<std::string> a = assigned;
                    inner |= seq(b, a, false);
                    uni.insert(a.begin(), a.end());
                }
                if (inner) { need = true; placed = true; assigned.clear(); }
                else assigned = uni;
            } else {
                assigned.insert(mine.begin(), mine.end());  // ASSIGN, WITH (no label may go inside a with)
                if (s->k == Stmt::GOTO) need = true;
            }
        }
        return placed;
    }
};
void add_missing_labels(Module &m) {
    for (const auto &p : m.procs) if (AutoLabel::any_label(p.body)) return;
    AutoLabel al;
    for (auto &p : m.procs) {
        std::set<std::string> assigned;
        al.seq(p.body, assigned, true);
    }
}
}  // namespace

// ---- procedures: expanded into the processes that call them (pcal.h says what this preserves of pcal2tla's translation)
namespace {
struct ExpandError { std::string msg; };
// RECORD PARAMETERS (round 5, last part): a procedure parameter that some call passes a record (a record variable, an element of a record
// array, a constructor) becomes a record variable itself — its initial value the constructor [f |-> defaultInitValue, ...], which is what
// pcal2tla's `param = defaultInitValue` is, field by field — so that the `param := argument` of the expanded call and the reset of the
// `return` are record assignments the flattener knows.  Not for recursive procedures (their frames are plain cells).
void record_parameters(Module &m) {
    auto decl_shape = [&](const std::string &name) -> EP {
        auto look = [&](const std::vector<VarDecl> &v) -> EP {
            for (const auto &d : v)
                if (d.name == name && d.init && !d.in_set) {
                    if (d.init->k == Expr::RECORD) return d.init;
                    if (d.init->k == Expr::FUNCDEF && d.init->a[1]->k == Expr::RECORD) return d.init->a[1];
                }
            return nullptr;
        };
        if (EP r = look(m.globals)) return r;
        for (const auto &p : m.procs) if (EP r = look(p.locals)) return r;
        for (const auto &p : m.procedures) { if (EP r = look(p.params)) return r; if (EP r = look(p.locals)) return r; }
        return nullptr;
    };
    auto shape_of = [&](const EP &x) -> EP {
        if (!x) return nullptr;
        if (x->k == Expr::RECORD) return x;
        if (x->k == Expr::ID) return decl_shape(x->s);
        if (x->k == Expr::INDEX && x->a[0]->k == Expr::ID) return decl_shape(x->a[0]->s);
        return nullptr;
    };
    for (bool changed = true; changed;) {   // (a record handed on by a procedure to the next one: until nothing changes)
        changed = false;
        std::function<void(const std::vector<SP> &)> walk = [&](const std::vector<SP> &v) {
            for (const auto &s : v) {
                if (s->k == Stmt::CALL)
                    for (auto &pr : m.procedures)
                        if (pr.name == s->var)
                            for (size_t k = 0; k < pr.params.size() && k < s->args.size(); k++) {
                                VarDecl &pd = pr.params[k];
                                const EP sh = shape_of(s->args[k]);
                                if (!sh || (pd.init && pd.init->k == Expr::RECORD)) continue;
                                if (!pd.no_init) throw ExpandError{"line " + std::to_string(s->pos.line) + ", col " + std::to_string(s->pos.col) + ": parameter " + pd.name + " of " + pr.name + " has a default value and is passed a record: not supported"};
                                auto rc = std::make_shared<Expr>();
                                rc->k = Expr::RECORD;
                                rc->pos = pd.pos;
                                rc->names = sh->names;
                                for (size_t f = 0; f < sh->names.size(); f++) {
                                    if (sh->a[f]->k == Expr::RECORD || sh->a[f]->k == Expr::FUNCDEF) throw ExpandError{"line " + std::to_string(s->pos.line) + ", col " + std::to_string(s->pos.col) + ": a record with record / function fields as a procedure argument is not supported"};
                                    auto dv = std::make_shared<Expr>();
                                    dv->k = Expr::ID;
                                    dv->s = "defaultInitValue";
                                    dv->pos = pd.pos;
                                    rc->a.push_back(dv);
                                }
                                pd.init = rc;
                                pd.no_init = false;
                                changed = true;
                            }
                for (const auto &b : s->blocks) walk(b);
            }
        };
        for (const auto &p : m.procs) walk(p.body);
        for (const auto &pr : m.procedures) walk(pr.body);
    }
}

struct ProcExpander {
    Module &m;
    int copies = 0;
    [[noreturn]] static void fail(const Pos &at, const std::string &what) {
        throw ExpandError{"line " + std::to_string(at.line) + ", col " + std::to_string(at.col) + ": " + what};
    }
    const Procedure &find(const SP &call) const {
        for (const auto &p : m.procedures) if (p.name == call->var) return p;
        fail(call->pos, "`call " + call->var + "`: no such procedure");
    }
    static EP rename(const EP &e, const std::map<std::string, std::string> &r) {
        if (!e) return e;
        auto c = std::make_shared<Expr>(*e);
        if (c->k == Expr::ID) { auto it = r.find(c->s); if (it != r.end()) c->s = it->second; return c; }
        // (a bound variable of a quantifier / function constructor that shadows a procedure variable: PlusCal forbids the clash)
        for (auto &x : c->a) x = rename(x, r);
        return c;
    }
    static bool mentions(const EP &e, const std::set<std::string> &names) {
        if (!e) return false;
        if (e->k == Expr::ID && names.count(e->s)) return true;
        for (const auto &x : e->a) if (mentions(x, names)) return true;
        return false;
    }
    // one process being expanded
    struct Job {
        Proc *proc;
        std::string suffix;                         // appended to the procedure variables' names when several processes call procedures
        std::vector<std::string> active;            // procedures on the current call chain (recursion check)
        std::set<std::string> declared;             // procedure variables this process already has
        std::vector<SP> copies;                     // the expanded bodies, appended behind the process's own
        // RECURSIVE procedures (round 5): ONE copy of the body per process and a bounded call stack per procedure, kept as plain
        // variables — depth counter P_sp, and for every level K = 1 .. D a return-site code P_retK and one slot vK per variable v of
        // the procedure (what pcal2tla's frame [pc |-> ..., v |-> ...] holds: the value v had BEFORE the call).
        struct Rec {
            std::string entry, sp;                          // the copy's first label; the depth counter
            std::vector<std::string> ret;                   // P_ret1 .. P_retD
            std::vector<const VarDecl *> pv;                // parameters, then locals
            std::map<std::string, std::string> ren;         // v -> the process's variable
            std::map<std::string, std::vector<std::string>> slot;  // v -> its D slots
            std::vector<std::string> sites;                 // label a call returns to; its code is index + 1
            std::vector<std::pair<SP, Pos>> returns;        // the IF statements standing for `return`, filled in by finish_returns
        };
        std::map<std::string, Rec> rec;
    };
    std::set<std::string> recursive;   // procedures that can reach themselves through `call`
    static int stack_depth() {
        const char *e = getenv("TLAMC_PCAL_STACK");
        const int d = e ? atoi(e) : 4;
        return d < 1 ? 1 : d > 16 ? 16 : d;
    }
    static EP mk(Expr::K k, const Pos &at) { auto e = std::make_shared<Expr>(); e->k = k; e->pos = at; return e; }
    static EP num(long long v, const Pos &at) { auto e = mk(Expr::NUM, at); e->num = v; return e; }
    static EP ident(const std::string &n, const Pos &at) { auto e = mk(Expr::ID, at); e->s = n; return e; }
    static EP bin(const std::string &op, const EP &a, const EP &b, const Pos &at) { auto e = mk(Expr::BINOP, at); e->s = op; e->a = {a, b}; return e; }
    static EP ife(const EP &c, const EP &t, const EP &f, const Pos &at) { auto e = mk(Expr::IF, at); e->a = {c, t, f}; e->paren = true; return e; }
    // the slot the depth counter points at: IF sp = 1 THEN s1 ELSE IF sp = 2 THEN s2 ... ELSE sD
    static EP select(const std::vector<std::string> &slots, const std::string &sp, const Pos &at) {
        EP acc = ident(slots.back(), at);
        for (size_t k = slots.size() - 1; k-- > 0;) acc = ife(bin("=", ident(sp, at), num((long long)k + 1, at), at), ident(slots[k], at), acc, at);
        return acc;
    }
    static void collect_calls(const std::vector<SP> &v, std::set<std::string> &out) {
        for (const auto &s : v) {
            if (s->k == Stmt::CALL) out.insert(s->var);
            for (const auto &b : s->blocks) collect_calls(b, out);
        }
    }
    void find_recursive() {
        std::map<std::string, std::set<std::string>> g;
        for (const auto &pr : m.procedures) collect_calls(pr.body, g[pr.name]);
        for (const auto &pr : m.procedures) {
            std::set<std::string> seen;
            std::vector<std::string> todo(g[pr.name].begin(), g[pr.name].end());
            while (!todo.empty()) {
                const std::string x = todo.back();
                todo.pop_back();
                if (!seen.insert(x).second) continue;
                for (const auto &y : g[x]) todo.push_back(y);
            }
            if (seen.count(pr.name)) recursive.insert(pr.name);
        }
    }
    // `call P(args)` of a recursive procedure, in the caller's step: the stack must have room (an assertion: a run that needs more than
    // $TLAMC_PCAL_STACK levels — default 4 — fails HERE, at the call, instead of being cut short silently); push the procedure's
    // variables and the return site; parameters := arguments, locals := their initial values; goto the body's first label
    std::vector<SP> call_recursive(const SP &s, const Procedure &pr, const std::string &label, Job &job, const std::map<std::string, std::string> &outer,
                                   const std::string &after) {
        const int D = stack_depth();
        const bool first_call = !job.rec.count(pr.name);
        Job::Rec &rc = job.rec[pr.name];
        if (first_call) {
            for (const auto &d : pr.params) rc.pv.push_back(&d);
            for (const auto &d : pr.locals) rc.pv.push_back(&d);
            {
                std::set<std::string> own;
                for (const VarDecl *d : rc.pv) own.insert(d->name);
                for (const VarDecl *d : rc.pv)
                    if (d->init && mentions(d->init, own))
                        fail(d->pos, "the initial value of procedure variable " + d->name + " of " + pr.name +
                                     " mentions another variable of the procedure: not supported (assign it in the body's first step instead)");
            }
            auto declare = [&](const std::string &nm, const VarDecl *like, const EP &init) {
                if (!job.declared.insert(nm).second) return;
                VarDecl l;
                if (like) l = *like;
                l.name = nm;
                if (init) { l.init = init; l.no_init = false; l.in_set = false; }
                job.proc->locals.push_back(l);
            };
            rc.sp = pr.name + "_sp" + job.suffix;
            declare(rc.sp, nullptr, num(0, pr.pos));
            for (int k = 1; k <= D; k++) {
                rc.ret.push_back(pr.name + "_ret" + std::to_string(k) + job.suffix);
                declare(rc.ret.back(), nullptr, num(0, pr.pos));
            }
            for (const VarDecl *d : rc.pv) {
                if (d->in_set) fail(d->pos, "procedure variable " + d->name + ": `\\in` initial values are not supported in a recursive procedure");
                const std::string nm = d->name + job.suffix;
                rc.ren[d->name] = nm;
                declare(nm, d, nullptr);
                for (int k = 1; k <= D; k++) {
                    rc.slot[d->name].push_back(d->name + "_stk" + std::to_string(k) + job.suffix);
                    declare(rc.slot[d->name].back(), d, nullptr);
                }
            }
            rc.entry = pr.body[0]->label + "_p" + std::to_string(++copies);
        }
        const Pos at = s->pos;
        rc.sites.push_back(after);
        const long long code = (long long)rc.sites.size();
        std::vector<SP> out;
        auto chk = std::make_shared<Stmt>();
        chk->k = Stmt::ASSERT;
        chk->pos = at;
        chk->label = label;
        chk->e = bin("<", ident(rc.sp, at), num(D, at), at);   // call stack of procedure P deeper than D: raise TLAMC_PCAL_STACK
        // (NOT an assertion of the algorithm — a capacity limit of this translation, like a sequence that outgrows its
        //  cells: the compiled program reports MC_EOVERFLOW for it, and the text of the translation says what it is)
        chk->var = "$stack " + s->var + " " + std::to_string(D);
        out.push_back(chk);
        // One step, its statements in an order in which none reads what an earlier one wrote — so they need not be ONE multiple
        // assignment (whose right-hand sides the compiled program would all have to hold in temporaries): first the frame (reads the
        // depth counter and the variables' values before the call), then parameters := arguments || locals := initial values (an
        // argument may mention any of the procedure's own variables: `call down(n - 1)`), the depth counter last.
        for (int k = 1; k <= D; k++) {
            const EP here = bin("=", ident(rc.sp, at), num(k - 1, at), at);
            out.push_back(assign(rc.ret[k - 1], ife(here, num(code, at), ident(rc.ret[k - 1], at), at), at));
            for (const VarDecl *d : rc.pv) out.push_back(assign(rc.slot[d->name][k - 1], ife(here, ident(rc.ren[d->name], at), ident(rc.slot[d->name][k - 1], at), at), at));
        }
        SP first;
        auto add = [&](const std::string &var, const EP &e) {
            SP x = assign(var, e, at);
            if (!first) first = x; else first->more.push_back(x);
        };
        for (size_t a = 0; a < pr.params.size(); a++) add(rc.ren[pr.params[a].name], rename(s->args[a], outer));
        for (const auto &d : pr.locals) add(rc.ren[d.name], d.init);
        if (first) out.push_back(first);
        out.push_back(assign(rc.sp, bin("+", ident(rc.sp, at), num(1, at), at), at));
        auto g = std::make_shared<Stmt>();
        g->k = Stmt::GOTO;
        g->pos = at;
        g->var = rc.entry;
        out.push_back(g);
        if (first_call) {   // the ONE copy of the body in this process; its `return`s are IF statements filled in by finish_returns
            const std::string lsuf = rc.entry.substr(pr.body[0]->label.size());
            const std::map<std::string, std::string> inner = rc.ren;
            std::vector<SP> body = expand(pr.body, job, inner, lsuf, std::string(), "@" + pr.name, {});
            job.copies.insert(job.copies.end(), body.begin(), body.end());
        }
        return out;
    }
    // `return` of a recursive procedure: restore the variables from the top frame, clear the frame (a popped slot goes back to its
    // initial value: two states must not differ in what lies ABOVE the stack), pop, and go where the frame says
    void finish_returns(Job &job) {
        const int D = stack_depth();
        for (auto &kv : job.rec) {
            Job::Rec &rc = kv.second;
            for (auto &ph : rc.returns) {
                const Pos at = ph.second;
                // (again single assignments in an order in which none reads what an earlier one wrote: the variables from the top
                //  frame, the frame back to its initial values, the depth counter last)
                auto restore = [&]() {
                    std::vector<SP> v;
                    for (const VarDecl *d : rc.pv) v.push_back(assign(rc.ren[d->name], select(rc.slot[d->name], rc.sp, at), at));
                    for (int k = 1; k <= D; k++) {
                        const EP here = bin("=", ident(rc.sp, at), num(k, at), at);
                        v.push_back(assign(rc.ret[k - 1], ife(here, num(0, at), ident(rc.ret[k - 1], at), at), at));
                        for (const VarDecl *d : rc.pv) v.push_back(assign(rc.slot[d->name][k - 1], ife(here, d->init, ident(rc.slot[d->name][k - 1], at), at), at));
                    }
                    v.push_back(assign(rc.sp, bin("-", ident(rc.sp, at), num(1, at), at), at));
                    return v;
                };
                auto jump = [&](const std::string &to) { auto g = std::make_shared<Stmt>(); g->k = Stmt::GOTO; g->pos = at; g->var = to; return g; };
                // IF code = 1 THEN restore; goto site1 ELSE IF code = 2 ... ELSE (no such site: unreachable) assert FALSE; goto Done
                std::vector<SP> tail;
                {
                    auto bad = std::make_shared<Stmt>();
                    bad->k = Stmt::ASSERT;
                    bad->pos = at;
                    bad->e = mk(Expr::BOOL, at);   // FALSE
                    tail = {bad, jump("Done")};
                }
                for (size_t i = rc.sites.size(); i-- > 0;) {
                    const EP cond = bin("=", select(rc.ret, rc.sp, at), num((long long)i + 1, at), at);
                    std::vector<SP> then = restore();
                    then.push_back(jump(rc.sites[i]));
                    if (i == 0) {
                        ph.first->e = cond;
                        ph.first->blocks = {then, tail};
                    } else {
                        auto nested = std::make_shared<Stmt>();
                        nested->k = Stmt::IF;
                        nested->pos = at;
                        nested->e = cond;
                        nested->blocks = {then, tail};
                        tail = {nested};
                    }
                }
            }
        }
    }
    // statements of `v` with the procedures' variables renamed, labels given the copy's suffix, calls / returns expanded.
    // `cont` = the label control reaches after the last statement of v ("" = unknown: a call may not be last then);
    // `ret` = the label a `return` goes to ("" = not inside a procedure); `reset` = what a return assigns
    std::vector<SP> expand(const std::vector<SP> &v, Job &job, const std::map<std::string, std::string> &ren, const std::string &lsuf,
                           const std::string &cont, const std::string &ret, const std::vector<SP> &reset) {
        std::vector<SP> out;
        for (size_t i = 0; i < v.size(); i++) {
            const SP &s = v[i];
            const std::string next = i + 1 < v.size() ? (v[i + 1]->label.empty() ? std::string() : v[i + 1]->label + lsuf) : cont;
            auto c = std::make_shared<Stmt>(*s);
            if (!c->label.empty()) c->label += lsuf;
            c->e = rename(s->e, ren);
            c->idx = rename(s->idx, ren);
            if (s->k == Stmt::ASSIGN || s->k == Stmt::WITH) { auto it = ren.find(s->var); if (it != ren.end()) c->var = it->second; }
            if (s->k == Stmt::GOTO && s->var != "Done") c->var = s->var + lsuf;
            c->more.clear();
            for (const auto &o : s->more) {
                auto oc = std::make_shared<Stmt>(*o);
                oc->e = rename(o->e, ren);
                oc->idx = rename(o->idx, ren);
                auto it = ren.find(o->var);
                if (it != ren.end()) oc->var = it->second;
                c->more.push_back(oc);
            }
            if (s->k == Stmt::CALL) {
                if (i + 1 < v.size() && v[i + 1]->k == Stmt::RETURN && v[i + 1]->label.empty())
                    fail(s->pos, "`call` directly followed by `return` (pcal2tla's tail call: the frame is replaced, not pushed) is not supported: put a label on the return");
                if (next.empty()) fail(s->pos, "the statement after a `call` must have a label (p-manual section 3.5)");
                std::vector<SP> stmts = call(s, c->label, job, ren, next);
                out.insert(out.end(), stmts.begin(), stmts.end());
                continue;
            }
            if (s->k == Stmt::RETURN) {
                if (ret.empty()) fail(s->pos, "`return` outside a procedure");
                if (ret[0] == '@') {  // of a recursive procedure: where to is in the frame (finish_returns)
                    auto ph = std::make_shared<Stmt>();
                    ph->k = Stmt::IF;
                    ph->pos = s->pos;
                    ph->label = c->label;
                    out.push_back(ph);
                    job.rec[ret.substr(1)].returns.emplace_back(ph, s->pos);
                    continue;
                }
                bool first = true;
                for (const auto &r : reset) {
                    auto rc = std::make_shared<Stmt>(*r);
                    rc->pos = s->pos;
                    if (first) rc->label = c->label;
                    first = false;
                    out.push_back(rc);
                }
                auto g = std::make_shared<Stmt>();
                g->k = Stmt::GOTO;
                g->pos = s->pos;
                g->var = ret;
                if (first) g->label = c->label;
                out.push_back(g);
                continue;
            }
            c->blocks.clear();
            for (size_t b = 0; b < s->blocks.size(); b++) {
                // what follows a branch of an if / either is what follows the statement; the body of a while goes back to its test
                const std::string bcont = s->k == Stmt::WHILE ? c->label : next;
                if (s->k == Stmt::WITH) for (const auto &x : s->blocks[b]) if (x->k == Stmt::CALL || x->k == Stmt::RETURN) { /* allowed: no label inside a with, the goto is fine */ }
                c->blocks.push_back(expand(s->blocks[b], job, ren, lsuf, bcont, ret, reset));
            }
            out.push_back(c);
        }
        return out;
    }
    // every path through v ends in a return / goto (the copies of procedure bodies lie one behind the other: none may run into the next)
    static bool ends_in_jump(const std::vector<SP> &v) {
        if (v.empty()) return false;
        const SP &l = v.back();
        if (l->k == Stmt::RETURN || l->k == Stmt::GOTO) return true;
        if (l->k == Stmt::IF || l->k == Stmt::EITHER) {
            for (const auto &b : l->blocks) if (!ends_in_jump(b)) return false;
            return l->k == Stmt::EITHER || l->blocks.size() == 2;
        }
        if (l->k == Stmt::WITH) return ends_in_jump(l->blocks[0]);
        return false;
    }
    static SP assign(const std::string &var, const EP &e, const Pos &at) {
        auto a = std::make_shared<Stmt>();
        a->k = Stmt::ASSIGN;
        a->pos = at;
        a->var = var;
        a->e = e;
        return a;
    }
    // the statements that replace `call P(args)` in its step; P's body is copied behind the process's own
    std::vector<SP> call(const SP &s, const std::string &label, Job &job, const std::map<std::string, std::string> &outer, const std::string &after) {
        const Procedure &pr = find(s);
        if (s->args.size() != pr.params.size())
            fail(s->pos, "procedure " + pr.name + " takes " + std::to_string(pr.params.size()) + " arguments");
        if (pr.body.empty()) fail(pr.pos, "procedure " + pr.name + " has an empty body");
        if (pr.body[0]->label.empty()) fail(pr.body[0]->pos, "the first statement of procedure " + pr.name + " must have a label (p-manual section 3.5)");
        if (!ends_in_jump(pr.body)) fail(pr.body.back()->pos, "control can run off the end of procedure " + pr.name + ": it must end with `return` (or a `goto`) on every path");
        if (recursive.count(pr.name)) return call_recursive(s, pr, label, job, outer, after);
        for (const auto &a : job.active) if (a == pr.name) fail(s->pos, "recursive call of procedure " + pr.name + " (internal error: not recognised as recursive)");
        // the procedure's variables as locals of this process (once per process)
        std::map<std::string, std::string> ren;
        std::vector<const VarDecl *> pv;
        for (const auto &d : pr.params) pv.push_back(&d);
        for (const auto &d : pr.locals) pv.push_back(&d);
        // An initial value that mentions another variable of the procedure (`variables x = a + 1`) is refused: pcal2tla evaluates it
        // ONCE, when the frame is pushed, and a `return` pops the saved values — the expansion re-evaluates the expression at every
        // call and at every return, with whatever the mentioned variable holds then, and (with several calling processes) under
        // names the expression was never renamed to.  Constants, globals and `self` are fine.  
        {
            std::set<std::string> own;
            for (const VarDecl *d : pv) own.insert(d->name);
            for (const VarDecl *d : pv)
                if (d->init && mentions(d->init, own))
                    fail(d->pos, "the initial value of procedure variable " + d->name + " of " + pr.name +
                                 " mentions another variable of the procedure: not supported (assign it in the body's first step instead)");
        }
        for (const VarDecl *d : pv) {
            const std::string nm = d->name + job.suffix;
            ren[d->name] = nm;
            if (job.declared.insert(nm).second) {
                VarDecl l = *d;
                l.name = nm;
                job.proc->locals.push_back(l);
            }
        }
        const int k = ++copies;
        const std::string lsuf = "_p" + std::to_string(k);
        // in the caller's step: parameters := arguments (all at once: an argument may mention a parameter of an enclosing call),
        // locals := their initial values, goto the copy's first label
        std::vector<SP> out;
        SP first;
        for (size_t a = 0; a < pr.params.size(); a++) {
            SP x = assign(ren[pr.params[a].name], rename(s->args[a], outer), s->pos);
            if (!first) first = x; else first->more.push_back(x);
        }
        for (const auto &d : pr.locals) {
            SP x = assign(ren[d.name], d.init, s->pos);
            if (!first) first = x; else first->more.push_back(x);
        }
        if (first) { first->label = label; out.push_back(first); }
        auto g = std::make_shared<Stmt>();
        g->k = Stmt::GOTO;
        g->pos = s->pos;
        g->var = pr.body[0]->label + lsuf;
        if (!first) g->label = label;
        out.push_back(g);
        // what `return` assigns: every variable of the procedure back to its initial value (what pcal2tla's frame had saved)
        std::vector<SP> reset;
        {
            SP r0;
            for (const VarDecl *d : pv) {
                SP x = assign(ren[d->name], d->init, s->pos);
                if (!r0) r0 = x; else r0->more.push_back(x);
            }
            if (r0) reset.push_back(r0);
        }
        job.active.push_back(pr.name);
        std::map<std::string, std::string> inner = ren;  // (a procedure sees globals and its own variables; an enclosing procedure's are not in scope)
        std::vector<SP> body = expand(pr.body, job, inner, lsuf, std::string(), after, reset);
        job.active.pop_back();
        job.copies.insert(job.copies.end(), body.begin(), body.end());
        return out;
    }
    static bool uses_procedures(const std::vector<SP> &v) {
        for (const auto &s : v) {
            if (s->k == Stmt::CALL) return true;
            for (const auto &b : s->blocks) if (uses_procedures(b)) return true;
        }
        return false;
    }
    void run() {
        for (const auto &pr : m.procedures) {  // every declared procedure is checked, called or not
            if (pr.body.empty()) fail(pr.pos, "procedure " + pr.name + " has an empty body");
            if (pr.body[0]->label.empty()) fail(pr.body[0]->pos, "the first statement of procedure " + pr.name + " must have a label (p-manual section 3.5)");
            if (!ends_in_jump(pr.body)) fail(pr.body.back()->pos, "control can run off the end of procedure " + pr.name + ": it must end with `return` (or a `goto`) on every path");
        }
        find_recursive();
        for (const auto &pr : m.procedures)
            if (recursive.count(pr.name))
                for (const auto &d : pr.params)
                    if (d.init && d.init->k == Expr::RECORD) fail(d.pos, "parameter " + d.name + " of the RECURSIVE procedure " + pr.name + " is passed a record: not supported (its frames are plain cells)");
        int callers = 0;
        for (auto &p : m.procs) callers += uses_procedures(p.body) ? 1 : 0;
        for (auto &p : m.procs) {
            for (const auto &s : p.body) if (s->k == Stmt::RETURN) fail(s->pos, "`return` outside a procedure");
            if (!uses_procedures(p.body)) continue;
            Job job;
            job.proc = &p;
            job.suffix = callers > 1 ? "_" + (p.name.empty() ? std::string("main") : p.name) : std::string();
            std::vector<SP> body = expand(p.body, job, {}, "", "Done", "", {});
            finish_returns(job);
            if (!job.copies.empty()) {  // the process's own body must not run into the copies
                if (body.empty() || body.back()->k != Stmt::GOTO) {
                    auto g = std::make_shared<Stmt>();
                    g->k = Stmt::GOTO;
                    g->var = "Done";
                    g->pos = body.empty() ? Pos{} : body.back()->pos;
                    body.push_back(g);
                }
                body.insert(body.end(), job.copies.begin(), job.copies.end());
            }
            p.body = body;
        }
        m.had_procedures = !m.procedures.empty();
    }
};
}  // namespace

// ---- records: kept field by field (pcal.h, RECORDS)
namespace {
struct FlattenError { std::string msg; };
// NESTED records (round 5): one LEVEL per pass.  Pass k replaces every variable whose value is a record constructor by one variable per
// field; a field whose own value is a constructor becomes a variable with a record value (`pending`), which pass k + 1 replaces in turn:
// r = [a |-> 0, f |-> [g |-> 1, h |-> 2]]  ->  r_a, r_f = [g |-> 1, h |-> 2]  ->  r_a, r_f_g, r_f_h.  Expressions follow: `r.f.g` is
// `r_f.g` after pass k and `r_f_g` after pass k + 1; a record value that has to survive a pass (the right-hand side of `r.f := ...`,
// an operand of `=`) is handed on as a constructor over the fields' variables (as_value).
struct RecordFlattener {
    Module &m;
    std::map<std::string, RecordVar> recs;   // the record variables THIS pass replaces, by name
    std::map<std::string, EP> shapes;        // their constructors, and those of the variables this pass declares with a record value
    std::set<std::string> pending;           // ... the latter: the next pass's record variables
    std::map<std::string, EP> rsets;         // SETS of records (RecordVar::set): they stay variables; name -> the elements' constructor
    mutable std::vector<std::pair<std::string, EP>> bound;   // `with m \in msgs` / `\E m \in msgs`: m is a record value with msgs' fields
    mutable std::vector<std::pair<std::string, EP>> split;   // `with v = r`: v is a record value bound FIELD BY FIELD (v.f is the bound name v_f)
    int depth = 0;
    [[noreturn]] static void fail(const Pos &at, const std::string &msg) { throw FlattenError{"line " + std::to_string(at.line) + ", column " + std::to_string(at.col) + ": " + msg}; }
    static EP node(Expr::K k, const Pos &at) { auto e = std::make_shared<Expr>(); e->k = k; e->pos = at; return e; }
    static EP id(const std::string &name, const Pos &at) { auto e = node(Expr::ID, at); e->s = name; return e; }
    static const std::string &base_name(const EP &e) {  // of r or r[i]
        static const std::string none;
        const EP &b = e->k == Expr::INDEX ? e->a[0] : e;
        return b->k == Expr::ID ? b->s : none;
    }
    const RecordVar *rec_of(const EP &e) const {  // r or r[i] of a record variable r of this pass
        auto it = recs.find(base_name(e));
        if (it == recs.end() || it->second.seq) return nullptr;
        return &it->second;
    }
    // ---- sequences of records (pcal.h, RecordVar::seq): q / box[i] is kept as one sequence per field
    const RecordVar *recseq(const EP &e) const {  // e = q, or box[i] of an array of record sequences
        if (e->k == Expr::ID) {
            auto it = recs.find(e->s);
            return it != recs.end() && it->second.seq && !it->second.array ? &it->second : nullptr;
        }
        if (e->k == Expr::INDEX && e->a[0]->k == Expr::ID) {
            auto it = recs.find(e->a[0]->s);
            return it != recs.end() && it->second.seq && it->second.array ? &it->second : nullptr;
        }
        return nullptr;
    }
    const RecordVar *recseq_elem(const EP &e) const {  // e = Head(Q) or Q[k]: an ELEMENT of a record sequence Q
        if (e->k == Expr::CALL && e->s == "Head" && e->a.size() == 1) return recseq(e->a[0]);
        if (e->k == Expr::INDEX) return recseq(e->a[0]);
        return nullptr;
    }
    EP proj_seq(const EP &q, const std::string &f) const {  // the sequence of the f fields of Q
        const RecordVar *r = recseq(q);
        if (!r) fail(q->pos, "expected a sequence of records (a variable, or an element of an array of them)");
        check_field(*r, f, q->pos);
        if (q->k == Expr::ID) return id(r->name + "_" + f, q->pos);
        auto x = node(Expr::INDEX, q->pos);
        x->a = {id(r->name + "_" + f, q->a[0]->pos), rw(q->a[1])};
        return x;
    }
    // field f of a record-SEQUENCE-valued expression: <<R, ..>>, Q, Append(Q, R), Tail(Q), Q \o <<R, ..>>
    EP proj_seq_expr(const EP &e, const std::string &f, const RecordVar &dst) const {
        auto elem = [&](const EP &x) {
            if (!record_valued(x)) fail(x->pos, "an element of the sequence of records " + dst.name + " must be a record constructor, a record variable or an element of a sequence / array of records");
            if (!same_fields(dst.fields, fields_of(x))) fail(x->pos, "the record put into " + dst.name + " does not have its fields");
            return field_of(x, f);
        };
        auto c = std::make_shared<Expr>(*e);
        if (e->k == Expr::TUPLE) { for (auto &x : c->a) x = elem(x); return c; }
        if (recseq(e)) return proj_seq(e, f);
        if (e->k == Expr::CALL && e->s == "Append" && e->a.size() == 2) { c->a = {proj_seq(e->a[0], f), elem(e->a[1])}; return c; }
        if (e->k == Expr::CALL && e->s == "Tail" && e->a.size() == 1) { c->a = {proj_seq(e->a[0], f)}; return c; }
        if (e->k == Expr::BINOP && (e->s == "\\o" || e->s == "\\circ") && e->a[1]->k == Expr::TUPLE) {
            auto t = std::make_shared<Expr>(*e->a[1]);
            for (auto &x : t->a) x = elem(x);
            c->a = {proj_seq(e->a[0], f), t};
            return c;
        }
        fail(e->pos, "a sequence of records can be assigned <<...>>, itself, Append(q, r), Tail(q) or q \\o <<...>>");
    }
    bool pending_ref(const EP &e) const { return (e->k == Expr::ID || e->k == Expr::INDEX) && pending.count(base_name(e)) != 0; }
    static int field_index(const EP &rec, const std::string &f) {
        for (size_t i = 0; i < rec->names.size(); i++) if (rec->names[i] == f) return (int)i;
        return -1;
    }
    // the constructor that says which fields a record-valued expression has (null: not a record value)
    EP shape(const EP &e) const {
        if (!e) return nullptr;
        if (e->k == Expr::RECORD) return e;
        if (const EP bs = bound_shape(e)) return bs;
        if (const RecordVar *r = recseq_elem(e)) return shapes.at(r->name);
        if (recseq(e)) return nullptr;
        if (e->k == Expr::ID || e->k == Expr::INDEX) {
            const std::string &n = base_name(e);
            if (!n.empty() && recs.count(n) && recs.at(n).seq) return nullptr;
            if (!n.empty()) {
                auto it = shapes.find(n);
                return it == shapes.end() ? nullptr : it->second;
            }
            if (e->k == Expr::INDEX && e->a[0]->k == Expr::DOT) {  // r.f[i]: an element of a field that is a function to records
                const EP sb = shape(e->a[0]->a[0]);
                const int k = sb ? field_index(sb, e->a[0]->s) : -1;
                if (k >= 0 && sb->a[(size_t)k]->k == Expr::FUNCDEF && sb->a[(size_t)k]->a[1]->k == Expr::RECORD) return sb->a[(size_t)k]->a[1];
            }
            return nullptr;
        }
        if (e->k == Expr::DOT) {
            const EP sb = shape(e->a[0]);
            const int k = sb ? field_index(sb, e->s) : -1;
            return k >= 0 && sb->a[(size_t)k]->k == Expr::RECORD ? sb->a[(size_t)k] : nullptr;
        }
        return nullptr;
    }
    bool record_valued(const EP &e) const { return shape(e) != nullptr; }
    std::vector<std::string> fields_of(const EP &e) const { return shape(e)->names; }
    static void check_field(const RecordVar &r, const std::string &f, const Pos &at) {
        for (const auto &x : r.fields) if (x == f) return;
        fail(at, "record " + r.name + " has no field " + f);
    }
    EP with_index(const EP &e) const {  // r or r[i] of a variable that stays for the next pass: the index is rewritten now
        if (e->k != Expr::INDEX) return e;
        auto c = std::make_shared<Expr>(*e);
        c->a[1] = rw(e->a[1]);
        return c;
    }
    // a record value as the NEXT pass reads it: a constructor over the fields' variables, or a variable the next pass replaces
    EP as_value(const EP &e) const {
        if (e->k == Expr::RECORD) {
            auto c = std::make_shared<Expr>(*e);
            for (auto &x : c->a) x = record_valued(x) ? as_value(x) : rw(x);
            return c;
        }
        if (const RecordVar *r = rec_of(e) ? rec_of(e) : recseq_elem(e)) {
            auto c = node(Expr::RECORD, e->pos);
            c->names = r->fields;
            for (const auto &f : r->fields) c->a.push_back(field_of(e, f));
            return c;
        }
        if (pending_ref(e)) return with_index(e);
        if (const EP bs = bound_shape(e)) {   // an element of a set of records, named by `with` / a quantifier: [f |-> m.f, ...]
            auto c = node(Expr::RECORD, e->pos);
            c->names = bs->names;
            for (const auto &f : bs->names) c->a.push_back(field_of(e, f));
            return c;
        }
        return rw(e);  // a path `x.f` to a record-valued field
    }
    EP bound_shape(const EP &e) const {
        if (e->k == Expr::ID) for (size_t i = split.size(); i-- > 0;) if (split[i].first == e->s) return split[i].second;
        if (e->k == Expr::ID) for (size_t i = bound.size(); i-- > 0;) if (bound[i].first == e->s) return bound[i].second;
        return nullptr;
    }
    bool is_split(const EP &e) const {
        if (e->k == Expr::ID) for (const auto &x : split) if (x.first == e->s) return true;
        return false;
    }
    // field f of a record-valued expression
    EP field_of(const EP &e, const std::string &f) const {
        if (e->k == Expr::RECORD) {
            const int k = field_index(e, f);
            if (k < 0) fail(e->pos, "the record has no field " + f);
            const EP &v = e->a[(size_t)k];
            return record_valued(v) ? as_value(v) : rw(v);
        }
        if (const EP bs = bound_shape(e)) {  // m.f of an element of a set of records: the compiled program reads the field's cell
            if (field_index(bs, f) < 0) fail(e->pos, "the record has no field " + f);
            if (is_split(e)) return id(e->s + "_" + f, e->pos);   // with v = r: v.f is the bound name v_f
            auto d = node(Expr::DOT, e->pos);
            d->s = f;
            d->a = {e};
            return d;
        }
        if (const RecordVar *r = recseq_elem(e)) {  // Head(Q).f = Head(Q_f), Q[k].f = Q_f[k]
            check_field(*r, f, e->pos);
            auto c = std::make_shared<Expr>(*e);
            c->a[0] = proj_seq(e->a[0], f);
            if (e->k == Expr::INDEX) c->a[1] = rw(e->a[1]);
            return c;
        }
        if (const RecordVar *r = rec_of(e)) {
            check_field(*r, f, e->pos);
            if (e->k == Expr::ID) return id(r->name + "_" + f, e->pos);
            auto x = node(Expr::INDEX, e->pos);
            x->a = {id(r->name + "_" + f, e->a[0]->pos), rw(e->a[1])};
            return x;
        }
        // a deeper level: the path is resolved as far as this pass can, the rest is the next pass's
        const EP sb = shape(e);
        if (!sb || field_index(sb, f) < 0) fail(e->pos, "the record has no field " + f);
        auto d = node(Expr::DOT, e->pos);
        d->s = f;
        d->a = {as_value(e)};
        return d;
    }
    static bool same_fields(std::vector<std::string> a, std::vector<std::string> b) {
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        return a == b;
    }
    // ---- sets of records (pcal.h, RecordVar::set): the variable stays; record values around it become constructors over plain values
    // which operand of `S \cup {..}`, `S \ {..}`, `r \in S`, `S = {}` is a set-of-records variable (-1: none)
    int rset_side(const EP &e) const {
        static const char *ops[] = {"\\cup", "\\union", "\\", "\\in", "\\notin", "=", "#", "\\subseteq"};
        bool is = false;
        for (const char *o : ops) is |= e->s == o;
        if (!is || e->a.size() != 2) return -1;
        for (int side = 1; side >= 0; side--) if (e->a[(size_t)side]->k == Expr::ID && rsets.count(e->a[(size_t)side]->s) && !bound_shape(e->a[(size_t)side])) return side;
        return -1;
    }
    EP rset_elem(const EP &x, const EP &want, const std::string &name) const {  // a record value as an element of the set `name`
        if (!record_valued(x)) fail(x->pos, "an element of the set of records " + name + " must be a record constructor, a record variable or an element of a set / sequence / array of records");
        if (!same_fields(want->names, fields_of(x))) fail(x->pos, "the record does not have the fields of the elements of " + name);
        auto c = node(Expr::RECORD, x->pos);   // in the set's own field order
        c->names = want->names;
        for (const auto &f : want->names) c->a.push_back(field_of(x, f));
        return c;
    }
    // S | chain \cup {r, ...} | chain \ {r, ...}: what a set of records can be assigned
    bool rset_chain(const EP &e, const std::string &name) const {
        if (e->k == Expr::ID) return e->s == name && !bound_shape(e);
        return e->k == Expr::BINOP && (e->s == "\\cup" || e->s == "\\union" || e->s == "\\") && e->a[1]->k == Expr::SETENUM && rset_chain(e->a[0], name);
    }
    EP rw_rset_chain(const EP &e, const std::string &name) const {
        if (e->k == Expr::ID) return e;
        auto c = std::make_shared<Expr>(*e);
        c->a[0] = rw_rset_chain(e->a[0], name);
        auto t = std::make_shared<Expr>(*e->a[1]);
        for (auto &x : t->a) x = rset_elem(x, rsets.at(name), name);
        c->a[1] = t;
        return c;
    }
    EP rw_rset(const EP &e) const {
        const int side = rset_side(e);
        const std::string &name = e->a[(size_t)side]->s;
        const EP &want = rsets.at(name);
        const EP &other = e->a[(size_t)(1 - side)];
        auto c = std::make_shared<Expr>(*e);
        if (e->s == "\\in" || e->s == "\\notin") {
            if (side != 1) fail(e->pos, "a set of records cannot be a member of something");
            c->a[0] = rset_elem(other, want, name);
            return c;
        }
        if (other->k == Expr::ID && rsets.count(other->s)) {
            if (e->s == "=" || e->s == "#" || e->s == "\\subseteq") fail(e->pos, "comparing two sets of records is not supported (supported: = {}, # {}, \\in, Cardinality)");
            fail(e->pos, "a set of records can be united with / reduced by {r, ...} only");
        }
        if (other->k != Expr::SETENUM) fail(other->pos, "a set of records works with {r, ...} ({} included) only: " + name + " \\cup {r}, " + name + " \\ {r}, r \\in " + name + ", " + name + " = {}");
        if ((e->s == "=" || e->s == "#") && !other->a.empty()) fail(e->pos, "a set of records can only be compared with {}");
        if (e->s == "\\subseteq") fail(e->pos, "\\subseteq on a set of records is not supported");
        if (e->s == "\\" && side != 0) fail(e->pos, "only " + name + " \\ {r, ...} is supported");
        auto t = std::make_shared<Expr>(*other);
        for (auto &x : t->a) x = rset_elem(x, want, name);
        c->a[(size_t)(1 - side)] = t;
        return c;
    }
    EP rw(const EP &e) const {
        if (!e) return e;
        switch (e->k) {
        case Expr::DOT: {
            EP b = e->a[0];
            if (b->k == Expr::DOT) b = rw(b);  // the inner path first: r.f.g
            else if (b->k == Expr::INDEX && b->a[0]->k == Expr::DOT) {  // r.f[i].g
                auto c = std::make_shared<Expr>(*b);
                c->a[0] = rw(b->a[0]);
                c->a[1] = rw(b->a[1]);
                b = c;
            }
            if (b->k == Expr::RECORD || rec_of(b) || recseq_elem(b) || bound_shape(b)) return field_of(b, e->s);
            if (pending_ref(b) || b->k == Expr::DOT) {  // a field of a record this pass has only just made a variable of (or a path into it): the next pass's
                const EP sb = shape(b);
                if (!sb) fail(e->pos, "`." + e->s + "`: not a field of a record");
                if (field_index(sb, e->s) < 0) fail(e->pos, "the record has no field " + e->s);
                auto c = std::make_shared<Expr>(*e);
                c->a = {b->k == Expr::DOT ? b : with_index(b)};
                return c;
            }
            fail(e->pos, "`." + e->s + "`: field access is supported on record variables (r." + e->s + ", r[i]." + e->s + ") only");
        }
        case Expr::CALL:
            if (e->s == "Len" && e->a.size() == 1) if (const RecordVar *r = recseq(e->a[0])) {
                auto c = std::make_shared<Expr>(*e);
                c->a = {proj_seq(e->a[0], r->fields[0])};
                return c;
            }
            if (recseq_elem(e)) fail(e->pos, "an element of a sequence of records is used as a whole value here (supported: Head(q).f, r := Head(q), Head(q) = ...)");
            break;
        case Expr::INDEX:
            if (recseq_elem(e)) fail(e->pos, "an element of a sequence of records is used as a whole value here (supported: q[k].f, r := q[k], q[k] = ...)");
            if (recseq(e)) fail(e->pos, "a sequence of records is used as a whole value here (supported: Len, Head, [k], = / # <<>>, :=)");
            break;
        case Expr::QUANT:
            if (e->a[0]->k == Expr::ID && rsets.count(e->a[0]->s) && e->s != "CHOOSE") {   // \E m \in msgs : P(m.f)
                auto c = std::make_shared<Expr>(*e);
                bound.push_back({e->bound, rsets.at(e->a[0]->s)});
                try { c->a[1] = rw(e->a[1]); } catch (...) { bound.pop_back(); throw; }
                bound.pop_back();
                return c;
            }
            break;
        case Expr::BINOP:
            if (rset_side(e) >= 0) return rw_rset(e);
            if ((e->s == "=" || e->s == "#") && (recseq(e->a[0]) || recseq(e->a[1]))) {
                const int side = recseq(e->a[0]) ? 0 : 1;
                const EP &other = e->a[(size_t)(1 - side)];
                if (other->k != Expr::TUPLE || !other->a.empty()) fail(e->pos, "a sequence of records can only be compared with <<>>");
                auto c = std::make_shared<Expr>(*e);
                c->a[(size_t)side] = proj_seq(e->a[(size_t)side], recseq(e->a[(size_t)side])->fields[0]);
                return c;
            }
            if ((e->s == "=" || e->s == "#") && (record_valued(e->a[0]) || record_valued(e->a[1]))) {
                if (!record_valued(e->a[0]) || !record_valued(e->a[1])) fail(e->pos, "a record can only be compared with a record variable, an element of a record array or a record constructor");
                const auto fs = fields_of(e->a[0]);
                if (!same_fields(fs, fields_of(e->a[1]))) fail(e->pos, "comparison of records with different fields");
                EP acc;
                for (const auto &f : fs) {
                    auto eq = node(Expr::BINOP, e->pos);
                    eq->s = "=";
                    eq->a = {field_of(e->a[0], f), field_of(e->a[1], f)};   // (record-valued fields: a comparison of records for the next pass)
                    if (!acc) { acc = eq; continue; }
                    auto both = node(Expr::BINOP, e->pos);
                    both->s = "/\\";
                    both->a = {acc, eq};
                    acc = both;
                }
                acc->paren = true;
                if (e->s == "=") return acc;
                auto no = node(Expr::UNOP, e->pos);
                no->s = "~";
                no->a = {acc};
                no->paren = true;
                return no;
            }
            break;
        case Expr::ID:
            if (bound_shape(e)) fail(e->pos, "the record " + e->s + " (an element of a set of records) is used as a whole value here (supported: " + e->s + ".f, r := " + e->s + ", " + e->s + " = ..., {" + e->s + "})");
            if (recs.count(e->s) && recs.at(e->s).seq) fail(e->pos, "the sequence of records " + e->s + " is used as a whole value here (supported: Len, Head, [k], = / # <<>>, :=)");
            if (recs.count(e->s)) fail(e->pos, "record variable " + e->s + " is used as a whole value here (supported: " + e->s + ".f, " + e->s + " := ..., " + e->s + " = ...)");
            return e;
        case Expr::RECORD:
            fail(e->pos, "a record constructor is supported as the initial value of a variable, on the right of `:=` to a record variable and in `=` / `#` only");
        default: break;
        }
        auto c = std::make_shared<Expr>(*e);
        for (auto &x : c->a) x = rw(x);
        return c;
    }
    // one assignment `var[idx].path := e` -> the assignments to the fields' variables
    void assignment(const Stmt &a, std::vector<SP> &out) const {
        auto mk = [&](const std::string &var, const EP &idx, const EP &e, const std::string &whole, const std::string &path) {
            auto o = std::make_shared<Stmt>();
            o->k = Stmt::ASSIGN;
            o->pos = a.pos;
            o->var = var;
            o->idx = idx;
            o->e = e;
            o->whole = whole;
            o->field = path;
            for (const auto &x : out) if (x->var == var && paths_overlap(x->field, path)) fail(a.pos, "two assignments to " + var + " in one statement");
            out.push_back(o);
        };
        if (rsets.count(a.var)) {   // msgs := msgs \cup {r} | msgs \ {r} | {r, ...} | {}
            if (!a.field.empty() || a.idx) fail(a.pos, a.var + " is a set of records: assign the set (" + a.var + " := " + a.var + " \\cup {r}, ...)");
            const EP &want = rsets.at(a.var);
            EP v;
            if (a.e->k == Expr::SETENUM) {
                auto t = std::make_shared<Expr>(*a.e);
                for (auto &x : t->a) x = rset_elem(x, want, a.var);
                v = t;
            } else if (rset_chain(a.e, a.var)) {   // msgs, msgs \cup {r}, (msgs \ {m}) \cup {r}, ...
                v = rw_rset_chain(a.e, a.var);
            } else {
                fail(a.e->pos, "a set of records can be assigned {r, ...}, " + a.var + " \\cup {r, ...} or " + a.var + " \\ {r, ...}");
            }
            mk(a.var, nullptr, v, a.whole, "");
            return;
        }
        auto it = recs.find(a.var);
        if (it == recs.end()) {
            if (!a.field.empty()) fail(a.pos, a.var + " is not a record variable (its initial value is not a record constructor)");
            if (record_valued(a.e)) fail(a.e->pos, "a record is assigned to " + a.var + ", which is not a record variable (its initial value is not a record constructor)");
            mk(a.var, rw(a.idx), rw(a.e), a.whole, "");
            return;
        }
        const RecordVar &r = it->second;
        const std::string root = a.whole.empty() ? r.name : a.whole;   // the one variable pcal2tla sees (one assignment per step)
        if (r.seq) {
            if (!a.field.empty()) fail(a.pos, r.name + " is a sequence of records: assign an element (" + r.name + "[k] := r) or the sequence (Append, Tail, <<...>>)");
            if (r.array && !a.idx) fail(a.pos, "assignment to the whole array of sequences " + r.name + " (supported: " + r.name + "[i] := Append(" + r.name + "[i], r), ...)");
            if (!r.array && a.idx) {   // q[k] := R: one element
                if (!record_valued(a.e) || !same_fields(r.fields, fields_of(a.e))) fail(a.e->pos, "the value assigned to an element of " + r.name + " must be a record with its fields");
                for (const auto &f : r.fields) mk(r.name + "_" + f, rw(a.idx), field_of(a.e, f), root, "");
                return;
            }
            for (const auto &f : r.fields) mk(r.name + "_" + f, rw(a.idx), proj_seq_expr(a.e, f, r), root, "");
            return;
        }
        if (r.array != (a.idx != nullptr)) fail(a.pos, r.array ? "assignment to the whole record array " + r.name + " (supported: " + r.name + "[i] := ..., " + r.name + "[i].f := ...)" : r.name + " is a record, not an array of records");
        if (!a.field.empty()) {
            const size_t dot = a.field.find('.');
            const std::string first = a.field.substr(0, dot), rest = dot == std::string::npos ? std::string() : a.field.substr(dot + 1);
            check_field(r, first, a.pos);
            const bool sub = std::find(r.sub.begin(), r.sub.end(), first) != r.sub.end();
            if (!rest.empty()) {
                if (!sub) fail(a.pos, r.name + "." + first + " is not a record");
                mk(r.name + "_" + first, rw(a.idx), record_valued(a.e) ? as_value(a.e) : rw(a.e), root, rest);
                return;
            }
            if (sub) {
                const EP want = shapes.at(r.name)->a[(size_t)field_index(shapes.at(r.name), first)];
                if (!record_valued(a.e)) fail(a.e->pos, "the value assigned to " + r.name + "." + first + " must be a record constructor, a record variable or an element of a record array");
                if (!same_fields(want->names, fields_of(a.e))) fail(a.e->pos, "the value assigned to " + r.name + "." + first + " does not have its fields");
                mk(r.name + "_" + first, rw(a.idx), as_value(a.e), root, "");
                return;
            }
            if (record_valued(a.e)) fail(a.e->pos, "a record is assigned to " + r.name + "." + first + ", which is not a record (its initial value is not a record constructor)");
            mk(r.name + "_" + first, rw(a.idx), rw(a.e), root, "");
            return;
        }
        if (!record_valued(a.e)) fail(a.e->pos, "the value assigned to record variable " + r.name + " must be a record constructor, a record variable or an element of a record array");
        if (!same_fields(r.fields, fields_of(a.e))) fail(a.e->pos, "the value assigned to " + r.name + " does not have its fields");
        for (const auto &f : r.fields) mk(r.name + "_" + f, rw(a.idx), field_of(a.e, f), root, "");
    }
    void stmts(std::vector<SP> &v) const {
        for (auto &s : v) {
            if (s->k == Stmt::ASSIGN) {
                std::vector<SP> flat;
                assignment(*s, flat);
                for (const auto &o : s->more) assignment(*o, flat);
                auto c = std::make_shared<Stmt>(*flat[0]);
                c->label = s->label;
                c->more.assign(flat.begin() + 1, flat.end());
                s = c;
                continue;
            }
            if (s->k == Stmt::WITH && s->e && s->with_eq && record_valued(s->e)) {
                // with v = r do ... v.f ... : one `with v_f = r.f` per field, the body inside the innermost
                const EP sh = shape(s->e);
                for (const auto &x : sh->a) if (x->k == Expr::RECORD) fail(s->e->pos, "`with " + s->var + " = ...` over a record with record fields is not supported");
                std::vector<EP> vals;
                for (const auto &f : sh->names) vals.push_back(field_of(s->e, f));
                auto body = s->blocks;
                split.push_back({s->var, sh});
                try { for (auto &b : body) stmts(b); } catch (...) { split.pop_back(); throw; }
                split.pop_back();
                SP inner;
                for (size_t k = sh->names.size(); k-- > 0;) {
                    auto w = std::make_shared<Stmt>(*s);
                    w->var = s->var + "_" + sh->names[k];
                    w->e = vals[k];
                    w->label = k == 0 ? s->label : "";
                    w->blocks = inner ? std::vector<std::vector<SP>>{{inner}} : body;
                    inner = w;
                }
                s = inner;
                continue;
            }
            if (s->k == Stmt::WITH && s->e && record_valued(s->e)) fail(s->e->pos, "`with` over a record value is supported as `with v = r` only");
            auto c = std::make_shared<Stmt>(*s);
            if (s->k == Stmt::WITH && !s->with_eq && s->e && s->e->k == Expr::ID && rsets.count(s->e->s)) {   // with m \in msgs do ... m.f ... end with
                bound.push_back({s->var, rsets.at(s->e->s)});
                try { for (auto &b : c->blocks) stmts(b); } catch (...) { bound.pop_back(); throw; }
                bound.pop_back();
                s = c;
                continue;
            }
            c->e = rw(s->e);
            c->idx = rw(s->idx);
            for (auto &b : c->blocks) stmts(b);
            s = c;
        }
    }
    // the constructor that says which fields the ELEMENTS of the sequence variable `name` have: the first record constructor (or record
    // variable, by its declaration) among its initial elements and among what the algorithm puts into it — null: not a sequence of records
    EP decl_shape(const std::string &name) const {
        auto look = [&](const std::vector<VarDecl> &v) -> EP {
            for (const auto &d : v)
                if (d.name == name && d.init && !d.in_set) {
                    if (d.init->k == Expr::RECORD) return d.init;
                    if (d.init->k == Expr::FUNCDEF && d.init->a[1]->k == Expr::RECORD) return d.init->a[1];
                }
            return nullptr;
        };
        if (EP r = look(m.globals)) return r;
        for (const auto &p : m.procs) if (EP r = look(p.locals)) return r;
        return nullptr;
    }
    EP seq_elem_shape(const std::string &name, const EP &init_tuple) const {
        auto of_value = [&](const EP &x) -> EP {
            if (!x) return nullptr;
            if (x->k == Expr::RECORD) return x;
            if (x->k == Expr::ID) return decl_shape(x->s);
            if (x->k == Expr::INDEX && x->a[0]->k == Expr::ID) return decl_shape(x->a[0]->s);
            return nullptr;
        };
        for (const auto &x : init_tuple->a) if (EP r = of_value(x)) return r;
        EP found;
        std::function<void(const std::vector<SP> &)> walk = [&](const std::vector<SP> &v) {
            for (const auto &st : v) {
                if (found) return;
                if (st->k == Stmt::ASSIGN) {
                    std::vector<const Stmt *> all{st.get()};
                    for (const auto &o : st->more) all.push_back(o.get());
                    for (const Stmt *a : all) {
                        if (a->var != name || !a->e || !a->field.empty() || found) continue;
                        const EP &e = a->e;
                        if (e->k == Expr::TUPLE) { for (const auto &x : e->a) if (!found) found = of_value(x); }
                        else if (e->k == Expr::CALL && e->s == "Append" && e->a.size() == 2) found = of_value(e->a[1]);
                        else if (e->k == Expr::BINOP && (e->s == "\\o" || e->s == "\\circ") && e->a[1]->k == Expr::TUPLE) { for (const auto &x : e->a[1]->a) if (!found) found = of_value(x); }
                    }
                }
                for (const auto &b : st->blocks) walk(b);
            }
        };
        for (const auto &p : m.procs) walk(p.body);
        return found;
    }
    // the elements' constructor of the SET variable `name` (initial elements, `name \cup {r}`, `{r}` somewhere in the algorithm) — null: not a set of records
    EP set_elem_shape(const std::string &name, const EP &init_set) const {
        auto of_value = [&](const EP &x) -> EP {
            if (!x) return nullptr;
            if (x->k == Expr::RECORD) return x;
            if (x->k == Expr::ID) return decl_shape(x->s);
            if (x->k == Expr::INDEX && x->a[0]->k == Expr::ID) return decl_shape(x->a[0]->s);
            return nullptr;
        };
        for (const auto &x : init_set->a) if (EP r = of_value(x)) return r;
        EP found;
        std::function<void(const std::vector<SP> &)> walk = [&](const std::vector<SP> &v) {
            for (const auto &st : v) {
                if (found) return;
                if (st->k == Stmt::ASSIGN) {
                    std::vector<const Stmt *> all{st.get()};
                    for (const auto &o : st->more) all.push_back(o.get());
                    for (const Stmt *a : all) {
                        if (a->var != name || !a->e || !a->field.empty() || a->idx || found) continue;
                        const EP &e = a->e;
                        const EP lit = e->k == Expr::SETENUM ? e : e->k == Expr::BINOP && (e->s == "\\cup" || e->s == "\\union") && e->a[1]->k == Expr::SETENUM ? e->a[1] : nullptr;
                        if (lit) for (const auto &x : lit->a) if (!found) found = of_value(x);
                    }
                }
                for (const auto &b : st->blocks) walk(b);
            }
        };
        for (const auto &p : m.procs) walk(p.body);
        return found;
    }
    void find_rsets() {
        auto scan = [&](std::vector<VarDecl> &v, int proc) {
            for (auto &d : v) {
                if (!d.init || d.in_set || d.init->k != Expr::SETENUM) continue;
                const EP es = set_elem_shape(d.name, d.init);
                if (!es) continue;
                for (size_t i = 0; i < es->names.size(); i++)
                    if (es->a[i]->k == Expr::RECORD || es->a[i]->k == Expr::FUNCDEF || es->a[i]->k == Expr::TUPLE || es->a[i]->k == Expr::SETENUM)
                        fail(es->a[i]->pos, "an element of the set of records " + d.name + " can only have plain fields (field " + es->names[i] + " is a record, a function, a sequence or a set)");
                if (proc >= 0 && m.procs[(size_t)proc].is_set) fail(d.pos, "a set of records local to a process SET is not supported (`" + d.name + "`)");
                rsets[d.name] = es;
                RecordVar r;
                r.name = d.name;
                r.fields = es->names;
                r.set = true;
                r.shape = es;
                r.proc = proc;
                m.records.push_back(r);
            }
        };
        scan(m.globals, -1);
        for (size_t k = 0; k < m.procs.size(); k++) scan(m.procs[k].locals, (int)k);
    }
    // declarations: a record variable becomes one variable per field
    void decls(std::vector<VarDecl> &v, int proc, const std::set<std::string> &taken) {
        std::vector<VarDecl> out;
        for (const auto &d : v) {
            const EP &e = d.init;
            const bool scalar = e && !d.in_set && e->k == Expr::RECORD;
            const bool array = e && !d.in_set && e->k == Expr::FUNCDEF && e->a[1]->k == Expr::RECORD;
            if (!scalar && !array) {
                // a SEQUENCE of records (or an array of them): `q = <<>>` / `box = [p \in S |-> <<>>]` into which records are put
                const bool sq = e && !d.in_set && e->k == Expr::TUPLE, sqa = e && !d.in_set && e->k == Expr::FUNCDEF && e->a[1]->k == Expr::TUPLE;
                const EP es = sq || sqa ? seq_elem_shape(d.name, sq ? e : e->a[1]) : nullptr;
                if (!es) { out.push_back(d); continue; }
                const EP &tup = sq ? e : e->a[1];
                RecordVar r;
                r.name = d.name;
                r.fields = es->names;
                r.array = sqa;
                r.seq = true;
                r.shape = es;
                r.proc = proc;
                r.depth = depth;
                if (sqa) { r.bound = e->bound; r.domain = e->a[0]; }
                for (size_t i = 0; i < es->names.size(); i++) {
                    if (es->a[i]->k == Expr::RECORD || es->a[i]->k == Expr::FUNCDEF || es->a[i]->k == Expr::TUPLE)
                        fail(es->a[i]->pos, "an element of the sequence of records " + d.name + " can only have plain fields (field " + es->names[i] + " is a record, a function or a sequence)");
                    VarDecl f = d;
                    f.name = d.name + "_" + es->names[i];
                    if (taken.count(f.name)) fail(d.pos, "field " + es->names[i] + " of the sequence of records " + d.name + " is kept as a variable " + f.name + ", and that name is taken");
                    auto ft = std::make_shared<Expr>(*tup);   // the initial elements, field by field
                    for (auto &x : ft->a) {
                        if (x->k != Expr::RECORD || !same_fields(x->names, es->names)) fail(x->pos, "the initial elements of " + d.name + " must be record constructors with the same fields");
                        x = x->a[(size_t)field_index(x, es->names[i])];
                    }
                    if (sq) f.init = ft;
                    else { auto fn = std::make_shared<Expr>(*e); fn->a[1] = ft; f.init = fn; }
                    out.push_back(f);
                }
                shapes[r.name] = es;
                m.records.push_back(r);
                recs[r.name] = r;
                continue;
            }
            const EP &rc = scalar ? e : e->a[1];
            RecordVar r;
            r.name = d.name;
            r.fields = rc->names;
            r.array = array;
            r.proc = proc;
            r.depth = depth;
            if (array) { r.bound = e->bound; r.domain = e->a[0]; }
            for (size_t i = 0; i < rc->names.size(); i++) {
                VarDecl f = d;
                f.name = d.name + "_" + rc->names[i];
                if (taken.count(f.name)) fail(d.pos, "field " + rc->names[i] + " of record variable " + d.name + " is kept as a variable " + f.name + ", and that name is taken");
                if (scalar) {
                    f.init = rc->a[i];
                    f.no_init = f.init->k == Expr::ID && f.init->s == "defaultInitValue";   // (a record PARAMETER: pcal2tla's `param = defaultInitValue`, field by field)
                } else { auto fn = std::make_shared<Expr>(*e); fn->a[1] = rc->a[i]; f.init = fn; }
                if (rc->a[i]->k == Expr::RECORD) {  // a nested record: the next pass's record variable
                    if (depth >= 6) fail(rc->a[i]->pos, "records nest too deeply");
                    r.sub.push_back(rc->names[i]);
                    pending.insert(f.name);
                    shapes[f.name] = rc->a[i];
                } else if (scalar && rc->a[i]->k == Expr::FUNCDEF && rc->a[i]->a[1]->k == Expr::RECORD) {  // a function to records: the next pass's record ARRAY
                    if (depth >= 6) fail(rc->a[i]->pos, "records nest too deeply");
                    pending.insert(f.name);
                    shapes[f.name] = rc->a[i]->a[1];
                } else if (array && rc->a[i]->k == Expr::FUNCDEF) {
                    fail(rc->a[i]->pos, "a function as a field of an ELEMENT of a record array is not supported (" + d.name + "[i]." + rc->names[i] + "[j])");
                }
                out.push_back(f);
            }
            shapes[r.name] = rc;
            m.records.push_back(r);
            recs[r.name] = r;
        }
        v = out;
    }
    void run() {
        find_rsets();
        for (depth = 0;; depth++) {
            std::set<std::string> taken(m.constants.begin(), m.constants.end());
            for (const auto &g : m.globals) taken.insert(g.name);
            for (const auto &p : m.procs) for (const auto &l : p.locals) taken.insert(l.name);
            for (const auto &d : m.defs) taken.insert(d.name);
            for (const auto &r : m.records) taken.insert(r.name);   // (a replaced record's name stays a DEFINITION of the translation)
            recs.clear();
            shapes.clear();
            pending.clear();
            const size_t before = m.records.size();
            decls(m.globals, -1, taken);
            for (size_t k = 0; k < m.procs.size(); k++) decls(m.procs[k].locals, (int)k, taken);
            if (depth > 0 && m.records.size() == before) break;  // nothing left to replace
            // (pass 0 without a record variable too: a `.f` or a record constructor anywhere is refused by rw(), with its position)
            auto init_of = [&](const VarDecl &d) -> EP {
                if (pending.count(d.name)) return d.init;
                if (rsets.count(d.name)) {   // the initial elements of a set of records, in the set's field order
                    auto t = std::make_shared<Expr>(*d.init);
                    for (auto &x : t->a) x = rset_elem(x, rsets.at(d.name), d.name);
                    return t;
                }
                return rw(d.init);
            };
            for (auto &g : m.globals) g.init = init_of(g);
            for (auto &p : m.procs) {
                for (auto &l : p.locals) l.init = init_of(l);
                stmts(p.body);
            }
            for (auto &d : m.defs) {
                try {
                    d.body = rw(d.body);
                } catch (const FlattenError &) {
                    if (d.in_define) throw;
                    d.body = nullptr;  // a definition of the module text beyond the subset: unusable as an invariant, like an unparsed one
                }
            }
            std::vector<Definition> kept;
            for (auto &d : m.defs) if (d.body) kept.push_back(d);
            m.defs = kept;
            if (pending.empty()) break;
        }
        // the translation DEFINES every replaced record from its fields' variables (`r == [f |-> r_f, ...]`): an inner record before
        // the record that holds it
        std::stable_sort(m.records.begin(), m.records.end(), [](const RecordVar &a, const RecordVar &b) { return a.depth > b.depth; });
    }
};
}  // namespace

std::string parse_module(const std::string &text, Module &m) {
    try {
        size_t a = text.find("--algorithm");
        const size_t fa = text.find("--fair algorithm");
        size_t kw_len = strlen("--algorithm");
        if (fa != std::string::npos && (a == std::string::npos || fa < a)) { a = fa; kw_len = strlen("--fair algorithm"); }
        if (a == std::string::npos) return "no PlusCal algorithm (`--algorithm`) in the module";
        const size_t cbeg = text.rfind("(*", a);
        if (cbeg == std::string::npos) return "the PlusCal algorithm must be inside a (* ... *) comment";
        // end of the enclosing comment (comments nest); the p-syntax ends with `end algorithm`, the c-syntax with `}`
        size_t cend = std::string::npos;
        {
            int depth = 1;
            for (size_t k = cbeg + 2; k + 1 < text.size(); k++) {
                if (text[k] == '(' && text[k + 1] == '*') { depth++; k++; }
                else if (text[k] == '*' && text[k + 1] == ')') { if (--depth == 0) { cend = k + 2; break; } k++; }
            }
        }
        if (cend == std::string::npos) return "the algorithm comment is not closed";
        const size_t aend = cend - 2;
        m.alg_first_line = line_of(text, cbeg);
        m.alg_last_line = line_of(text, cend - 1);
        // header: module name, constants
        {
            Parser h(lex(text, 0, cbeg));
            for (; h.cur().t != Tok::END; h.i++) {
                if (h.cur().t == Tok::IDENT && h.cur().s == "MODULE" && h.peek().t == Tok::IDENT) m.name = h.peek().s;
                if (h.cur().t == Tok::IDENT && (h.cur().s == "CONSTANT" || h.cur().s == "CONSTANTS")) {
                    h.i++;
                    while (h.cur().t == Tok::IDENT) {
                        m.constants.push_back(h.cur().s);
                        h.i++;
                        if (h.is_sym(",")) h.i++; else break;
                    }
                    h.i--;
                }
            }
        }
        {
            Parser p(lex(text, a + kw_len, aend));
            p.algorithm(m);
        }
        {   // (always: a `call` / `return` in an algorithm without procedures is refused there)
            try {
                record_parameters(m);
                ProcExpander{m}.run();
            } catch (const ExpandError &e) {
                return e.msg;
            }
        }
        add_missing_labels(m);
        {   // one name, one variable (pcal2tla renames a second `t` to `t_`; here the author does: both back-ends refuse alike)
            std::set<std::string> names{"pc"};
            auto declare = [&](const VarDecl &d) { return names.insert(d.name).second; };
            for (const auto &g : m.globals) if (!declare(g)) return "variable `" + g.name + "` is declared twice";
            for (const auto &p : m.procs) for (const auto &l : p.locals) if (!declare(l)) return "variable `" + l.name + "` is declared twice (a variable of one process may not have the name of a global variable or of another process's variable)";
        }
        {   // a label becomes a DEFINITION of the translation (one action per label): it may not take the name of one the translation
            // writes itself (pcal2tla renames such a label; here the author does) — `Next: ...` would make `Next` call itself
            std::set<std::string> own{"Init", "Next", "Spec", "Termination", "vars", "ProcSet"};
            for (const auto &p : m.procs) if (!p.name.empty()) own.insert(p.name);
            std::function<std::string(const std::vector<SP> &)> clash = [&](const std::vector<SP> &v) -> std::string {
                for (const auto &st : v) {
                    if (!st->label.empty() && own.count(st->label))
                        return "line " + std::to_string(st->pos.line) + ": the label `" + st->label + "` has the name of a definition of the translation (Init, Next, Spec, Termination, vars, ProcSet, a process): rename it";
                    for (const auto &b : st->blocks) { const std::string r = clash(b); if (!r.empty()) return r; }
                }
                return "";
            };
            for (const auto &p : m.procs) { const std::string r = clash(p.body); if (!r.empty()) return r; }
        }
        // the rest: an existing translation (skipped) and the definitions
        size_t rest = cend;
        const size_t tb = text.find("\\* BEGIN TRANSLATION", cend);
        size_t te = std::string::npos;
        if (tb != std::string::npos) {
            te = text.find("\\* END TRANSLATION", tb);
            if (te == std::string::npos) return "`\\* BEGIN TRANSLATION` without `\\* END TRANSLATION`";
            m.has_translation = true;
            m.tr_first_line = line_of(text, tb);
            m.tr_last_line = line_of(text, te);
        }
        std::vector<Tok> toks;
        if (m.has_translation) {
            toks = lex(text, rest, tb);
            toks.pop_back();
            const size_t after = text.find('\n', te);
            auto more = lex(text, after == std::string::npos ? text.size() : after, text.size());
            toks.insert(toks.end(), more.begin(), more.end());
        } else {
            toks = lex(text, rest, text.size());
        }
        Parser d(toks);
        while (d.cur().t != Tok::END) {
            size_t hdr = 0;  // tokens of "Name ==" or "Name(p, q) =="
            std::vector<std::string> params;
            if (d.cur().t == Tok::IDENT && d.cur().col == 1) {
                if (d.peek().t == Tok::SYM && d.peek().s == "==") hdr = 2;
                else if (d.peek().t == Tok::SYM && d.peek().s == "(") {
                    size_t j = d.i + 2;
                    bool ok = true;
                    while (ok) {
                        if (d.t[j].t != Tok::IDENT) { ok = false; break; }
                        params.push_back(d.t[j].s);
                        j++;
                        if (d.t[j].t == Tok::SYM && d.t[j].s == ",") { j++; continue; }
                        break;
                    }
                    if (ok && d.t[j].t == Tok::SYM && d.t[j].s == ")" && d.t[j + 1].t == Tok::SYM && d.t[j + 1].s == "==") hdr = j + 2 - d.i;
                }
            }
            if (hdr) {
                Definition def;
                def.name = d.cur().s;
                def.line = d.cur().line;
                def.params = params;
                d.i += hdr;
                // the body ends at the next token in column 1 (next definition / separator / keyword)
                size_t j = d.i;
                while (d.t[j].t != Tok::END && d.t[j].col != 1) j++;
                std::vector<Tok> body(d.t.begin() + (long)d.i, d.t.begin() + (long)j);
                Tok end;
                end.line = d.t[j].line;
                end.col = d.t[j].col;
                body.push_back(end);
                Parser bp(body);
                try {
                    def.body = bp.expr(0);
                    if (bp.cur().t == Tok::END) m.defs.push_back(def);  // otherwise: beyond the expression subset, ignored unless used
                } catch (const ParseError &) {
                }
                d.i = j;
            } else {
                d.i++;
            }
        }
        try {
            RecordFlattener{m}.run();
        } catch (const FlattenError &e) {
            return e.msg;
        }
        return "";
    } catch (const LexError &e) {
        return e.msg;
    } catch (const ParseError &e) {
        return e.msg;
    }
}

// ============================================================================================ translator
namespace {

struct Ctx {
    const Module *m = nullptr;
    const Proc *proc = nullptr;
    std::set<std::string> locals;   // names of this process's local variables
    std::string self;               // "self" for a process set, the id text for a single process, "" for a uniprocess
    bool multi = false;             // multiprocess algorithm: pc is a function
};

std::string pe(const EP &e, const Ctx &c, const std::set<std::string> &primed, const std::set<std::string> &shadow);

std::string pe_inner(const EP &e, const Ctx &c, const std::set<std::string> &primed, const std::set<std::string> &shadow) {
    switch (e->k) {
    case Expr::NUM: return std::to_string(e->num);
    case Expr::STR: return "\"" + e->s + "\"";
    case Expr::BOOL: return e->num ? "TRUE" : "FALSE";
    case Expr::ID: {
        if (shadow.count(e->s)) return e->s;
        if (e->s == "self" && c.proc && !c.proc->is_set && !c.self.empty()) return c.self;
        std::string s = e->s;
        if (primed.count(s)) s += "'";
        if (c.locals.count(e->s) && c.proc && c.proc->is_set) s += "[self]";
        return s;
    }
    case Expr::UNOP: return e->s + (e->s == "DOMAIN" ? " " : "") + pe(e->a[0], c, primed, shadow);
    case Expr::BINOP: {
        const std::string l = pe(e->a[0], c, primed, shadow), r = pe(e->a[1], c, primed, shadow);
        if (e->s == "..") return l + ".." + r;
        return l + " " + e->s + " " + r;
    }
    case Expr::INDEX: return pe(e->a[0], c, primed, shadow) + "[" + pe(e->a[1], c, primed, shadow) + "]";
    case Expr::PRIME: return pe(e->a[0], c, primed, shadow) + "'";
    case Expr::CALL: {
        std::string r = e->s + "(";
        for (size_t i = 0; i < e->a.size(); i++) r += (i ? ", " : "") + pe(e->a[i], c, primed, shadow);
        return r + ")";
    }
    case Expr::IF:
        return "IF " + pe(e->a[0], c, primed, shadow) + " THEN " + pe(e->a[1], c, primed, shadow) + " ELSE " + pe(e->a[2], c, primed, shadow);
    case Expr::QUANT: {
        std::set<std::string> sh = shadow;
        const std::string dom = pe(e->a[0], c, primed, shadow);
        sh.insert(e->bound);
        return e->s + " " + e->bound + " \\in " + dom + " : " + pe(e->a[1], c, primed, sh);
    }
    case Expr::SETOF: {
        std::set<std::string> sh = shadow;
        const std::string dom = pe(e->a[0], c, primed, shadow);
        sh.insert(e->bound);
        if (e->s == "filter") return "{" + e->bound + " \\in " + dom + " : " + pe(e->a[1], c, primed, sh) + "}";
        return "{" + pe(e->a[1], c, primed, sh) + " : " + e->bound + " \\in " + dom + "}";
    }
    case Expr::FUNCDEF: {
        std::set<std::string> sh = shadow;
        const std::string dom = pe(e->a[0], c, primed, shadow);
        sh.insert(e->bound);
        return "[" + e->bound + " \\in " + dom + " |-> " + pe(e->a[1], c, primed, sh) + "]";
    }
    case Expr::DOT: return pe(e->a[0], c, primed, shadow) + "." + e->s;
    case Expr::RECORD: {
        std::string s = "[";
        for (size_t i = 0; i < e->a.size(); i++) s += (i ? ", " : "") + e->names[i] + " |-> " + pe(e->a[i], c, primed, shadow);
        return s + "]";
    }
    case Expr::SETENUM:
    case Expr::TUPLE: {
        std::string s = e->k == Expr::SETENUM ? "{" : "<<";
        for (size_t i = 0; i < e->a.size(); i++) s += (i ? ", " : "") + pe(e->a[i], c, primed, shadow);
        return s + (e->k == Expr::SETENUM ? "}" : ">>");
    }
    }
    return "?";
}
std::string pe(const EP &e, const Ctx &c, const std::set<std::string> &primed, const std::set<std::string> &shadow) {
    const std::string s = pe_inner(e, c, primed, shadow);
    return e->paren ? "(" + s + ")" : s;
}

// the right-hand side of `v' = e` (or of an EXCEPT `= e`): an operator that binds no tighter than `=` needs parentheses,
// or `f' = a \\/ b` would read as `(f' = a) \\/ b`
std::string pe_rhs(const EP &e, const Ctx &c, const std::set<std::string> &primed, const std::set<std::string> &shadow) {
    const std::string s = pe(e, c, primed, shadow);
    if (e->paren) return s;
    bool loose = e->k == Expr::QUANT;
    if (e->k == Expr::BINOP) {
        static const char *ops[] = {"=>", "<=>", "\\/", "/\\", "=", "#", "<", ">", "<=", ">=", "\\in", "\\notin", "\\subseteq"};
        for (const char *o : ops) loose |= e->s == o;
    }
    return loose ? "(" + s + ")" : s;
}

// formula tree of one action, rendered with the translator's column conventions
struct Node;
using NP = std::shared_ptr<Node>;
struct Node {
    enum K { LINE, CONJ, DISJ, IF, EXISTS, ASSERT } k = LINE;
    std::string text, text2;   // LINE; IF condition; EXISTS "x \in S"; ASSERT expression / message
    std::vector<NP> kids;      // CONJ / DISJ items; IF: [then, else]; EXISTS: [body]
};
NP line(const std::string &s) { auto n = std::make_shared<Node>(); n->text = s; return n; }
NP conj() { auto n = std::make_shared<Node>(); n->k = Node::CONJ; return n; }

// renders `n` whose first character lands at column `col` (1-based) of the current line
void render(const NP &n, int col, std::string &out) {
    auto nl = [&](int c) { out += "\n"; out.append((size_t)(c - 1), ' '); };
    switch (n->k) {
    case Node::LINE: out += n->text; break;
    case Node::ASSERT:
        out += "Assert(" + n->text + ", ";
        nl(col + 7);
        out += "\"" + n->text2 + "\")";
        break;
    case Node::CONJ:
    case Node::DISJ:
        for (size_t i = 0; i < n->kids.size(); i++) {
            if (i) nl(col);
            out += n->k == Node::CONJ ? "/\\ " : "\\/ ";
            render(n->kids[i], col + 3, out);
        }
        break;
    case Node::IF:
        out += "IF " + n->text;
        nl(col + 3);
        out += "THEN ";
        render(n->kids[0], col + 8, out);
        nl(col + 3);
        out += "ELSE ";
        render(n->kids[1], col + 8, out);
        break;
    case Node::EXISTS:
        out += "\\E " + n->text + ":";
        nl(col + 2);
        render(n->kids[0], col + 2, out);
        break;
    }
}

bool contains_label(const std::vector<SP> &v);
bool contains_label(const SP &s) {
    for (const auto &b : s->blocks) if (contains_label(b)) return true;
    return false;
}
bool contains_label(const std::vector<SP> &v) {  // a label or a goto: either one makes the enclosing statement set pc itself
    for (const auto &s : v) if (!s->label.empty() || s->k == Stmt::GOTO || contains_label(s)) return true;
    return false;
}

struct TranslateError { std::string msg; };

struct ActionGen {
    const Module &m;
    Ctx c;
    std::vector<std::string> var_order;  // globals then all locals (VARIABLES order without pc)
    ActionGen(const Module &mod, const Ctx &ctx) : m(mod), c(ctx) {
        for (const auto &g : m.globals) var_order.push_back(g.name);
        for (const auto &p : m.procs) for (const auto &l : p.locals) var_order.push_back(l.name);
    }
    std::string pc_ref() const { return c.multi ? "pc[" + c.self + "]" : "pc"; }
    std::string pc_set(const std::string &label) const {
        return c.multi ? "pc' = [pc EXCEPT ![" + c.self + "] = \"" + label + "\"]" : "pc' = \"" + label + "\"";
    }
    std::string unchanged(const std::set<std::string> &vs, int col) const {
        std::vector<std::string> names;
        for (const auto &v : var_order) if (vs.count(v)) names.push_back(v);
        if (names.size() == 1) return "UNCHANGED " + names[0];
        // wrapped like the translator: a line is broken (after ", ") before it would pass column 78
        std::string s = "UNCHANGED << ";
        int cur = col + (int)s.size();
        const int item_col = cur;
        for (size_t i = 0; i < names.size(); i++) {
            const bool last = i + 1 == names.size();
            const int need = (int)names[i].size() + (last ? 3 : 1);
            if (i && cur + need > 78) { s += "\n" + std::string((size_t)(item_col - 1), ' '); cur = item_col; }
            s += names[i];
            cur += (int)names[i].size();
            if (!last) { s += ", "; cur += 2; }
        }
        return s + " >>";
    }

    struct Out {
        std::vector<NP> items;
        std::set<std::string> assigned;
    };
    // one statement without labels inside; `primed` = variables assigned so far on this path
    void simple(const SP &s, Out &o, std::set<std::string> &primed, const std::set<std::string> &shadow, int col) {
        switch (s->k) {
        case Stmt::ASSIGN: {
            if (s->var == "pc") throw TranslateError{"assignment to pc at line " + std::to_string(s->pos.line)};
            if (shadow.count(s->var)) throw TranslateError{"assignment to the `with` variable " + s->var + " at line " + std::to_string(s->pos.line)};
            bool known = false;
            for (const auto &v : var_order) known |= v == s->var;
            if (!known) throw TranslateError{"assignment to undeclared variable " + s->var + " at line " + std::to_string(s->pos.line)};
            if (primed.count(s->var))
                throw TranslateError{"second assignment to " + s->var + " in one step (line " + std::to_string(s->pos.line) + "): a label is needed between them"};
            // a field's variable: pcal2tla sees ONE variable, the record (only `||` may assign two of its fields in a step)
            const std::string rec_mark = s->whole.empty() ? std::string() : s->whole + ".";
            if (!rec_mark.empty() && primed.count(rec_mark))
                throw TranslateError{"second assignment to " + s->whole + " in one step (line " + std::to_string(s->pos.line) + "): a label is needed between them"};
            const std::string rhs = pe_rhs(s->e, c, primed, shadow);
            const bool local_fn = c.locals.count(s->var) && c.proc && c.proc->is_set;
            std::string t;
            if (!s->idx && !local_fn) t = s->var + "' = " + rhs;
            else {
                std::string path;
                if (local_fn) path += "[self]";
                if (s->idx) path += "[" + pe(s->idx, c, primed, shadow) + "]";
                t = s->var + "' = [" + s->var + " EXCEPT !" + path + " = " + rhs + "]";
            }
            o.items.push_back(line(t));
            o.assigned.insert(s->var);
            if (!s->more.empty()) {  // simultaneous: every right-hand side reads the values before the statement
                std::set<std::string> before = primed;
                std::set<std::string> after = primed;
                after.insert(s->var);
                for (const auto &x : s->more) {
                    Out tmp;
                    std::set<std::string> pr = before;
                    if (after.count(x->var)) throw TranslateError{"second assignment to " + x->var + " in one step (line " + std::to_string(x->pos.line) + ")"};
                    simple(x, tmp, pr, shadow, col);
                    for (auto &it : tmp.items) o.items.push_back(it);
                    o.assigned.insert(x->var);
                    after.insert(x->var);
                    if (!x->whole.empty()) after.insert(x->whole + ".");
                }
                if (!rec_mark.empty()) after.insert(rec_mark);
                primed = after;
                break;
            }
            primed.insert(s->var);
            if (!rec_mark.empty()) primed.insert(rec_mark);
            break;
        }
        case Stmt::AWAIT: o.items.push_back(line(pe(s->e, c, primed, shadow))); break;
        case Stmt::SKIP: o.items.push_back(line("TRUE")); break;
        case Stmt::PRINT: o.items.push_back(line("PrintT(" + pe(s->e, c, primed, shadow) + ")")); break;
        case Stmt::ASSERT: {
            auto n = std::make_shared<Node>();
            n->k = Node::ASSERT;
            n->text = pe(s->e, c, primed, shadow);
            n->text2 = "Failure of assertion at line " + std::to_string(s->pos.line) + ", column " + std::to_string(s->pos.col) + ".";
            if (s->var.rfind("$stack ", 0) == 0) {   // the bounded call stack of a recursive procedure (expand_procedures)
                const std::string rest = s->var.substr(7);
                const size_t sp = rest.rfind(' ');
                n->text2 = "The call at line " + std::to_string(s->pos.line) + ", column " + std::to_string(s->pos.col) + " needs more than the " + rest.substr(sp + 1) +
                           " stack frames this translation reserves for procedure " + rest.substr(0, sp) + ": raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm).";
            }
            o.items.push_back(n);
            break;
        }
        default: throw TranslateError{"internal: not a simple statement"};
        }
        (void)col;
    }

    // statements [i, n) of `v` inside the current action; `cont` = label control reaches when the sequence ends.
    // `top` = the first statement is the action's own labeled statement.
    void seq(const std::vector<SP> &v, size_t i, const std::string &cont, bool top, Out &o, std::set<std::string> primed,
             const std::set<std::string> &shadow, int col) {
        for (size_t j = i; j < v.size(); j++) {
            const SP &s = v[j];
            if (!s->label.empty() && !(top && j == i)) {  // control passes to another action
                o.items.push_back(line(pc_set(s->label)));
                return;
            }
            const std::string after = j + 1 < v.size() ? (v[j + 1]->label.empty() ? std::string() : v[j + 1]->label) : cont;
            auto need_after = [&]() {
                if (after.empty())
                    throw TranslateError{"the statement after the " + std::string(s->k == Stmt::IF ? "if" : s->k == Stmt::EITHER ? "either" : "with") +
                                         " at line " + std::to_string(s->pos.line) + " needs a label (a label occurs inside it)"};
            };
            switch (s->k) {
            case Stmt::GOTO: o.items.push_back(line(pc_set(s->var))); return;
            case Stmt::WHILE: {
                if (!(top && j == i)) throw TranslateError{"the while at line " + std::to_string(s->pos.line) + " needs a label"};
                if (s->e->k == Expr::BOOL && s->e->num == 1) {
                    // p-manual App. B p.61 (action ncs of FastMutex): "The evaluation of the while test does not appear
                    // explicitly in the action because it equals TRUE" — the body is the action, control returns to the label
                    seq(s->blocks[0], 0, s->label, false, o, primed, shadow, col);
                    return;
                }
                auto n = std::make_shared<Node>();
                n->k = Node::IF;
                n->text = pe(s->e, c, primed, shadow);
                Out a, b;
                seq(s->blocks[0], 0, s->label, false, a, primed, shadow, col + 11);
                seq(v, j + 1, cont, false, b, primed, shadow, col + 11);
                branch_unchanged({&a, &b}, col + 11);
                n->kids = {wrap(a), wrap(b)};
                o.items.push_back(n);
                o.assigned.insert(a.assigned.begin(), a.assigned.end());
                o.assigned.insert(b.assigned.begin(), b.assigned.end());
                return;
            }
            case Stmt::IF:
            case Stmt::EITHER: {
                const bool lab = contains_label(s);
                if (lab) need_after();
                std::vector<Out> outs(s->blocks.size());
                std::vector<Out *> ptrs;
                const int inner = s->k == Stmt::IF ? col + 11 : col + 6;
                for (size_t b = 0; b < s->blocks.size(); b++) {
                    if (lab) seq(s->blocks[b], 0, after, false, outs[b], primed, shadow, inner);
                    else {
                        std::set<std::string> pr = primed;
                        for (const auto &st : s->blocks[b]) nested(st, outs[b], pr, shadow, inner);
                        if (outs[b].items.empty()) outs[b].items.push_back(line("TRUE"));
                    }
                    ptrs.push_back(&outs[b]);
                }
                branch_unchanged(ptrs, inner);
                auto n = std::make_shared<Node>();
                if (s->k == Stmt::IF) {
                    n->k = Node::IF;
                    n->text = pe(s->e, c, primed, shadow);
                } else {
                    n->k = Node::DISJ;
                }
                for (auto &ob : outs) { n->kids.push_back(wrap(ob)); o.assigned.insert(ob.assigned.begin(), ob.assigned.end()); }
                o.items.push_back(n);
                if (lab) return;  // every branch set pc'
                for (const auto &ob : outs) primed.insert(ob.assigned.begin(), ob.assigned.end());
                break;
            }
            case Stmt::WITH: {
                const bool lab = contains_label(s);
                if (lab) throw TranslateError{"labels inside `with` (line " + std::to_string(s->pos.line) + ") are not allowed"};
                std::set<std::string> sh = shadow;
                sh.insert(s->var);
                Out body;
                std::set<std::string> pr = primed;
                for (const auto &st : s->blocks[0]) nested(st, body, pr, sh, col + 5);
                if (body.items.empty()) body.items.push_back(line("TRUE"));
                auto n = std::make_shared<Node>();
                if (s->with_eq) {
                    n->k = Node::LINE;  // LET x == e IN /\ ...
                    std::string txt = "LET " + s->var + " == " + pe(s->e, c, primed, shadow) + " IN";
                    std::string bodytxt;
                    render(wrap(body), col + 5, bodytxt);
                    n->text = txt + "\n" + std::string((size_t)(col + 5 - 1), ' ') + bodytxt;
                } else {
                    n->k = Node::EXISTS;
                    n->text = s->var + " \\in " + pe(s->e, c, primed, shadow);
                    n->kids = {wrap(body)};
                }
                o.items.push_back(n);
                o.assigned.insert(body.assigned.begin(), body.assigned.end());
                primed.insert(body.assigned.begin(), body.assigned.end());
                break;
            }
            default: simple(s, o, primed, shadow, col); break;
            }
        }
        o.items.push_back(line(pc_set(cont)));
    }
    // a statement nested in a label-free compound statement
    void nested(const SP &s, Out &o, std::set<std::string> &primed, const std::set<std::string> &shadow, int col) {
        if (s->k == Stmt::GOTO || s->k == Stmt::WHILE)
            throw TranslateError{std::string(s->k == Stmt::GOTO ? "goto" : "while") + " at line " + std::to_string(s->pos.line) +
                                 " must be the last statement of its step (put a label after the enclosing statement)"};
        if (s->k == Stmt::IF || s->k == Stmt::EITHER || s->k == Stmt::WITH) {
            std::vector<SP> one{s};
            Out tmp;
            // reuse seq() on a one-statement sequence, then drop the trailing pc' it appends
            seq(one, 0, "?", false, tmp, primed, shadow, col);
            tmp.items.pop_back();
            for (auto &it : tmp.items) o.items.push_back(it);
            o.assigned.insert(tmp.assigned.begin(), tmp.assigned.end());
            primed.insert(tmp.assigned.begin(), tmp.assigned.end());
            return;
        }
        simple(s, o, primed, shadow, col);
    }
    void branch_unchanged(const std::vector<Out *> &bs, int col) {
        std::set<std::string> all;
        for (auto *b : bs) all.insert(b->assigned.begin(), b->assigned.end());
        for (auto *b : bs) {
            std::set<std::string> missing;
            for (const auto &v : all) if (!b->assigned.count(v)) missing.insert(v);
            if (missing.empty()) continue;
            // goes before a trailing pc' conjunct? the translator appends it after the branch's statements
            b->items.push_back(line(unchanged(missing, col + 3)));
        }
    }
    static NP wrap(const Out &o) {
        auto n = conj();
        n->kids = o.items;
        return n;
    }
};

struct LabelSite { const std::vector<SP> *seq; size_t idx; std::string cont; };

void collect_labels(const std::vector<SP> &v, const std::string &cont, std::vector<LabelSite> &out) {
    for (size_t j = 0; j < v.size(); j++) {
        const SP &s = v[j];
        const std::string after = j + 1 < v.size() ? v[j + 1]->label : cont;
        if (!s->label.empty()) out.push_back({&v, j, cont});
        for (const auto &b : s->blocks) collect_labels(b, s->k == Stmt::WHILE ? s->label : after, out);
    }
}

}  // namespace

std::string translate(const Module &m) {
    std::string o = "\\* BEGIN TRANSLATION\n";
    const bool multi = !(m.procs.size() == 1 && m.procs[0].name.empty());
    std::vector<std::string> vars;
    for (const auto &g : m.globals) vars.push_back(g.name);
    vars.push_back("pc");
    for (const auto &p : m.procs) for (const auto &l : p.locals) vars.push_back(l.name);
    auto join = [](const std::vector<std::string> &v, const char *sep) { std::string s; for (size_t i = 0; i < v.size(); i++) s += (i ? sep : "") + v[i]; return s; };
    bool any_default = false;
    for (const auto &g : m.globals) any_default |= g.no_init;
    for (const auto &p : m.procs) for (const auto &l : p.locals) any_default |= l.no_init;
    if (any_default) o += "CONSTANT defaultInitValue\n";  // p-manual App. B p.60: omitted if every variable is initialised
    Ctx none;
    none.m = &m;
    const std::set<std::string> empty;
    bool has_define = false;
    for (const auto &d : m.defs) has_define |= d.in_define;
    if (!has_define) {
        o += "VARIABLES " + join(vars, ", ") + "\n\n";
    } else {
        // p-manual App. B p.60: with a define statement there are two VARIABLES statements — the global variables and pc
        // first, then the definitions (which may mention those), then the remaining (process-local) variables
        std::vector<std::string> first, rest;
        for (const auto &g : m.globals) first.push_back(g.name);
        first.push_back("pc");
        for (const auto &p : m.procs) for (const auto &l : p.locals) rest.push_back(l.name);
        o += "VARIABLES " + join(first, ", ") + "\n\n(* define statement *)\n";
        for (const auto &d : m.defs) {
            if (!d.in_define) continue;
            std::set<std::string> sh(d.params.begin(), d.params.end());
            o += d.name + (d.params.empty() ? "" : "(" + join(d.params, ", ") + ")") + " == " + pe(d.body, none, empty, sh) + "\n\n";
        }
        if (!rest.empty()) o += "VARIABLES " + join(rest, ", ") + "\n\n";
    }
    o += "vars == << " + join(vars, ", ") + " >>\n\n";
    if (std::any_of(m.records.begin(), m.records.end(), [](const RecordVar &r) { return !r.set; })) {  // pcal.h, RECORDS: the record as the text around the algorithm knows it
        o += "(* record variables are kept field by field: r.f is r_f *)\n";
        for (const auto &r : m.records) {
            if (r.set) continue;   // a set of records stays the variable it is
            const bool per_process = r.proc >= 0 && multi && m.procs[(size_t)r.proc].is_set;
            if (r.seq) {   // q == [n_ \in 1..Len(q_f) |-> [f |-> q_f[n_], ...]]: a function on 1..n IS a sequence
                const std::string at = std::string(per_process ? "[self]" : "") + (r.array ? "[" + r.bound + "]" : "");
                std::string rec = "[n_ \\in 1..Len(" + r.name + "_" + r.fields[0] + at + ") |-> [";
                for (size_t i = 0; i < r.fields.size(); i++) rec += (i ? ", " : "") + r.fields[i] + " |-> " + r.name + "_" + r.fields[i] + at + "[n_]";
                rec += "]]";
                if (r.array) { std::set<std::string> sh{r.bound}; rec = "[" + r.bound + " \\in " + pe(r.domain, none, empty, sh) + " |-> " + rec + "]"; }
                if (per_process) rec = "[self \\in " + pe(m.procs[(size_t)r.proc].id, none, empty, empty) + " |-> " + rec + "]";
                o += r.name + " == " + rec + "\n";
                continue;
            }
            const std::string at = r.array ? "[" + r.bound + "]" : "";
            std::string rec = "[";
            for (size_t i = 0; i < r.fields.size(); i++) rec += (i ? ", " : "") + r.fields[i] + " |-> " + r.name + "_" + r.fields[i] + (per_process ? "[self]" : "") + at;
            rec += "]";
            if (r.array) { std::set<std::string> sh{r.bound}; rec = "[" + r.bound + " \\in " + pe(r.domain, none, empty, sh) + " |-> " + rec + "]"; }
            if (per_process) rec = "[self \\in " + pe(m.procs[(size_t)r.proc].id, none, empty, empty) + " |-> " + rec + "]";
            o += r.name + " == " + rec + "\n";
        }
        o += "\n";
    }
    if (multi) {
        std::vector<std::string> parts;
        for (const auto &p : m.procs) parts.push_back(p.is_set ? "(" + pe(p.id, none, empty, empty) + ")" : "{" + pe(p.id, none, empty, empty) + "}");
        o += "ProcSet == " + join(parts, " \\cup ") + "\n\n";
    }
    // ---- Init
    {
        std::string pad(8, ' ');
        o += "Init == ";
        bool first = true;
        auto item = [&](const std::string &s) { o += (first ? "" : pad) + s + "\n"; first = false; };
        if (!m.globals.empty()) {
            item("(* Global variables *)");
            for (const auto &g : m.globals) item("/\\ " + g.name + (g.in_set ? " \\in " : " = ") + pe(g.init, none, empty, empty));
        }
        for (const auto &p : m.procs) {
            if (p.locals.empty()) continue;
            if (multi) item("(* Process " + p.name + " *)");
            Ctx c = none;
            c.proc = &p;
            const std::string ids = p.id ? pe(p.id, none, empty, empty) : std::string();   // (a uniprocess algorithm has no identifier: its procedures' variables are plain variables)
            for (const auto &l : p.locals) {
                const std::string e = pe(l.init, none, empty, empty);
                if (!multi || !p.is_set) item("/\\ " + l.name + (l.in_set ? " \\in " : " = ") + e);
                else if (l.in_set) item("/\\ " + l.name + " \\in [" + ids + " -> " + e + "]");
                else item("/\\ " + l.name + " = [self \\in " + ids + " |-> " + e + "]");
            }
        }
        auto first_label = [&](const Proc &p) -> std::string {
            if (p.body.empty() || p.body[0]->label.empty()) throw TranslateError{"the first statement of " + (p.name.empty() ? std::string("the algorithm") : "process " + p.name) + " needs a label"};
            return p.body[0]->label;
        };
        try {
            if (!multi) item("/\\ pc = \"" + first_label(m.procs[0]) + "\"");
            else if (m.procs.size() == 1) item("/\\ pc = [self \\in ProcSet |-> \"" + first_label(m.procs[0]) + "\"]");
            else {
                const std::string head = "/\\ pc = [self \\in ProcSet |-> CASE ";
                std::string s = head;
                for (size_t k = 0; k < m.procs.size(); k++) {
                    const auto &p = m.procs[k];
                    if (k) s += "\n" + pad + std::string(head.size() - 3, ' ') + "[] ";
                    s += std::string("self ") + (p.is_set ? "\\in " : "= ") + pe(p.id, none, empty, empty) + " -> \"" + first_label(p) + "\"";
                }
                item(s + "]");
            }
        } catch (const TranslateError &e) {
            return "\\* TRANSLATION ERROR: " + e.msg + "\n";
        }
        o += "\n";
    }
    // ---- actions
    try {
        for (const auto &p : m.procs) {
            Ctx c = none;
            c.proc = &p;
            c.multi = multi;
            for (const auto &l : p.locals) c.locals.insert(l.name);
            c.self = !multi ? "" : p.is_set ? "self" : pe(p.id, none, empty, empty);
            ActionGen g(m, c);
            std::vector<LabelSite> sites;
            collect_labels(p.body, "Done", sites);
            std::vector<std::string> names;
            for (const auto &site : sites) {
                const SP &s = (*site.seq)[site.idx];
                const std::string head = s->label + (multi && p.is_set ? "(self)" : "") + " == ";
                const int col = (int)head.size() + 1;
                ActionGen::Out out;
                out.items.push_back(line(g.pc_ref() + " = \"" + s->label + "\""));
                g.seq(*site.seq, site.idx, site.cont, true, out, {}, {}, col);
                std::set<std::string> rest;
                for (const auto &v : g.var_order) if (!out.assigned.count(v)) rest.insert(v);
                if (!rest.empty()) out.items.push_back(line(g.unchanged(rest, col + 3)));
                std::string body;
                render(ActionGen::wrap(out), col, body);
                o += head + body + "\n\n";
                names.push_back(s->label + (multi && p.is_set ? "(self)" : ""));
            }
            if (multi) o += p.name + (p.is_set ? "(self)" : "") + " == " + join(names, " \\/ ") + "\n\n";
            else {
                o += "Next == " + join(names, " \\/ ") + "\n";
                o += "           \\/ (* Disjunct to prevent deadlock on termination *)\n";
                o += "              (pc = \"Done\" /\\ UNCHANGED vars)\n\n";
            }
        }
    } catch (const TranslateError &e) {
        return "\\* TRANSLATION ERROR: " + e.msg + "\n";
    }
    if (multi) {
        std::vector<std::string> dis;
        for (const auto &p : m.procs) if (!p.is_set) dis.push_back(p.name);
        for (const auto &p : m.procs) if (p.is_set) dis.push_back("(\\E self \\in " + pe(p.id, none, empty, empty) + ": " + p.name + "(self))");
        o += "Next == " + dis[0] + "\n";
        for (size_t k = 1; k < dis.size(); k++) o += "           \\/ " + dis[k] + "\n";
        o += "           \\/ (* Disjunct to prevent deadlock on termination *)\n";
        o += "              ((\\A self \\in ProcSet: pc[self] = \"Done\") /\\ UNCHANGED vars)\n\n";
    }
    o += "Spec == Init /\\ [][Next]_vars\n\n";
    o += multi ? "Termination == <>(\\A self \\in ProcSet: pc[self] = \"Done\")\n\n" : "Termination == <>(pc = \"Done\")\n\n";
    o += "\\* END TRANSLATION\n";
    return o;
}

std::string transpile_text(const std::string &text, const Module &m) {
    const std::string tr = translate(m);
    if (m.has_translation) {
        const size_t b = find_line_start(text, m.tr_first_line);
        size_t e = find_line_start(text, m.tr_last_line + 1);
        return text.substr(0, b) + tr + text.substr(e);
    }
    const size_t after = find_line_start(text, m.alg_last_line + 1);
    return text.substr(0, after) + tr + text.substr(after);
}

}  // namespace pcal
