// frontend.cpp — host-side front-end of libtlamc.so: TLC .cfg parser, spec registry lookup,
// `tlc X.tla` end-to-end driver and TLC-format report.
//
// Replaces, for the lowered specs only, what the external TLC tool does before and after its
// BFS (reference Makefile:6-7, README.md:262):
//   * reading X.cfg — grammar examples/SpecifyingSystems/TLC/ConfigFileGrammar.tla:4-32, plus
//     the instance-scoped override `Id <-[Module] Id` used by examples/Paxos/MCPaxos.cfg:9 and
//     both comment styles (pcal_intro.cfg:1 `\*`, AsynchInterface.cfg:1-5 `(* *)`);
//   * choosing Init/Next/invariants/constraint from it (pcal_intro.cfg:2-3);
//   * printing the report (README.md:267-321, testout2:260-266).
// There is no general TLA+ evaluator: a TLA+-only module is matched by name to a
// hand-lowered spec and its text is fingerprinted so a changed spec is refused, not mis-checked.  A module with a
// PlusCal algorithm that has no hand lowering is compiled (pcal.cpp, pcal_compile.cpp) and runs through the
// bytecode interpreter of spec_vm.h; `mc --transpile` is the reference's `pcal2tla` step (Makefile:3-4).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/tlamc.h"
#include "pcal.h"
#include "tlaeval.h"
#include "spec_vm.h"

extern "C" void mc_set_error_internal(const char *msg);  // engine.hip

struct mc_program {
    pcal::Program prog;
};
extern "C" long pcal_codegen_text(const pcal::Program *p, char *buf, size_t cap);   // pcal_codegen.cpp

namespace {


struct CfgValue {
    enum Kind { IDENT, NUMBER, STRING, SET } kind = IDENT;
    std::string text;
    long long num = 0;
    std::vector<CfgValue> elems;
};
struct CfgConst {
    std::string name;
    bool replacement = false;  // `<-` (definition override) vs `=` (value)
    std::string module;        // `<-[Module]`
    std::string target;
    CfgValue value;
};

}  // namespace

struct mc_cfg {
    std::string specification, init, next, view, symmetry;
    std::vector<std::string> invariants, properties, constraints, action_constraints;
    std::vector<CfgConst> constants;
    int check_deadlock = -1;   // TLC's `CHECK_DEADLOCK TRUE | FALSE` statement (-1: not given; the command line's -deadlock decides)
};

namespace {

struct Tok {
    enum T { END, IDENT, NUMBER, STRING, LBRACE, RBRACE, COMMA, EQ, ARROW, ARROW_MOD } t = END;
    std::string s;
    int line = 0;
};

struct Lexer {
    const char *p, *e;
    int line = 1;
    std::string err;
    Lexer(const char *b, size_t n) : p(b), e(b + n) {}
    static bool idch(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }
    bool skip() {
        for (;;) {
            while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n')) { if (*p == '\n') line++; p++; }
            if (p + 1 < e && p[0] == '\\' && p[1] == '*') { while (p < e && *p != '\n') p++; continue; }
            if (p + 1 < e && p[0] == '(' && p[1] == '*') {
                int depth = 1;
                p += 2;
                while (p < e && depth) {
                    if (p + 1 < e && p[0] == '(' && p[1] == '*') { depth++; p += 2; }
                    else if (p + 1 < e && p[0] == '*' && p[1] == ')') { depth--; p += 2; }
                    else { if (*p == '\n') line++; p++; }
                }
                if (depth) { err = "unterminated (* comment"; return false; }
                continue;
            }
            return true;
        }
    }
    bool next(Tok &t) {
        if (!skip()) return false;
        t = Tok();
        t.line = line;
        if (p >= e) { t.t = Tok::END; return true; }
        const char c = *p;
        if (idch(c)) {
            const char *b = p;
            while (p < e && idch(*p)) p++;
            t.s.assign(b, p);
            // ACTION-CONSTRAINT(S) is the one keyword with a hyphen (ConfigFileGrammar.tla:9-10)
            if (t.s == "ACTION" && p < e && *p == '-') {
                const char *q = p + 1;
                while (q < e && idch(*q)) q++;
                std::string tail(p + 1, q);
                if (tail == "CONSTRAINT" || tail == "CONSTRAINTS") { t.s += "-" + tail; p = q; }
            }
            bool letter = false;
            for (char ch : t.s) letter |= !(ch >= '0' && ch <= '9');
            t.t = letter ? Tok::IDENT : Tok::NUMBER;
            return true;
        }
        if (c == '-' && p + 1 < e && p[1] >= '0' && p[1] <= '9') {
            const char *b = p++;
            while (p < e && *p >= '0' && *p <= '9') p++;
            t.s.assign(b, p);
            t.t = Tok::NUMBER;
            return true;
        }
        if (c == '"') {
            const char *b = ++p;
            while (p < e && *p != '"' && *p != '\n') p++;
            if (p >= e || *p != '"') { err = "unterminated string"; return false; }
            t.s.assign(b, p++);
            t.t = Tok::STRING;
            return true;
        }
        if (c == '{') { p++; t.t = Tok::LBRACE; return true; }
        if (c == '}') { p++; t.t = Tok::RBRACE; return true; }
        if (c == ',') { p++; t.t = Tok::COMMA; return true; }
        if (c == '=') { p++; t.t = Tok::EQ; return true; }
        if (c == '<' && p + 1 < e && p[1] == '-') {
            p += 2;
            if (p < e && *p == '[') {
                const char *b = ++p;
                while (p < e && idch(*p)) p++;
                if (p >= e || *p != ']' || b == p) { err = "malformed <-[Module]"; return false; }
                t.s.assign(b, p++);
                t.t = Tok::ARROW_MOD;
            } else {
                t.t = Tok::ARROW;
            }
            return true;
        }
        err = std::string("unexpected character '") + c + "'";
        return false;
    }
};

bool is_singular(const std::string &s) { return s == "SPECIFICATION" || s == "INIT" || s == "NEXT" || s == "VIEW" || s == "SYMMETRY"; }
bool is_plural(const std::string &s) {
    return s == "CONSTRAINT" || s == "CONSTRAINTS" || s == "ACTION-CONSTRAINT" || s == "ACTION-CONSTRAINTS" || s == "INVARIANT" ||
           s == "INVARIANTS" || s == "PROPERTY" || s == "PROPERTIES";
}
bool is_keyword(const std::string &s) { return is_singular(s) || is_plural(s) || s == "CONSTANT" || s == "CONSTANTS" || s == "CHECK_DEADLOCK"; }

struct Parser {
    Lexer lx;
    Tok cur;
    std::string err;
    Parser(const char *b, size_t n) : lx(b, n) {}
    bool adv() {
        if (!lx.next(cur)) { err = "line " + std::to_string(lx.line) + ": " + lx.err; return false; }
        return true;
    }
    bool fail(const std::string &m) { err = "line " + std::to_string(cur.line) + ": " + m; return false; }
    bool value(CfgValue &v) {
        if (cur.t == Tok::IDENT) { v.kind = CfgValue::IDENT; v.text = cur.s; return adv(); }
        if (cur.t == Tok::NUMBER) { v.kind = CfgValue::NUMBER; v.text = cur.s; v.num = atoll(cur.s.c_str()); return adv(); }
        if (cur.t == Tok::STRING) { v.kind = CfgValue::STRING; v.text = cur.s; return adv(); }
        if (cur.t == Tok::LBRACE) {
            v.kind = CfgValue::SET;
            if (!adv()) return false;
            if (cur.t == Tok::RBRACE) return adv();
            for (;;) {
                CfgValue e;
                if (!value(e)) return false;
                v.elems.push_back(e);
                if (cur.t == Tok::COMMA) { if (!adv()) return false; continue; }
                if (cur.t == Tok::RBRACE) return adv();
                return fail("expected ',' or '}' in set value");
            }
        }
        return fail("expected a value (identifier, number, string or {set})");
    }
    bool parse(mc_cfg &c) {
        if (!adv()) return false;
        while (cur.t != Tok::END) {
            if (cur.t != Tok::IDENT || !is_keyword(cur.s)) return fail("expected a configuration keyword, got '" + cur.s + "'");
            const std::string kw = cur.s;
            if (!adv()) return false;
            if (kw == "CHECK_DEADLOCK") {   // TLC2's cfg statement (not in the 2001 grammar of TLC/ConfigFileGrammar.tla:4-32): same as -deadlock
                if (cur.t != Tok::IDENT || (cur.s != "TRUE" && cur.s != "FALSE")) return fail("CHECK_DEADLOCK must be followed by TRUE or FALSE");
                c.check_deadlock = cur.s == "TRUE";
                if (!adv()) return false;
            } else if (is_singular(kw)) {
                if (cur.t != Tok::IDENT || is_keyword(cur.s)) return fail(kw + " must be followed by an identifier");
                std::string &dst = kw == "SPECIFICATION" ? c.specification : kw == "INIT" ? c.init : kw == "NEXT" ? c.next
                                   : kw == "VIEW" ? c.view : c.symmetry;
                dst = cur.s;
                if (!adv()) return false;
            } else if (is_plural(kw)) {
                std::vector<std::string> &dst = kw[0] == 'I' ? c.invariants : kw[0] == 'P' ? c.properties
                                                : kw[0] == 'A' ? c.action_constraints : c.constraints;
                while (cur.t == Tok::IDENT && !is_keyword(cur.s)) {
                    dst.push_back(cur.s);
                    if (!adv()) return false;
                }
            } else {  // CONSTANT(S): (Replacement | Assignment)*
                while (cur.t == Tok::IDENT && !is_keyword(cur.s)) {
                    CfgConst k;
                    k.name = cur.s;
                    if (!adv()) return false;
                    if (cur.t == Tok::ARROW || cur.t == Tok::ARROW_MOD) {
                        k.replacement = true;
                        if (cur.t == Tok::ARROW_MOD) k.module = cur.s;
                        if (!adv()) return false;
                        if (cur.t != Tok::IDENT) return fail("'<-' must be followed by an identifier");
                        k.target = cur.s;
                        if (!adv()) return false;
                    } else if (cur.t == Tok::EQ) {
                        if (!adv()) return false;
                        if (!value(k.value)) return false;
                    } else {
                        return fail("expected '=' or '<-' after constant '" + k.name + "'");
                    }
                    c.constants.push_back(k);
                }
            }
        }
        return true;
    }
};

void json_str(std::string &o, const std::string &s) {
    o += '"';
    for (char ch : s) {
        if (ch == '"' || ch == '\\') { o += '\\'; o += ch; }
        else if (ch == '\n') o += "\\n";
        else o += ch;
    }
    o += '"';
}
void json_list(std::string &o, const std::vector<std::string> &v) {
    o += '[';
    for (size_t i = 0; i < v.size(); i++) { if (i) o += ", "; json_str(o, v[i]); }
    o += ']';
}
void json_value(std::string &o, const CfgValue &v) {
    switch (v.kind) {
    case CfgValue::IDENT: o += "{\"model_value\": "; json_str(o, v.text); o += "}"; break;
    case CfgValue::NUMBER: o += std::to_string(v.num); break;
    case CfgValue::STRING: json_str(o, v.text); break;
    case CfgValue::SET:
        o += "{\"set\": [";
        for (size_t i = 0; i < v.elems.size(); i++) { if (i) o += ", "; json_value(o, v.elems[i]); }
        o += "]}";
        break;
    }
}

const CfgConst *find_const(const mc_cfg *c, const char *name) {
    const CfgConst *r = nullptr;
    for (const auto &k : c->constants) if (k.name == name) r = &k;  // last one wins
    return r;
}
bool const_int(const mc_cfg *c, const char *name, long long &out) {
    const CfgConst *k = find_const(c, name);
    if (!k || k->replacement || k->value.kind != CfgValue::NUMBER) return false;
    out = k->value.num;
    return true;
}
int fe_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    mc_set_error_internal(buf);
    return code;
}

// text fingerprints of the reference specs the lowerings were written against (whitespace removed, FNV-1a)
uint64_t text_hash(const std::string &s) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (unsigned char ch : s) {
        if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') continue;
        h ^= ch;
        h *= 0x100000001b3ull;
    }
    return h;
}
constexpr uint64_t H_PCAL_INTRO = 0x6da4921b5bd7b79aull;        // pcal_intro.tla:4-19 (--algorithm .. end algorithm)
constexpr uint64_t H_PCAL_INTRO_README = 0x446ad8dac291d64full; // README.md:224-240 (labels A:, B:)
constexpr uint64_t H_ATOMIC_ADD = 0x6c3a51af80fccd40ull;        // atomic_add.tla:4-23
constexpr uint64_t H_RAFT = 0x289fe41014391a24ull;              // examples/raft.tla:8-517 (EXTENDS .. before ====)
constexpr uint64_t H_TEXTBOOK_SI = 0x26b4e8333db4314cull;      // examples/textbookSnapshotIsolation.tla (EXTENDS .. before ====)
constexpr uint64_t H_SSI = 0x85c02cbaf85b39ceull;              // examples/serializableSnapshotIsolation.tla:21-1579
constexpr uint64_t H_ATOMIC_ADD_N = 0x586f8fd08bcb696bull;      // specs/atomic_add_n.tla (--algorithm .. end algorithm)
// the model wrappers of this repo (EXTENDS .. before ====): StateConstraint, the invariant definitions, Perms
constexpr uint64_t H_MCRAFT = 0xa9171a6df0ae47d0ull;            // specs/MCraft.tla
constexpr uint64_t H_MCSSI = 0xefcc77c2546e41bfull;             // specs/MCssi.tla
constexpr uint64_t H_MCTEXTBOOK_SI = 0xe66bad24776b2048ull;     // specs/MCtextbookSI.tla
constexpr uint64_t H_VOTING = 0xa61c2608a8ec9233ull;            // examples/Paxos/Voting.tla:6-199 (EXTENDS .. before ====)
constexpr uint64_t H_PAXOS = 0x979510bc1153e9fbull;             // examples/Paxos/Paxos.tla
constexpr uint64_t H_CONSENSUS = 0x96d890afa5b1ac82ull;         // examples/Paxos/Consensus.tla

bool algorithm_text(const std::string &t, std::string &out) {
    const size_t i = t.find("--algorithm");
    if (i == std::string::npos) return false;
    const size_t j = t.find("end algorithm", i);
    if (j != std::string::npos) { out = t.substr(i, j + strlen("end algorithm") - i); return true; }
    // c-syntax: the algorithm runs to the end of its (* ... *) comment
    int depth = 1;
    for (size_t k = i; k + 1 < t.size(); k++) {
        if (t[k] == '(' && t[k + 1] == '*') { depth++; k++; }
        else if (t[k] == '*' && t[k + 1] == ')') { if (--depth == 0) { out = t.substr(i, k - i); return true; } k++; }
    }
    return false;
}
bool module_body(const std::string &t, std::string &out) {
    const size_t i = t.find("EXTENDS");
    if (i == std::string::npos) return false;
    size_t pos = i;
    while (pos < t.size()) {  // a line made of '=' only ends the module
        size_t eol = t.find('\n', pos);
        if (eol == std::string::npos) eol = t.size();
        size_t a = pos, b = eol;
        while (a < b && (t[a] == ' ' || t[a] == '\r')) a++;
        while (b > a && (t[b - 1] == ' ' || t[b - 1] == '\r')) b--;
        if (b - a >= 4 && t.find_first_not_of('=', a) >= b) { out = t.substr(i, pos - i); return true; }
        pos = eol + 1;
    }
    return false;
}
bool read_file(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}
bool module_name(const std::string &t, std::string &name) {
    const size_t i = t.find("MODULE");
    if (i == std::string::npos) return false;
    size_t a = i + 6;
    while (a < t.size() && (t[a] == ' ' || t[a] == '\t')) a++;
    size_t b = a;
    while (b < t.size() && Lexer::idch(t[b])) b++;
    if (b == a) return false;
    name = t.substr(a, b - a);
    return true;
}
std::string dir_of(const std::string &p) {
    const size_t i = p.find_last_of('/');
    return i == std::string::npos ? "." : p.substr(0, i);
}

// ---- source locations, as TLC prints them (README.md:278 "<Action line 35, col 19 to line 40, col 42 of module pcal_intro>")
struct Span { int l1 = 0, c1 = 0, l2 = 0, c2 = 0; bool ok = false; };
std::vector<std::string> split_lines(const std::string &t) {
    std::vector<std::string> v;
    size_t pos = 0;
    while (pos <= t.size()) {
        size_t eol = t.find('\n', pos);
        if (eol == std::string::npos) eol = t.size();
        std::string line = t.substr(pos, eol - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        v.push_back(line);
        pos = eol + 1;
    }
    return v;
}
int last_nonspace(const std::string &l) {  // 1-based column of the last non-blank character, 0 if none
    int c = (int)l.size();
    while (c > 0 && (l[c - 1] == ' ' || l[c - 1] == '\t')) c--;
    return c;
}
// body of the top-level definition `name(...) ==` / `name ==`: from the first token after "==" to the last
// token before the next blank line or top-level definition
Span definition_span(const std::vector<std::string> &L, const std::string &name, int *first_line = nullptr) {
    Span sp;
    for (size_t i = 0; i < L.size(); i++) {
        const std::string &l = L[i];
        if (l.compare(0, name.size(), name) != 0) continue;
        const char nx = l.size() > name.size() ? l[name.size()] : 0;
        if (nx != '(' && nx != ' ') continue;
        const size_t eq = l.find("==");
        if (eq == std::string::npos) continue;
        size_t b = eq + 2;
        while (b < l.size() && l[b] == ' ') b++;
        if (b >= l.size()) continue;
        sp.l1 = (int)i + 1;
        sp.c1 = (int)b + 1;
        size_t j = i;
        while (j + 1 < L.size() && last_nonspace(L[j + 1]) > 0 && (L[j + 1][0] == ' ' || L[j + 1][0] == '\t')) j++;
        sp.l2 = (int)j + 1;
        sp.c2 = last_nonspace(L[j]);
        sp.ok = true;
        if (first_line) *first_line = (int)i;
        return sp;
    }
    return sp;
}

struct Out {
    char *b; size_t cap, k;
    void put(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        va_list ap;
        va_start(ap, fmt);
        if (k + 1 < cap) { int w = vsnprintf(b + k, cap - k, fmt, ap); if (w > 0) k += (size_t)w; if (k >= cap) k = cap - 1; }
        va_end(ap);
    }
};

}  // namespace

extern "C" {

int mc_cfg_parse(const char *text, size_t len, mc_cfg **out) {
    if (!text || !out) return MC_EBADCFG;
    *out = nullptr;
    mc_cfg *c = new mc_cfg();
    Parser p(text, len);
    if (!p.parse(*c)) {
        mc_set_error_internal(("cfg: " + p.err).c_str());
        delete c;
        return MC_EPARSE;
    }
    *out = c;
    return MC_OK;
}
void mc_cfg_free(mc_cfg *c) { delete c; }

int mc_cfg_json(const mc_cfg *c, char *buf, size_t cap) {
    if (!c || !buf || !cap) return MC_EBADCFG;
    std::string o = "{";
    o += "\"SPECIFICATION\": "; json_str(o, c->specification);
    o += ", \"INIT\": "; json_str(o, c->init);
    o += ", \"NEXT\": "; json_str(o, c->next);
    o += ", \"VIEW\": "; json_str(o, c->view);
    o += ", \"SYMMETRY\": "; json_str(o, c->symmetry);
    if (c->check_deadlock >= 0) o += std::string(", \"CHECK_DEADLOCK\": ") + (c->check_deadlock ? "true" : "false");
    o += ", \"INVARIANTS\": "; json_list(o, c->invariants);
    o += ", \"PROPERTIES\": "; json_list(o, c->properties);
    o += ", \"CONSTRAINTS\": "; json_list(o, c->constraints);
    o += ", \"ACTION_CONSTRAINTS\": "; json_list(o, c->action_constraints);
    o += ", \"CONSTANTS\": [";
    for (size_t i = 0; i < c->constants.size(); i++) {
        const CfgConst &k = c->constants[i];
        if (i) o += ", ";
        o += "{\"name\": "; json_str(o, k.name);
        if (k.replacement) {
            o += ", \"replace_by\": "; json_str(o, k.target);
            if (!k.module.empty()) { o += ", \"module\": "; json_str(o, k.module); }
        } else {
            o += ", \"value\": "; json_value(o, k.value);
        }
        o += "}";
    }
    o += "]}";
    if (o.size() + 1 > cap) return fe_fail(MC_EBADCFG, "mc_cfg_json: buffer too small");
    memcpy(buf, o.c_str(), o.size() + 1);
    return (int)o.size();
}

// Registry: module name -> lowering (+ constants taken from the cfg).
int mc_spec_resolve(const char *module, const mc_cfg *c, mc_spec_desc *out) {
    if (!module || !c || !out) return MC_EBADCFG;
    memset(out, 0, sizeof *out);
    const std::string m = module;
    if (!c->properties.empty()) return fe_fail(MC_ENOSPEC, "PROPERTY (temporal) checking is not supported; only INVARIANT safety checking");
    if (!c->view.empty()) return fe_fail(MC_ENOSPEC, "VIEW is not supported");
    const bool is_ssi = m == "MCssi" || m == "serializableSnapshotIsolation" || m == "MCtextbookSI" || m == "textbookSnapshotIsolation";
    if (!c->symmetry.empty() && !is_ssi)
        return fe_fail(MC_ENOSPEC, "SYMMETRY is supported for the snapshot-isolation models only (their run-book makes Key and TxnId symmetry sets)");
    // the hand lowerings of the PlusCal root specs know no CONSTRAINT: MC_ENOSPEC sends the module through the compiled-program
    // path, which evaluates the constraint (or refuses an ACTION-CONSTRAINT) instead of silently ignoring it
    if ((m == "atomic_add" || m == "atomic_add_n" || m == "pcal_intro") && (!c->constraints.empty() || !c->action_constraints.empty()))
        return fe_fail(MC_ENOSPEC, "the hand lowering of %s takes no CONSTRAINT / ACTION-CONSTRAINT", module);
    if (m == "atomic_add" || m == "atomic_add_n") {  // atomic_add.tla:4-23; atomic_add_n: N adders (specs/atomic_add_n.tla)
        long long n = 2;
        if (m == "atomic_add_n" && !const_int(c, "N", n)) return fe_fail(MC_EBADCFG, "atomic_add_n needs CONSTANT N = <number>");
        if (!c->invariants.empty()) return fe_fail(MC_ENOSPEC, "atomic_add defines no invariant named '%s'", c->invariants[0].c_str());
        out->spec_id = MC_SPEC_ATOMIC_ADD;
        out->nparams = 1;
        out->params[0] = n;
        return MC_OK;
    }
    if (m == "pcal_intro") {  // pcal_intro.tla:4-23 + pcal_intro.cfg:2-3
        long long inv = 0;
        for (const auto &i : c->invariants) {
            if (i == "MoneyInvariant") inv = 1;
            else return fe_fail(MC_ENOSPEC, "pcal_intro defines no invariant named '%s'", i.c_str());
        }
        out->spec_id = MC_SPEC_PCAL_INTRO;
        out->nparams = 4;
        out->params[0] = 0;  // variant: refined from the module text by mc_check_files
        out->params[1] = inv;
        out->params[2] = 20;
        out->params[3] = 2;
        return MC_OK;
    }
    if (m == "MCraft" || m == "raft") {  // examples/raft.tla under specs/MCraft.tla
        const CfgConst *srv = find_const(c, "Server");
        if (!srv || srv->replacement || srv->value.kind != CfgValue::SET || srv->value.elems.empty())
            return fe_fail(MC_EBADCFG, "raft needs CONSTANT Server = {s1, ..., sn}");
        long long mcr = 0, mt = 0, ml = 0, mm = 0, mk = 0;
        if (!const_int(c, "MaxClientRequests", mcr)) return fe_fail(MC_EBADCFG, "raft needs CONSTANT MaxClientRequests = <number> (raft.tla:23-24)");
        bool has_constraint = false;
        for (const auto &k : c->constraints) {
            if (k == "StateConstraint") has_constraint = true;
            else return fe_fail(MC_ENOSPEC, "unknown CONSTRAINT '%s' (MCraft defines StateConstraint)", k.c_str());
        }
        if (!has_constraint)
            return fe_fail(MC_EBADCFG, "raft.tla has an unbounded term counter (raft.tla:199): the cfg must name CONSTRAINT StateConstraint");
        if (!const_int(c, "MaxTerm", mt) || !const_int(c, "MaxLogLen", ml) || !const_int(c, "MaxMsgs", mm) || !const_int(c, "MaxMsgKeys", mk))
            return fe_fail(MC_EBADCFG, "StateConstraint needs CONSTANTS MaxTerm, MaxLogLen, MaxMsgs, MaxMsgKeys");
        if (mk < 1 || mk > 64) return fe_fail(MC_EBADCFG, "MaxMsgKeys must be 1..64 (the packed state holds at most 64 message keys)");
        long long mask = 0;
        for (const auto &i : c->invariants) {
            if (i == "NoTwoLeaders") mask |= 1;
            else if (i == "CommittedLogStable") mask |= 2;
            else return fe_fail(MC_ENOSPEC, "MCraft defines no invariant named '%s'", i.c_str());
        }
        out->spec_id = MC_SPEC_RAFT;
        out->nparams = 10;  // [6..8] = slot-array capacities (0 = defaults), [9] = MaxMsgKeys
        out->params[0] = (long long)srv->value.elems.size();
        out->params[1] = mcr; out->params[2] = mt; out->params[3] = ml; out->params[4] = mm; out->params[5] = mask;
        out->params[6] = mk; out->params[7] = 0;  // a stored state holds at most MaxMsgKeys message keys: exactly that many slots
        out->params[8] = 0; out->params[9] = mk;
        return MC_OK;
    }
    const bool textbook = m == "MCtextbookSI" || m == "textbookSnapshotIsolation";  // examples/textbookSnapshotIsolation.tla
    if (m == "MCssi" || m == "serializableSnapshotIsolation" || textbook) {  // serializableSnapshotIsolation.tla under specs/MCssi.tla
        const CfgConst *tx = find_const(c, "TxnId"), *ky = find_const(c, "Key"), *nl = find_const(c, "NoLock");
        if (!tx || tx->replacement || tx->value.kind != CfgValue::SET || tx->value.elems.empty() || tx->value.elems.size() > 4)
            return fe_fail(MC_EBADCFG, "SSI needs CONSTANT TxnId = {T1, ...} with 1..4 model values");
        if (!ky || ky->replacement || ky->value.kind != CfgValue::SET || ky->value.elems.empty() || ky->value.elems.size() > 3)
            return fe_fail(MC_EBADCFG, "SSI needs CONSTANT Key = {K1, ...} with 1..3 model values");
        if (!nl) return fe_fail(MC_EBADCFG, "NoLock is an unbounded CHOOSE (serializableSnapshotIsolation.tla:24): the cfg must give it a model value, NoLock = NoLock");
        if (!c->constraints.empty()) return fe_fail(MC_ENOSPEC, "SSI defines no CONSTRAINT");
        long long mask = 0, find = 0;
        static const struct { const char *name; long long bit, find; } inv[] = {
            {"TypeInv", 0, 0}, {"WellFormed", 1, 0}, {"CorrectnessOfHoldingXLocks", 2, 0}, {"CorrectnessOfWaitingForXLock", 4, 0},
            {"CorrectReadView", 8, 0}, {"FirstCommitterWins", 16, 0}, {"SemanticsOfSnapshotIsolation", 24, 0}, {"CahillOK", 32, 0},
            {"BernsteinOK", 64, 0}, {"NoVoluntaryAbort", 0, 1}, {"NoFCWAbort", 0, 2}, {"NoDeadlockAbort", 0, 3}, {"NoCommitAbort", 0, 4},
            {"NoReadAbort", 0, 5}, {"NoWriteAbort", 0, 6}, {"NoTwoWaiters", 0, 7}};
        for (const auto &i : c->invariants) {
            bool ok = false;
            for (const auto &e : inv)
                if (i == e.name) {
                    ok = true;
                    mask |= e.bit;
                    if (e.find) { if (find) return fe_fail(MC_ENOSPEC, "only one of the 'expected to be violated' predicates per run"); find = e.find; }
                }
            if (!ok) return fe_fail(MC_ENOSPEC, "MCssi defines no invariant named '%s'", i.c_str());
        }
        out->spec_id = MC_SPEC_SSI;
        out->nparams = 6;
        // SYMMETRY <name>: which of Permutations(TxnId) / Permutations(Key) the set <name> holds is read from the module
        // text by mc_check_files (symmetry_sets); through this entry point both are assumed, as :38-44 prescribe.
        out->params[5] = c->symmetry.empty() ? 0 : 3;
        out->params[0] = (long long)tx->value.elems.size();
        out->params[1] = (long long)ky->value.elems.size();
        out->params[2] = mask;
        out->params[3] = find;
        out->params[4] = textbook ? 1 : 0;
        return MC_OK;
    }
    return fe_fail(MC_ENOSPEC, "module '%s' is not one of the lowered specs (atomic_add, atomic_add_n, pcal_intro, MCraft, MCssi; "
                   "models of examples/Paxos are resolved from the module text: mc_resolve_files / mc_check_files)", module);
}

// ---------------------------------------------------------------------------------- PlusCal programs
static pcal::ConstVal to_const(const CfgValue &v) {
    pcal::ConstVal c;
    switch (v.kind) {
    case CfgValue::NUMBER: c.k = pcal::ConstVal::INT; c.i = v.num; break;
    case CfgValue::STRING: c.k = pcal::ConstVal::STR; c.s = v.text; break;
    case CfgValue::IDENT:
        if (v.text == "TRUE" || v.text == "FALSE") { c.k = pcal::ConstVal::INT; c.i = v.text == "TRUE"; }
        else { c.k = pcal::ConstVal::STR; c.s = v.text; }  // a model value
        break;
    case CfgValue::SET:
        c.k = pcal::ConstVal::SET;
        for (const auto &e : v.elems) c.elems.push_back(to_const(e));
        break;
    }
    return c;
}

int mc_pcal_translate(const char *tla_text, char *out, size_t cap) {
    if (!tla_text) return MC_EBADCFG;
    const std::string text(tla_text);
    pcal::Module m;
    const std::string err = pcal::parse_module(text, m);
    if (!err.empty()) return fe_fail(MC_EPARSE, "PlusCal: %s", err.c_str());
    const std::string tr = pcal::translate(m);
    if (tr.rfind("\\* TRANSLATION ERROR: ", 0) == 0) return fe_fail(MC_EPARSE, "PlusCal: %s", tr.substr(22).c_str());
    const std::string all = pcal::transpile_text(text, m);
    if (out && cap) {
        const size_t n = all.size() < cap ? all.size() : cap - 1;
        memcpy(out, all.data(), n);
        out[n] = 0;
    }
    return (int)all.size();
}

int mc_program_compile(const char *tla_text, const char *cfg_text, mc_program **out) {
    if (!tla_text || !out) return MC_EBADCFG;
    *out = nullptr;
    pcal::Config cf;
    if (cfg_text) {
        mc_cfg *c = nullptr;
        const int rc = mc_cfg_parse(cfg_text, strlen(cfg_text), &c);
        if (rc) return rc;
        cf.invariants = c->invariants;
        cf.constraints = c->constraints;
        for (const auto &k : c->constants) {
            if (k.replacement) { mc_cfg_free(c); return fe_fail(MC_ENOSPEC, "CONSTANT %s <- ...: definition overrides are not supported for PlusCal programs", k.name.c_str()); }
            cf.constants.push_back({k.name, to_const(k.value)});
        }
        const bool other = !c->action_constraints.empty() || !c->symmetry.empty() || !c->view.empty();
        mc_cfg_free(c);
        if (other) return fe_fail(MC_ENOSPEC, "ACTION-CONSTRAINT / SYMMETRY / VIEW are not supported for PlusCal programs");
    }
    const std::string text(tla_text);
    pcal::Module m;
    std::string err = pcal::parse_module(text, m);
    if (!err.empty()) return fe_fail(MC_EPARSE, "PlusCal: %s", err.c_str());
    auto *p = new mc_program();
    err = pcal::compile(m, text, cf, p->prog);
    if (!err.empty()) { delete p; return fe_fail(MC_ENOSPEC, "PlusCal: %s", err.c_str()); }
    *out = p;
    return MC_OK;
}
int mc_program_spec(const mc_program *p, mc_spec_desc *out) {
    if (!p || !out) return MC_EBADCFG;
    memset(out, 0, sizeof *out);
    out->spec_id = MC_SPEC_PCAL;
    out->nparams = 1;
    out->params[0] = (int64_t)(intptr_t)&p->prog;
    return MC_OK;
}
const char *mc_program_translated(const mc_program *p) { return p ? p->prog.translated.c_str() : ""; }
long mc_program_codegen(const mc_program *p, char *buf, size_t cap) { return p ? pcal_codegen_text(&p->prog, buf, cap) : (long)MC_EBADCFG; }
const char *mc_program_invariant(const mc_program *p, int index) {
    return p && index >= 0 && (size_t)index < p->prog.invariants.size() ? p->prog.invariants[(size_t)index].c_str() : "?";
}
int mc_program_assert_pos(const mc_program *p, int index, int *line, int *col) {
    if (!p || index < 0 || (size_t)index >= p->prog.asserts.size()) return MC_EBADCFG;
    if (line) *line = p->prog.asserts[(size_t)index].line;
    if (col) *col = p->prog.asserts[(size_t)index].col;
    return MC_OK;
}
void mc_program_free(mc_program *p) { delete p; }

static const char *invariant_name(const mc_spec_desc *d, int idx) {
    if (d->spec_id == MC_SPEC_PCAL) {
        const pcal::Program *P = (const pcal::Program *)(intptr_t)d->params[0];
        return idx >= 0 && (size_t)idx < P->invariants.size() ? P->invariants[(size_t)idx].c_str() : "?";
    }
    if (d->spec_id == MC_SPEC_PCAL_INTRO) return "MoneyInvariant";
    if (d->spec_id == MC_SPEC_RAFT) return idx == 1 ? "CommittedLogStable" : "NoTwoLeaders";
    if (d->spec_id == MC_SPEC_PAXOS) {
        static const char *px[] = {"Inv1", "Inv2", "Inv3", "Inv4", "VotingSpecBar"}, *vt[] = {"Inv", "ConsensusSpecBar"};
        if (d->params[0] == 1) return idx >= 0 && idx < 2 ? vt[idx] : "?";
        return idx >= 0 && idx < 5 ? px[idx] : "?";
    }
    if (d->spec_id == MC_SPEC_SSI) {
        static const char *nm[] = {"WellFormed", "CorrectnessOfHoldingXLocks", "CorrectnessOfWaitingForXLock", "CorrectReadView",
                                   "FirstCommitterWins", "CahillOK", "BernsteinOK", "(expected-to-be-violated predicate)"};
        return idx >= 0 && idx < 8 ? nm[idx] : "?";
    }
    return "?";
}

// `SYMMETRY Perms` names a definition of the model module, e.g. (examples/Paxos/MCPaxos.tla:14-style)
//   Perms == Permutations(TxnId) \cup Permutations(Key)
// Returns bit 0 for Permutations(TxnId), bit 1 for Permutations(Key), 0 if the definition is missing or names anything else.
static int symmetry_sets(const std::string &tla, const std::string &name) {
    size_t at = 0;
    for (;;) {
        at = tla.find(name, at);
        if (at == std::string::npos) return 0;
        const bool starts = at == 0 || tla[at - 1] == '\n';
        size_t q = at + name.size();
        while (q < tla.size() && (tla[q] == ' ' || tla[q] == '\t')) q++;
        if (starts && tla.compare(q, 2, "==") == 0) { at = q + 2; break; }
        at += name.size();
    }
    size_t end = tla.find("\n\n", at);   // a definition ends at the next blank line / the next definition / the module end
    for (const char *stop : {"==", "\n----", "\n===="}) {
        const size_t e = tla.find(stop, at);
        if (e != std::string::npos && (end == std::string::npos || e < end)) end = e;
    }
    std::string body = tla.substr(at, end == std::string::npos ? std::string::npos : end - at);
    if (body.find("==") == std::string::npos && end != std::string::npos && tla.compare(end, 2, "==") == 0) {
        const size_t nl = body.rfind('\n');   // the "==" belongs to the NEXT definition: drop its name line
        if (nl != std::string::npos) body.resize(nl);
    }
    int sets = 0;
    size_t k = 0;
    while ((k = body.find("Permutations", k)) != std::string::npos) {
        size_t a = body.find('(', k), b = body.find(')', k);
        if (a == std::string::npos || b == std::string::npos || b < a) return 0;
        std::string arg = body.substr(a + 1, b - a - 1);
        arg.erase(std::remove_if(arg.begin(), arg.end(), [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n'; }), arg.end());
        if (arg == "TxnId") sets |= 1;
        else if (arg == "Key") sets |= 2;
        else return 0;
        k = b;
    }
    return sets;
}


// ---------------------------------------------------------------------------------- the Paxos family (spec_paxos.h)
// examples/Paxos/MCVoting.tla + .cfg and MCPaxos.tla + .cfg: the cfg replaces Acceptor / Value / Quorum / Ballot by
// DEFINITIONS of the model module (`Acceptor <- MCAcceptor`, MCVoting.cfg:3-6), so the sizes of the model are read from the
// module text: `MCAcceptor == {a1, a2, a3}`, `MCQuorum == {{a1, a2}, ...}`, `MCBallot == 0..1` / `0..MCMaxBallot`.
namespace {
std::string strip_comments(const std::string &t) {
    std::string o;
    int depth = 0;
    for (size_t i = 0; i < t.size(); i++) {
        if (t.compare(i, 2, "(*") == 0) { depth++; i++; continue; }
        if (depth && t.compare(i, 2, "*)") == 0) { depth--; i++; continue; }
        if (depth) { if (t[i] == '\n') o += '\n'; continue; }
        if (t.compare(i, 2, "\\*") == 0) { while (i < t.size() && t[i] != '\n') i++; o += '\n'; continue; }
        o += t[i];
    }
    return o;
}
// body of the zero-argument definition `name == ...` (text up to the next line that starts in column 1), blanks removed
bool def_body(const std::string &t, const std::string &name, std::string &out) {
    size_t at = 0;
    for (;;) {
        at = t.find(name, at);
        if (at == std::string::npos) return false;
        size_t q = at + name.size();
        while (q < t.size() && (t[q] == ' ' || t[q] == '\t')) q++;
        if ((at == 0 || t[at - 1] == '\n') && t.compare(q, 2, "==") == 0) { at = q + 2; break; }
        at += name.size();
    }
    size_t end = at;
    while (end < t.size()) {
        const size_t eol = t.find('\n', end);
        if (eol == std::string::npos) { end = t.size(); break; }
        end = eol + 1;
        if (end < t.size() && t[end] != ' ' && t[end] != '\t' && t[end] != '\n' && t[end] != '\r') break;
    }
    out.clear();
    for (size_t i = at; i < end; i++)
        if (t[i] != ' ' && t[i] != '\t' && t[i] != '\n' && t[i] != '\r') out += t[i];
    return true;
}
// "{a1,a2,a3}" -> names
bool parse_id_set(const std::string &b, size_t &i, std::vector<std::string> &out) {
    if (i >= b.size() || b[i] != '{') return false;
    i++;
    out.clear();
    while (i < b.size() && b[i] != '}') {
        size_t j = i;
        while (j < b.size() && Lexer::idch(b[j])) j++;
        if (j == i) return false;
        out.push_back(b.substr(i, j - i));
        i = j;
        if (i < b.size() && b[i] == ',') i++;
    }
    if (i >= b.size()) return false;
    i++;
    return true;
}
}  // namespace

// $TLA_PATH: a ':'-separated list of directories searched (after the model's own) for the modules a model EXTENDS / INSTANCEs
static std::vector<std::string> tla_path_dirs() {
    std::vector<std::string> out;
    const char *env = getenv("TLA_PATH");
    if (!env) return out;
    const std::string e = env;
    size_t a = 0;
    while (a <= e.size()) {
        const size_t b = e.find(':', a);
        const std::string d = e.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (!d.empty()) out.push_back(d);
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}
static bool read_module(const char *model_path, const std::string &name, std::string &out) {
    if (read_file(dir_of(model_path) + "/" + name + ".tla", out)) return true;
    for (const auto &d : tla_path_dirs()) if (read_file(d + "/" + name + ".tla", out)) return true;
    return false;
}

// 1 = not a Paxos-family model (the other resolvers go on); MC_OK = R.d is filled in; < 0 = an error
static int resolve_paxos(const char *tla_path, const std::string &tla, const std::string &module, const mc_cfg *c, unsigned flags,
                         mc_spec_desc &d, std::string &def_text, std::string &def_module_name, std::string &warning) {
    const std::string t = strip_comments(tla);
    const size_t ex = t.find("EXTENDS");
    if (ex == std::string::npos) return 1;
    const size_t exend = t.find('\n', ex);
    const std::string exline = t.substr(ex, exend == std::string::npos ? std::string::npos : exend - ex);
    auto extends = [&](const char *m) {
        size_t k = exline.find(m);
        while (k != std::string::npos) {
            const bool l = k == 0 || !Lexer::idch(exline[k - 1]), r = k + strlen(m) >= exline.size() || !Lexer::idch(exline[k + strlen(m)]);
            if (l && r) return true;
            k = exline.find(m, k + 1);
        }
        return false;
    };
    const int kind = extends("Voting") ? 1 : extends("Paxos") ? 0 : -1;
    if (kind < 0) return 1;
    const char *base = kind ? "Voting" : "Paxos";
    auto target = [&](const char *name, std::string &out) {
        for (const auto &k : c->constants)
            if (k.name == name && k.replacement && k.module.empty()) { out = k.target; return true; }
        return false;
    };
    std::string tA, tV, tQ, tB, body;
    if (!target("Acceptor", tA) || !target("Value", tV) || !target("Quorum", tQ) || !target("Ballot", tB))
        return fe_fail(MC_EBADCFG, "%s: the cfg must replace Acceptor, Value, Quorum and Ballot by definitions of the model module "
                       "(Acceptor <- MCAcceptor ..., examples/Paxos/MCVoting.cfg:3-6)", module.c_str());
    if (kind == 0) {
        bool scoped = false;  // Voting's own Ballot == Nat must be replaced too (MCPaxos.cfg:9)
        for (const auto &k : c->constants) scoped |= k.name == "Ballot" && k.replacement && k.module == "Voting" && k.target == tB;
        if (!scoped) return fe_fail(MC_EBADCFG, "%s: Paxos instantiates Voting, whose Ballot == Nat needs `Ballot <-[Voting] %s` (MCPaxos.cfg:9)", module.c_str(), tB.c_str());
        if (!find_const(c, "None")) return fe_fail(MC_EBADCFG, "None is an unbounded CHOOSE (Paxos.tla:55): the cfg must give it a model value, None = None");
    }
    std::vector<std::string> acc, val, qs;
    size_t i = 0;
    if (!def_body(t, tA, body) || !parse_id_set(body, i = 0, acc) || i != body.size() || acc.empty() || acc.size() > 4)
        return fe_fail(MC_ENOSPEC, "%s: %s must be a set of 1..4 model values, {a1, ...}", module.c_str(), tA.c_str());
    if (!def_body(t, tV, body) || !parse_id_set(body, i = 0, val) || i != body.size() || val.empty() || val.size() > 3)
        return fe_fail(MC_ENOSPEC, "%s: %s must be a set of 1..3 model values, {v1, ...}", module.c_str(), tV.c_str());
    for (const auto &set : {acc, val})
        for (const auto &nm : set) {
            const CfgConst *k = find_const(c, nm.c_str());
            if (!k || k->replacement || k->value.kind != CfgValue::IDENT) return fe_fail(MC_EBADCFG, "%s must be a model value of the cfg (%s = %s)", nm.c_str(), nm.c_str(), nm.c_str());
        }
    std::vector<long long> masks;
    if (!def_body(t, tQ, body) || body.size() < 2 || body[0] != '{') return fe_fail(MC_ENOSPEC, "%s: %s must be a set of sets of acceptors", module.c_str(), tQ.c_str());
    for (i = 1; i < body.size() && body[i] != '}';) {
        if (!parse_id_set(body, i, qs) || qs.empty()) return fe_fail(MC_ENOSPEC, "%s: cannot read %s", module.c_str(), tQ.c_str());
        long long m = 0;
        for (const auto &nm : qs) {
            const auto it = std::find(acc.begin(), acc.end(), nm);
            if (it == acc.end()) return fe_fail(MC_ENOSPEC, "%s: %s is not an element of %s", tQ.c_str(), nm.c_str(), tA.c_str());
            m |= 1ll << (it - acc.begin());
        }
        masks.push_back(m);
        if (i < body.size() && body[i] == ',') i++;
    }
    if (masks.empty() || masks.size() > 8 || i + 1 != body.size()) return fe_fail(MC_ENOSPEC, "%s: %s must hold 1..8 quorums", module.c_str(), tQ.c_str());
    for (size_t a = 0; a < masks.size(); a++)  // QuorumAssumption (Voting.tla:16-17, Paxos.tla:12-13): TLC checks the ASSUME first
        for (size_t b = 0; b < masks.size(); b++)
            if (!(masks[a] & masks[b])) return fe_fail(MC_EBADCFG, "Assumption QuorumAssumption is false: two quorums of %s do not intersect", tQ.c_str());
    long long nb = -1;
    if (!def_body(t, tB, body) || body.compare(0, 3, "0..") != 0) return fe_fail(MC_ENOSPEC, "%s: %s must be 0..N", module.c_str(), tB.c_str());
    {
        std::string hi = body.substr(3), hb;
        if (!hi.empty() && !isdigit((unsigned char)hi[0]) && def_body(t, hi, hb)) hi = hb;
        char *endp = nullptr;
        nb = strtoll(hi.c_str(), &endp, 10) + 1;
        if (hi.empty() || *endp || nb < 1 || nb > 4) return fe_fail(MC_ENOSPEC, "%s: %s must be 0..N with N <= 3", module.c_str(), tB.c_str());
    }
    if (c->specification != "Spec" || !c->init.empty() || !c->next.empty())
        return fe_fail(MC_ENOSPEC, "%s: only SPECIFICATION Spec (%s.tla's own Init and Next) is lowered", module.c_str(), base);
    if (!c->constraints.empty() || !c->action_constraints.empty() || !c->view.empty()) return fe_fail(MC_ENOSPEC, "%s: CONSTRAINT / VIEW are not part of the lowering", module.c_str());
    long long inv = 0;
    for (const auto &nm : c->invariants) {
        std::string b;
        if (kind == 1) {
            if (nm == "Inv") inv |= 1;
            else if (nm == "TypeOK") {}  // holds by construction of the packed state
            else return fe_fail(MC_ENOSPEC, "the Voting lowering checks INVARIANT Inv (Voting.tla:160); '%s' is not lowered", nm.c_str());
        } else if (nm == "Inv") inv |= 15;
        else if (nm == "TypeOK") inv |= 1;
        else if (def_body(t, nm, b) && b.size() == 5 && b.compare(0, 4, "Inv!") == 0 && b[4] >= '1' && b[4] <= '4') inv |= 1ll << (b[4] - '1');  // Inv3 == Inv!3 (MCPaxos.tla:65-68)
        else return fe_fail(MC_ENOSPEC, "the Paxos lowering checks Inv (Paxos.tla:192-208) and its conjuncts Inv!1..Inv!4; '%s' is not one of them", nm.c_str());
    }
    long long prop = 0;
    for (const auto &nm : c->properties) {
        std::string b;
        if (def_body(t, nm, b) && b == (kind ? "C!Spec" : "V!Spec")) prop = 1;  // ConsensusSpecBar == C!Spec (MCVoting.tla:26)
        else return fe_fail(MC_ENOSPEC, "PROPERTY %s: only the refinement %s (its safety part, checked on every transition) is lowered", nm.c_str(), kind ? "C!Spec" : "V!Spec");
    }
    long long sym = 0;
    if (!c->symmetry.empty()) {
        std::string b;
        if (!def_body(t, c->symmetry, b)) return fe_fail(MC_ENOSPEC, "SYMMETRY %s: no such definition in %s", c->symmetry.c_str(), module.c_str());
        size_t k = 0;
        while (k < b.size()) {
            if (b.compare(k, 13, "Permutations(") != 0) return fe_fail(MC_ENOSPEC, "SYMMETRY %s must be a union of Permutations(%s) and Permutations(%s)", c->symmetry.c_str(), tA.c_str(), tV.c_str());
            const size_t e = b.find(')', k);
            if (e == std::string::npos) return fe_fail(MC_ENOSPEC, "SYMMETRY %s: cannot read the definition", c->symmetry.c_str());
            const std::string arg = b.substr(k + 13, e - k - 13);
            if (arg == tA) sym |= 1; else if (arg == tV) sym |= 2;
            else return fe_fail(MC_ENOSPEC, "SYMMETRY %s: Permutations(%s) is neither the acceptors nor the values", c->symmetry.c_str(), arg.c_str());
            k = e + 1;
            if (b.compare(k, 4, "\\cup") == 0) k += 4; else if (b.compare(k, 6, "\\union") == 0) k += 6;
        }
    }
    // the lowering is written against Voting.tla, Paxos.tla and Consensus.tla: verified where they are found (beside the
    // model or under $TLA_PATH), refused otherwise unless -unverified
    const std::vector<std::pair<const char *, uint64_t>> need = kind ? std::vector<std::pair<const char *, uint64_t>>{{"Voting", H_VOTING}, {"Consensus", H_CONSENSUS}}
                                                                      : std::vector<std::pair<const char *, uint64_t>>{{"Paxos", H_PAXOS}, {"Voting", H_VOTING}};
    for (const auto &nh : need) {
        std::string text, part;
        const bool found = read_module(tla_path, nh.first, text);
        if (found) {
            if (!module_body(text, part) || text_hash(part) != nh.second)
                return fe_fail(MC_ENOSPEC, "%s.tla differs from the text the lowering was written against (examples/Paxos/%s.tla)", nh.first, nh.first);
            if (!strcmp(nh.first, base)) { def_text = text; def_module_name = base; }
        } else if (flags & MC_F_UNVERIFIED) {
            warning = std::string("Warning: ") + nh.first + ".tla was found neither beside the module nor under $TLA_PATH; the built-in lowering was used without checking the module text.\n";
        } else {
            return fe_fail(MC_ENOSPEC, "module %s (needed by %s) was found neither beside it nor under $TLA_PATH: the lowering cannot be checked against its text "
                           "(put %s.tla there, or pass -unverified)", nh.first, module.c_str(), nh.first);
        }
    }
    memset(&d, 0, sizeof d);
    d.spec_id = MC_SPEC_PAXOS;
    d.params[0] = kind; d.params[1] = (long long)acc.size(); d.params[2] = (long long)val.size(); d.params[3] = nb;
    d.params[4] = inv; d.params[5] = sym; d.params[6] = prop; d.params[7] = (long long)masks.size();
    for (size_t q = 0; q < masks.size(); q++) d.params[8 + q] = masks[q];
    d.nparams = 8 + (uint32_t)masks.size();
    return MC_OK;
}

int mc_check_files(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report, size_t report_cap,
                   mc_result *res) {
    return mc_check_files_dump(tla_path, cfg_path, cfg, report, report_cap, res, nullptr);
}

int mc_check_files_dump(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report, size_t report_cap,
                        mc_result *res, const char *dump_path) {
    return mc_check_files_ckpt(tla_path, cfg_path, cfg, report, report_cap, res, dump_path, nullptr, nullptr);
}

// The front half of `tlc X.tla`: X.tla + X.cfg -> the lowering's descriptor (verified against the module text), or a compiled
// PlusCal program.  Shared by mc_check_files (one GPU) and mc_resolve_files (one engine per rank in sharded mode).
namespace {
struct Resolved {
    mc_spec_desc d;
    mc_program *prog = nullptr;
    std::string tla, module, def_text, def_module_name;  // def_*: text + name of the module that holds the action definitions
    std::string warning;  // printed at the top of the report (MC_F_UNVERIFIED)
    int check_deadlock = -1;   // the cfg's CHECK_DEADLOCK statement (-1: none)
    ~Resolved() { if (prog) mc_program_free(prog); }
};
}  // namespace
static int resolve_files(const char *tla_path, const char *cfg_path, unsigned flags, Resolved &R) {
    {   // MC_F_UNVERIFIED can also be given through the environment (callers that only have the `mc` command line of a script)
        const char *uv = getenv("TLAMC_UNVERIFIED");
        if (uv && uv[0] == '1') flags |= MC_F_UNVERIFIED;
    }
    std::string &tla = R.tla, &module = R.module, &def_text = R.def_text, &def_module_name = R.def_module_name;
    mc_spec_desc &d = R.d;
    mc_program *&prog = R.prog;
    std::string cfgtext;
    if (!read_file(tla_path, tla)) return fe_fail(MC_EPARSE, "cannot read %s", tla_path);
    if (!module_name(tla, module)) return fe_fail(MC_EPARSE, "%s: no MODULE header", tla_path);
    std::string cpath;
    if (cfg_path) cpath = cfg_path;
    else {  // X.cfg beside X.tla (README.md:356)
        cpath = tla_path;
        const size_t dot = cpath.rfind(".tla");
        if (dot != std::string::npos) cpath.replace(dot, 4, ".cfg"); else cpath += ".cfg";
    }
    if (!read_file(cpath, cfgtext)) {
        // a PlusCal module without a cfg is checked with no constants and no invariants (assert / deadlock only)
        if (cfg_path || tla.find("--algorithm") == std::string::npos) return fe_fail(MC_EPARSE, "cannot read configuration file %s", cpath.c_str());
        cfgtext.clear();
    }
    mc_cfg *c = nullptr;
    int rc = mc_cfg_parse(cfgtext.c_str(), cfgtext.size(), &c);
    if (rc) return rc;
    R.check_deadlock = c->check_deadlock;
    memset(&d, 0, sizeof d);
    // A PlusCal module goes through the hand lowering when its algorithm text is one the registry knows, and
    // through the compiled program (spec_vm.h) otherwise — or always with MC_F_GENERIC (A/B of the two paths).
    std::string part;
    const bool has_alg = algorithm_text(tla, part);
    const uint64_t alg_hash = has_alg ? text_hash(part) : 0;
    bool generic = has_alg && (flags & MC_F_GENERIC);
    if (!has_alg) {  // the Paxos family is recognised by what the model EXTENDS and by the cfg's replacements
        rc = resolve_paxos(tla_path, tla, module, c, flags, d, def_text, def_module_name, R.warning);
        if (rc <= 0) { mc_cfg_free(c); return rc; }
    }
    if (!generic) {
        rc = mc_spec_resolve(module.c_str(), c, &d);
        if (rc == MC_ENOSPEC && has_alg) generic = true;
        else if (rc) { mc_cfg_free(c); return rc; }
        if (!generic && d.spec_id == MC_SPEC_PCAL_INTRO && alg_hash != H_PCAL_INTRO && alg_hash != H_PCAL_INTRO_README) generic = true;
        if (!generic && d.spec_id == MC_SPEC_ATOMIC_ADD && module == "atomic_add" && alg_hash != H_ATOMIC_ADD) generic = true;
        // an edited atomic_add_n.tla (another await, an extra label) is not the N-adder hand lowering: compile it instead
        if (!generic && d.spec_id == MC_SPEC_ATOMIC_ADD && module == "atomic_add_n" && alg_hash != H_ATOMIC_ADD_N) generic = true;
    }
    const std::string symmetry_name = c->symmetry;
    mc_cfg_free(c);
    if (generic) {
        if ((rc = mc_program_compile(tla.c_str(), cfgtext.c_str(), &prog))) return rc;
        mc_program_spec(prog, &d);
        def_text = mc_program_translated(prog);  // action spans refer to the translation pcal2tla would insert
        def_module_name = module;
    } else if (d.spec_id == MC_SPEC_PCAL_INTRO) {
        d.params[0] = alg_hash == H_PCAL_INTRO ? 0 : 1;
    } else if (d.spec_id == MC_SPEC_ATOMIC_ADD && module == "atomic_add") {
    } else if (d.spec_id == MC_SPEC_RAFT) {
        // The lowering is written against two texts: examples/raft.tla and the wrapper specs/MCraft.tla (StateConstraint,
        // the invariant definitions).  Both are verified; a module that cannot be found is refused (never mis-checked)
        // unless the caller asks for the built-in lowering unverified (MC_F_UNVERIFIED, `mc -unverified`).
        std::string raft;
        bool found = read_module(tla_path, "raft", raft);
        if (module == "raft") { raft = tla; found = true; }
        else {
            if (!module_body(tla, part)) return fe_fail(MC_ENOSPEC, "%s: cannot find the module body", tla_path);
            const uint64_t hw = text_hash(part);
            if (hw != H_MCRAFT)
                return fe_fail(MC_ENOSPEC, "module %s differs from the wrapper the raft lowering was written against (specs/MCraft.tla; hash %016llx): "
                               "its StateConstraint / invariant definitions are hard-wired", module.c_str(), (unsigned long long)hw);
        }
        if (found) {
            if (!module_body(raft, part)) return fe_fail(MC_ENOSPEC, "raft.tla: cannot find the module body");
            const uint64_t h = text_hash(part);
            if (h != H_RAFT) return fe_fail(MC_ENOSPEC, "raft.tla differs from the text the lowering was written against (hash %016llx)", (unsigned long long)h);
            if (module != "raft") { def_text = raft; def_module_name = "raft"; }
        } else if (flags & MC_F_UNVERIFIED) {
            R.warning = "Warning: raft.tla was found neither beside the module nor under $TLA_PATH; the built-in lowering of examples/raft.tla was used without checking the module text.\n";
        } else {
            return fe_fail(MC_ENOSPEC, "module raft (EXTENDed by %s) was found neither beside it nor under $TLA_PATH: the lowering cannot be checked against its text "
                           "(put raft.tla there, or pass -unverified to use the built-in lowering of examples/raft.tla as is)", module.c_str());
        }
    }
    if (d.spec_id == MC_SPEC_SSI && d.params[5]) {
        const int sets = symmetry_sets(tla, symmetry_name);
        if (sets <= 0) return fe_fail(MC_ENOSPEC, "SYMMETRY %s: the module must define it from Permutations(TxnId) and / or Permutations(Key)", symmetry_name.c_str());
        d.params[5] = sets;
    }
    if (d.spec_id == MC_SPEC_SSI) {
        std::string ssi;  // the MC wrapper EXTENDS the spec: verify it when it can be found
        const bool tb = d.params[4] != 0;
        const std::string base = tb ? "textbookSnapshotIsolation" : "serializableSnapshotIsolation";
        bool found = read_module(tla_path, base, ssi);
        if (module == base) { ssi = tla; found = true; }
        else {
            if (!module_body(tla, part)) return fe_fail(MC_ENOSPEC, "%s: cannot find the module body", tla_path);
            const uint64_t hw = text_hash(part);
            if (hw != (tb ? H_MCTEXTBOOK_SI : H_MCSSI))
                return fe_fail(MC_ENOSPEC, "module %s differs from the wrapper the lowering was written against (specs/%s.tla; hash %016llx): "
                               "its invariant and Perms definitions are hard-wired", module.c_str(), tb ? "MCtextbookSI" : "MCssi", (unsigned long long)hw);
        }
        if (found) {
            if (!module_body(ssi, part)) return fe_fail(MC_ENOSPEC, "%s.tla: cannot find the module body", base.c_str());
            const uint64_t h = text_hash(part);
            if (h != (tb ? H_TEXTBOOK_SI : H_SSI))
                return fe_fail(MC_ENOSPEC, "%s.tla differs from the text the lowering was written against (hash %016llx)", base.c_str(), (unsigned long long)h);
            if (module != base) { def_text = ssi; def_module_name = base; }
        } else if (flags & MC_F_UNVERIFIED) {
            R.warning = "Warning: " + base + ".tla was found neither beside the module nor under $TLA_PATH; the built-in lowering was used without checking the module text.\n";
        } else {
            return fe_fail(MC_ENOSPEC, "module %s (EXTENDed by %s) was found neither beside it nor under $TLA_PATH: the lowering cannot be checked against its text "
                           "(put %s.tla there, or pass -unverified to use the built-in lowering as is)", base.c_str(), module.c_str(), base.c_str());
        }
    }
    return MC_OK;
}

int mc_resolve_files(const char *tla_path, const char *cfg_path, unsigned flags, mc_spec_desc *out, mc_program **prog_out) {
    if (!tla_path || !out || !prog_out) return MC_EBADCFG;
    *prog_out = nullptr;
    Resolved R;
    const int rc = resolve_files(tla_path, cfg_path, flags, R);
    if (rc) return rc;
    *out = R.d;
    *prog_out = R.prog;  // the descriptor of a compiled program points into it: the caller frees it after its engines
    R.prog = nullptr;
    return MC_OK;
}

// Is the module one of the families that HAVE a GPU lowering (the model-checking hot path)?  Those never reach the host
// evaluator: a raft / snapshot-isolation / Paxos / PlusCal model that the lowerings refuse stays refused (MC_ENOSPEC).
static bool lowered_family(const std::string &module, const std::string &tla) {
    static const char *const names[] = {"atomic_add", "atomic_add_n", "pcal_intro", "MCraft", "raft", "MCssi", "serializableSnapshotIsolation",
                                        "MCtextbookSI", "textbookSnapshotIsolation", "MCPaxos", "MCVoting", "Paxos", "Voting"};
    for (const char *n : names) if (module == n) return true;
    if (tla.find("--algorithm") != std::string::npos || tla.find("--fair") != std::string::npos) return true;
    const std::string t = strip_comments(tla);
    const size_t ex = t.find("EXTENDS");
    if (ex == std::string::npos) return false;
    const size_t exend = t.find('\n', ex);
    const std::string line = t.substr(ex, exend == std::string::npos ? std::string::npos : exend - ex);
    for (const char *n : {"raft", "serializableSnapshotIsolation", "textbookSnapshotIsolation", "Voting", "Paxos"}) {
        size_t k = line.find(n);
        while (k != std::string::npos) {
            const bool l = k == 0 || !Lexer::idch(line[k - 1]), r = k + strlen(n) >= line.size() || !Lexer::idch(line[k + strlen(n)]);
            if (l && r) return true;
            k = line.find(n, k + 1);
        }
    }
    return false;
}

// does X.cfg (read beside X.tla when not given) name neither SPECIFICATION nor INIT / NEXT?  (false when it cannot be read / parsed:
// the ordinary path then reports that)
static bool no_behavior_cfg(const char *tla_path, const char *cfg_path) {
    std::string cpath, text;
    if (cfg_path) cpath = cfg_path;
    else {
        cpath = tla_path;
        const size_t dot = cpath.rfind(".tla");
        if (dot != std::string::npos) cpath.replace(dot, 4, ".cfg"); else cpath += ".cfg";
    }
    if (!read_file(cpath, text)) return false;
    mc_cfg *c = nullptr;
    if (mc_cfg_parse(text.c_str(), text.size(), &c)) return false;
    const bool none = c->specification.empty() && c->init.empty() && c->next.empty();
    mc_cfg_free(c);
    return none;
}

// `tlc X.tla` for a TLA+ module that has NO GPU lowering (the Specifying Systems examples of the reference: MCInnerSerial.tla and
// its TLC log testout2): the general evaluator of tlaeval.h runs TLC's breadth-first search on the host and the report says so.
static int host_evaluate(const char *tla_path, const char *cfg_path, const mc_config *cfg, const std::string &module, char *report,
                         size_t report_cap, mc_result *res) {
    std::string cpath;
    if (cfg_path) cpath = cfg_path;
    else {
        cpath = tla_path;
        const size_t dot = cpath.rfind(".tla");
        if (dot != std::string::npos) cpath.replace(dot, 4, ".cfg"); else cpath += ".cfg";
    }
    tlaeval::Options opt;
    opt.max_levels = cfg->max_levels;
    opt.max_distinct = cfg->max_distinct;
    opt.check_deadlock = (cfg->flags & MC_F_DEADLOCK) != 0;
    if (cfg->flags & MC_F_PROGRESS) { const char *iv = getenv("TLAMC_PROGRESS_INTERVAL"); opt.progress_seconds = iv ? atof(iv) : 60.0; }
    opt.search = tla_path_dirs();
    tlaeval::Result r;
    std::string err;
    const int rc = tlaeval::check_files(tla_path, cpath, opt, r, err);
    if (rc) return fe_fail(rc, "%s: %s", module.c_str(), err.c_str());
    memset(res, 0, sizeof *res);
    res->distinct = r.distinct; res->generated = r.generated; res->queue_left = r.queue_left; res->depth = r.depth; res->verdict = r.verdict;
    res->violated_invariant = r.violated_invariant; res->trace_len = (uint32_t)r.trace.size(); res->seconds = r.seconds; res->host_evaluated = 1;
    res->levels = (uint32_t)std::min<size_t>(r.levels.size(), MC_MAX_LEVELS);
    for (uint32_t i = 0; i < res->levels; i++) res->level_distinct[i] = r.levels[i];
    res->unchecked_properties = (uint32_t)r.unchecked_properties.size();
    Out o{report, report_cap, 0};
    if (r.no_behavior) {
        // TLC's "No Behavior Spec" mode: what Print / PrintT printed while the ASSUMEs were evaluated, then TLC's closing lines
        o.put("Module %s, no behavior spec: the assumptions are evaluated on the host by the general TLA+ evaluator.\n", module.c_str());
        for (auto &ln : r.printed) o.put("%s\n", ln.c_str());
        if (r.verdict == MC_V_OK) o.put("Model checking completed. No error has been found.\n");
        else if (r.verdict == MC_V_ASSERT) o.put("The first argument of Assert evaluated to FALSE; the second argument was:\n\"%s\"\n", r.error_message.c_str());
        else o.put("Error: %s\n", r.error_message.c_str());
        o.put("0 states generated, 0 distinct states found, 0 states left on queue.\n");
        return MC_OK;
    }
    o.put("Module %s has no GPU lowering: evaluated on the host by the general TLA+ evaluator.\n", module.c_str());
    o.put("Finished computing initial states: %llu distinct state%s generated.\n", (unsigned long long)r.init_states, r.init_states == 1 ? "" : "s");
    if (!r.unchecked_properties.empty()) {
        // Liveness is out of scope — saying nothing about it is not (Liveness/LiveHourClock.cfg:10 PROPERTIES AlwaysTick AllTimes
        // TypeInvariance: TLC checks all three; here only []HCini is).  The line comes BEFORE the verdict it qualifies.
        std::string names;
        for (auto &n : r.unchecked_properties) names += (names.empty() ? "" : ", ") + n;
        o.put("Warning: temporal propert%s %s NOT checked: liveness (<>, ~>, WF_ / SF_ fairness) is not supported; only the safety parts of the "
              "PROPERTIES ([]P, [][A]_v, initial predicates) are checked.\n", r.unchecked_properties.size() == 1 ? "y" : "ies", names.c_str());
    }
    if (r.verdict == MC_V_OK || r.verdict == MC_V_BUDGET) {
        if (r.verdict == MC_V_OK) o.put("Model checking completed. No error has been found.\n");
        else o.put("Search stopped by the level/state budget; no error has been found so far.\n");
    } else {
        if (r.verdict == MC_V_ASSERT) o.put("The first argument of Assert evaluated to FALSE; the second argument was:\n\"%s\"\n", r.error_message.c_str());
        else if (r.verdict == MC_V_INVARIANT && r.violated_invariant >= 0 && (size_t)r.violated_invariant >= r.n_invariants) o.put("Error: Action property %s is violated.\n", r.violated_name.c_str());
        else if (r.verdict == MC_V_INVARIANT) o.put("Error: Invariant %s is violated.\n", r.violated_name.c_str());
        else if (r.verdict == MC_V_DEADLOCK) o.put("Error: Deadlock reached.\n");
        else o.put("Error: evaluation error: %s\n", r.error_message.c_str());
        if (!r.trace.empty()) {
            o.put("Error: The behavior up to this point is:\n");
            for (size_t k = 0; k < r.trace.size(); k++) {  // README.md:270-311
                if (r.trace[k].first.empty()) o.put("State %zu:\n%s\n\n", k + 1, r.trace[k].second.c_str());
                else o.put("State %zu: <%s>\n%s\n\n", k + 1, r.trace[k].first.c_str(), r.trace[k].second.c_str());
            }
        }
    }
    o.put("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)r.generated, (unsigned long long)r.distinct,
          (unsigned long long)r.queue_left);
    if (r.verdict == MC_V_OK) o.put("The state graph has diameter %u.\n", r.depth);  // testout2:266
    else o.put("The depth of the complete state graph search is %u.\n", r.depth);
    return MC_OK;
}

extern "C" void *mc_jit_factory(const void *program);   // pcal_codegen.cpp
static bool jit_compiler_present() {
    const char *hc = getenv("HIPCC");
    struct stat st;   // (<unistd.h>'s access() is not to be had here: its R_OK macro collides with the interpreter's status names)
    return stat(hc && *hc ? hc : "/opt/rocm/bin/hipcc", &st) == 0 && (st.st_mode & S_IXUSR);
}

int mc_check_files_ckpt(const char *tla_path, const char *cfg_path, const mc_config *cfg, char *report, size_t report_cap,
                        mc_result *res, const char *dump_path, const char *recover_path, const char *checkpoint_path) {
    if (!tla_path || !cfg || !report || !report_cap || !res) return MC_EBADCFG;
    report[0] = 0;
    // A cfg that names no behaviour (neither SPECIFICATION nor INIT / NEXT) asks for the module's ASSUMEs to be evaluated — TLC's "No
    // Behavior Spec" mode — whatever the module EXTENDS: the in-spec unit tests of the snapshot-isolation models are run that way
    // (serializableSnapshotIsolation.tla:1062-1066).  No state is generated, so no lowering is stood in for.
    if (no_behavior_cfg(tla_path, cfg_path)) {
        std::string tla, module;
        if (!read_file(tla_path, tla)) return fe_fail(MC_EPARSE, "cannot read %s", tla_path);
        if (!module_name(tla, module)) return fe_fail(MC_EPARSE, "%s: no MODULE header", tla_path);
        if (tla.find("--algorithm") == std::string::npos && tla.find("--fair") == std::string::npos) {
            if (dump_path || recover_path || checkpoint_path) return fe_fail(MC_ENOSPEC, "%s: -dump / -recover / -checkpoint need a behaviour spec", module.c_str());
            return host_evaluate(tla_path, cfg_path, cfg, module, report, report_cap, res);
        }
    }
    Resolved R;
    int rc = resolve_files(tla_path, cfg_path, cfg->flags, R);
    if (rc == MC_ENOSPEC && !R.module.empty() && !lowered_family(R.module, R.tla)) {
        if (dump_path || recover_path || checkpoint_path) return fe_fail(MC_ENOSPEC, "%s: -dump / -recover / -checkpoint need a GPU lowering (the module is evaluated on the host)", R.module.c_str());
        return host_evaluate(tla_path, cfg_path, cfg, R.module, report, report_cap, res);
    }
    if (rc) return rc;
    const mc_spec_desc &d = R.d;
    const std::string &tla = R.tla, &module = R.module, &def_text = R.def_text, &def_module_name = R.def_module_name;
    mc_program *const prog = R.prog;
    const bool generic = prog != nullptr;  // the module went through the PlusCal compiler
    mc_engine *e = nullptr;
    mc_config with_cfg_statements;
    if (R.check_deadlock >= 0) {   // the cfg's CHECK_DEADLOCK statement decides over the default (the command line's -deadlock still turns it off)
        with_cfg_statements = *cfg;
        if (!R.check_deadlock) with_cfg_statements.flags &= ~MC_F_DEADLOCK;
        else with_cfg_statements.flags |= MC_F_DEADLOCK;   // CHECK_DEADLOCK TRUE asks for the check whatever the caller's flags said
        cfg = &with_cfg_statements;
    }
    if ((rc = mc_engine_create(&d, cfg, &e))) return rc;
    if (recover_path && (rc = mc_engine_restore(e, recover_path))) {  // TLC -recover
        // a checkpoint written by the program's GENERATED code (a run that moved to it by itself, or -jit) holds rows packed to the cells'
        // inferred ranges: the interpreter's engine refuses it (another program identity) and the generated code continues it
        mc_engine *ej = nullptr;
        mc_config with_jit = *cfg;
        with_jit.flags |= MC_F_JIT;
        const std::string first_error = mc_last_error();
        if (rc == MC_EBADCFG && generic && d.spec_id == MC_SPEC_PCAL && !(cfg->flags & MC_F_JIT) && jit_compiler_present()) {
            mc_engine_destroy(e);
            e = nullptr;
            if (mc_engine_create(&d, &with_jit, &ej) == MC_OK && mc_engine_restore(ej, recover_path) == MC_OK) { e = ej; rc = MC_OK; }
            else { if (ej) mc_engine_destroy(ej); mc_set_error_internal(first_error.c_str()); }
        }
        if (rc) { if (e) mc_engine_destroy(e); return rc; }
    }
    const auto print_progress = [](void *, uint32_t levels, uint64_t g, uint64_t dst, uint64_t q) {
        printf("Progress(%u): %llu states generated, %llu distinct states found, %llu states left on queue.\n", levels,
               (unsigned long long)g, (unsigned long long)dst, (unsigned long long)q);
        fflush(stdout);
    };
    const char *const piv = getenv("TLAMC_PROGRESS_INTERVAL");
    const double progress_every = piv ? atof(piv) : 1.0;
    // testout2:4-259: one line per report, straight to stdout while the search runs
    if (cfg->flags & MC_F_PROGRESS) mc_engine_set_progress(e, print_progress, nullptr, progress_every);
    // FROM THE INTERPRETER TO GENERATED CODE, on its own (round 6).  A compiled PlusCal program starts on the device interpreter — no
    // compiler run, right for the models that finish in a blink — and once the search has lasted $TLAMC_AUTOJIT_AFTER seconds (default
    // 0.3) the program's generated code is built beside it (mc_jit_factory on a thread of its own: hipcc, 5 - 9 s, cached).  When
    // that library is there before the interpreter is done, the run stops at its next level (mc_engine_request_stop) and STARTS OVER
    // as generated code: 50 - 100 x the interpreter's rate, so what is searched again costs a percent or two of what
    // the interpreter had spent.  Same packed states, fingerprints, report.  Off: $TLAMC_AUTOJIT=0, -jit (generated code from the
    // first state), -recover (the checkpointed engine continues), no compiler on the machine.
    struct AutoJit {
        mc_engine *e = nullptr;
        std::chrono::steady_clock::time_point t0, last_print;
        double after = 1.0, every = 1.0;
        bool print = false, started = false, stopped = false;
        const void *program = nullptr;
        std::shared_ptr<std::atomic<int>> built;   // 0 = building, 1 = there, -1 = not to be had
    } aj;
    const char *const aje = getenv("TLAMC_AUTOJIT"), *const tje = getenv("TLAMC_JIT");
    const bool auto_jit = generic && d.spec_id == MC_SPEC_PCAL && d.nparams >= 1 && d.params[0] && !(cfg->flags & MC_F_JIT) && !(tje && *tje && *tje != '0') &&
                          !recover_path && !(aje && *aje == '0') && jit_compiler_present();
    if (auto_jit) {
        aj.e = e;
        aj.t0 = aj.last_print = std::chrono::steady_clock::now();
        const char *aa = getenv("TLAMC_AUTOJIT_AFTER");
        aj.after = aa ? atof(aa) : 0.3;
        aj.every = progress_every;
        aj.print = (cfg->flags & MC_F_PROGRESS) != 0;
        aj.program = (const void *)(intptr_t)d.params[0];
        aj.built = std::make_shared<std::atomic<int>>(0);
        mc_engine_set_progress(e, [](void *u, uint32_t levels, uint64_t g, uint64_t dst, uint64_t q) {
            AutoJit &a = *(AutoJit *)u;
            const auto now = std::chrono::steady_clock::now();
            if (a.print && std::chrono::duration<double>(now - a.last_print).count() >= a.every) {
                a.last_print = now;
                printf("Progress(%u): %llu states generated, %llu distinct states found, %llu states left on queue.\n", levels,
                       (unsigned long long)g, (unsigned long long)dst, (unsigned long long)q);
                fflush(stdout);
            }
            if (!a.started && std::chrono::duration<double>(now - a.t0).count() >= a.after) {
                a.started = true;
                std::thread([built = a.built, program = a.program] { built->store(mc_jit_factory(program) ? 1 : -1); }).detach();
            }
            if (a.started && !a.stopped && a.built->load() == 1) {
                a.stopped = true;
                mc_engine_request_stop(a.e);
            }
        }, &aj, 0.02);
    }
    rc = mc_engine_run(e, res);
    if (rc) { mc_engine_destroy(e); return rc; }
    if (auto_jit && aj.stopped && res->verdict == MC_V_BUDGET && !(cfg->max_levels && res->levels >= cfg->max_levels) &&
        !(cfg->max_distinct && res->distinct >= cfg->max_distinct)) {
        const uint64_t had = res->distinct;
        const double t_switch = std::chrono::duration<double>(std::chrono::steady_clock::now() - aj.t0).count();
        mc_config with_jit = *cfg;
        with_jit.flags |= MC_F_JIT;
        // (the second engine beside the first when the device has room for both: giving 180 GB back to the driver and asking for them again
        //  took 9 of a 20-second run, profiles/r06zq — the interpreter's engine is then destroyed after the search, where every run pays that)
        mc_engine *old_e = e, *e2 = nullptr;
        rc = mc_engine_create(&d, &with_jit, &e2);
        if (rc) {
            mc_engine_destroy(old_e);
            old_e = nullptr;
            if ((rc = mc_engine_create(&d, &with_jit, &e2))) return rc;
        }
        e = e2;
        const double t_ready = std::chrono::duration<double>(std::chrono::steady_clock::now() - aj.t0).count();
        if (cfg->flags & MC_F_PROGRESS) mc_engine_set_progress(e, print_progress, nullptr, progress_every);
        rc = mc_engine_run(e, res);
        fprintf(stderr, "mc: %llu distinct states after %.1f s on the device interpreter when the program's generated code was built: started over with it "
                        "(engine %s after %.1f s, search %.2f s)\n", (unsigned long long)had, t_switch, old_e ? "beside the first" : "in the first one's place",
                t_ready - t_switch, std::chrono::duration<double>(std::chrono::steady_clock::now() - aj.t0).count() - t_ready);
        if (old_e) mc_engine_destroy(old_e);
        if (rc) { mc_engine_destroy(e); return rc; }
    }
    const bool clean = res->verdict == MC_V_OK || res->verdict == MC_V_BUDGET;
    if (checkpoint_path && clean && (rc = mc_engine_checkpoint(e, checkpoint_path))) { mc_engine_destroy(e); return rc; }

    Out o{report, report_cap, 0};
    if (!R.warning.empty()) o.put("%s", R.warning.c_str());
    o.put("Finished computing initial states: %llu distinct state%s generated.\n", (unsigned long long)res->level_distinct[0],
          res->level_distinct[0] == 1 ? "" : "s");
    if (res->verdict == MC_V_OK || res->verdict == MC_V_BUDGET) {
        if (res->verdict == MC_V_OK) o.put("Model checking completed. No error has been found.\n");
        else o.put("Search stopped by the level/state budget; no error has been found so far.\n");
        if (checkpoint_path) o.put("-- Checkpointing of run %s completed.\n", checkpoint_path);  // testout1:10
        // testout2:261-264: optimistic estimate = (generated - distinct) * distinct / 2^64
        const double opt = (double)(res->generated - res->distinct) * (double)res->distinct / 18446744073709551616.0;
        o.put("  Estimates of the probability that TLC did not check all reachable states\n"
              "  because two distinct states had the same fingerprint:\n  calculated (optimistic):  val = %.2g\n", opt);
    } else {
        if (res->verdict == MC_V_ASSERT && !generic) {
            // the message pcal2tla puts into the translation names the position of the `assert` statement in THIS module's
            // text (pcal_intro.tla:16 col 4 in the reference's layout; the hash that selected the lowering ignores layout)
            int al = 16, ac = 4;
            const size_t a0 = tla.find("--algorithm");
            size_t at = a0 == std::string::npos ? a0 : tla.find("assert", a0);
            if (at != std::string::npos) {
                al = 1; ac = 1;
                for (size_t k = 0; k < at; k++) { if (tla[k] == '\n') { al++; ac = 1; } else ac++; }
            }
            o.put("The first argument of Assert evaluated to FALSE; the second argument was:\n\"Failure of assertion at line %d, column %d.\"\n", al, ac);
        }
        else if (res->verdict == MC_V_ASSERT) { /* compiled program: the message names the failing assert, found below */ }
        else if (res->verdict == MC_V_INVARIANT && d.spec_id == MC_SPEC_PAXOS && res->violated_invariant == (d.params[0] == 1 ? 1 : 4))
            o.put("Error: Action property %s is violated.\n", invariant_name(&d, res->violated_invariant));  // the step breaks [Next]_v of the PROPERTY
        else if (res->verdict == MC_V_INVARIANT) o.put("Error: Invariant %s is violated.\n", invariant_name(&d, res->violated_invariant));
        else if (res->verdict == MC_V_DEADLOCK) o.put("Error: Deadlock reached.\n");
        else o.put("Error: evaluation error (a function was applied outside its domain).\n");
        const size_t W = mc_state_bytes(&d);
        std::string assert_def = "C";  // the action whose Assert failed
        size_t n = res->trace_len ? res->trace_len : 1;
        std::vector<uint8_t> states(n * W);
        std::vector<int32_t> acts(n);
        // action definitions live in the checked module, or in the module an MC wrapper EXTENDS
        const std::vector<std::string> srcL = split_lines(def_text.empty() ? tla : def_text);
        const std::string &def_module = def_text.empty() ? module : def_module_name;
        const bool have_trace = mc_engine_trace(e, states.data(), acts.data(), &n) == MC_OK && n;
        if (generic && res->verdict == MC_V_ASSERT) {  // which assert: re-evaluate the last state of the trace on the host
            std::string msg = "Failure of assertion.";
            if (have_trace) {
                const pcal::Program &P = prog->prog;
                std::vector<int32_t> vals((size_t)P.nv + 1);
                const uint64_t *w = (const uint64_t *)&states[(n - 1) * W];
                for (int i = 0; i < P.nv; i++) vals[(size_t)i] = (int32_t)(uint32_t)(w[i / 2] >> (32 * (i & 1)));
                int label = -1;
                const int id = mc::vm_failed_assert(&P, vals.data(), &label);
                if (id >= 0) {
                    msg = "Failure of assertion at line " + std::to_string(P.asserts[(size_t)id].line) + ", column " + std::to_string(P.asserts[(size_t)id].col) + ".";
                    assert_def = P.strings[(size_t)label];
                }
            }
            o.put("The first argument of Assert evaluated to FALSE; the second argument was:\n\"%s\"\n", msg.c_str());
        }
        if (have_trace) {
            o.put("Error: The behavior up to this point is:\n");
            std::vector<char> txt(1 << 16);
            for (size_t k = 0; k < n; k++) {
                mc_state_format(&d, &states[k * W], txt.data(), txt.size());
                if (acts[k] < 0) { o.put("State %zu: <Initial predicate>\n%s\n\n", k + 1, txt.data()); continue; }
                const char *an = mc_action_name(&d, acts[k]);
                const Span sp = definition_span(srcL, an);
                if (sp.ok)  // README.md:278
                    o.put("State %zu: <Action line %d, col %d to line %d, col %d of module %s>\n%s\n\n", k + 1, sp.l1, sp.c1, sp.l2, sp.c2,
                          def_module.c_str(), txt.data());
                else
                    o.put("State %zu: <Action %s of module %s>\n%s\n\n", k + 1, an, def_module.c_str(), txt.data());
            }
        }
        if (res->verdict == MC_V_ASSERT) {  // README.md:313-316: the conjunct being evaluated and the Assert call itself
            int first = 0;
            const Span c = definition_span(srcL, assert_def.c_str(), &first);
            if (c.ok) {
                Span a0, a1;
                const std::string &l0 = srcL[first];
                size_t b = (size_t)c.c1 - 1;
                if (l0.compare(b, 2, "/\\") == 0) { b += 2; while (b < l0.size() && l0[b] == ' ') b++; }
                a0.l1 = a0.l2 = c.l1; a0.c1 = (int)b + 1; a0.c2 = last_nonspace(l0);
                for (int i = first; i < c.l2; i++) {
                    const size_t at = srcL[i].find("Assert(");
                    if (at == std::string::npos) continue;
                    a1.l1 = i + 1; a1.c1 = (int)at + 1;
                    int depth = 0;
                    for (int j = i; j < c.l2 && !a1.ok; j++)
                        for (size_t q = (j == i ? at : 0); q < srcL[j].size(); q++) {
                            if (srcL[j][q] == '(') depth++;
                            else if (srcL[j][q] == ')' && --depth == 0) { a1.l2 = j + 1; a1.c2 = (int)q + 1; a1.ok = true; break; }
                        }
                    break;
                }
                if (a1.ok) {
                    o.put("Error: The error occurred when TLC was evaluating the nested\nexpressions at the following positions:\n");
                    o.put("0. Line %d, column %d to line %d, column %d in %s\n", a0.l1, a0.c1, a0.l2, a0.c2, def_module.c_str());
                    o.put("1. Line %d, column %d to line %d, column %d in %s\n\n\n", a1.l1, a1.c1, a1.l2, a1.c2, def_module.c_str());
                }
            }
        }
    }
    o.put("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)res->generated,
          (unsigned long long)res->distinct, (unsigned long long)res->queue_left);
    o.put("The depth of the complete state graph search is %u.\n", res->depth);
    if (dump_path) {  // TLC -dump: every distinct state, in the order it was found
        FILE *f = fopen(dump_path, "w");
        if (!f) { mc_engine_destroy(e); return fe_fail(MC_EPARSE, "cannot write %s", dump_path); }
        const size_t W = mc_state_bytes(&d);
        const uint64_t batch = 1 << 14;
        std::vector<uint8_t> buf(batch * W);
        std::vector<char> txt(1 << 16);
        for (uint64_t first = 0; first < res->distinct; first += batch) {
            const uint64_t n = res->distinct - first < batch ? res->distinct - first : batch;
            if ((rc = mc_engine_read_states(e, first, n, buf.data()))) { fclose(f); mc_engine_destroy(e); return rc; }
            for (uint64_t k = 0; k < n; k++) {
                mc_state_format(&d, &buf[k * W], txt.data(), txt.size());
                fprintf(f, "State %llu:\n%s\n\n", (unsigned long long)(first + k + 1), txt.data());
            }
        }
        fclose(f);
    }
    mc_engine_destroy(e);
    return MC_OK;
}

}  // extern "C"
