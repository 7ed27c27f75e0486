// pcal.h — PlusCal front-end (p-syntax): parser, pcal2tla-style translator, compiler to the bytecode of
// spec_vm.h.  Host only.
//
// The reference keeps its two root specs UNtranslated (pcal_intro.tla:4-19, atomic_add.tla:4-23) and its
// Makefile runs `pcal2tla *tla` before `tlc *tla` (Makefile:3-7).  This is the `transpile` half: the
// translation follows examples/p-manual.pdf §3.8 and App. B (one action per label guarded by pc[self],
// `await` as an enabling conjunct, Assert(...) with the source position, the terminating disjunct), in the
// layout of the translator whose output the README quotes (action spans README.md:278-318).
#pragma once
#include <memory>
#include <string>
#include <vector>

namespace pcal {

struct Pos { int line = 0, col = 0; };

struct Expr;
using EP = std::shared_ptr<Expr>;
struct Expr {
    enum K { NUM, STR, BOOL, ID, UNOP, BINOP, INDEX, IF, QUANT, SETENUM, TUPLE, FUNCDEF, PRIME, CALL, RECORD, DOT, SETOF } k = NUM;
    long long num = 0;      // NUM, BOOL
    std::string s;          // ID name, STR text, operator text, QUANT "\\A" / "\\E", CALL operator name, DOT field name
    std::string bound;      // QUANT / FUNCDEF / SETOF bound variable (SETOF: s = "filter" {x \in S : P} or "map" {e : x \in S}; a = [domain, P or e])
    std::vector<EP> a;      // operands; INDEX: [fn, index]; IF: [c, t, e]; QUANT / FUNCDEF: [domain, body]; DOT: [record]; RECORD: the values
    std::vector<std::string> names;  // RECORD [f |-> e, g |-> h]: the field names, in the order written
    bool paren = false;     // written inside ( )
    Pos pos;
};

struct Stmt;
using SP = std::shared_ptr<Stmt>;
struct Stmt {
    enum K { ASSIGN, IF, WHILE, EITHER, WITH, AWAIT, ASSERT, SKIP, GOTO, PRINT, CALL, RETURN } k = SKIP;
    std::string label;                     // "" = unlabeled
    Pos pos;                               // of the statement keyword / lhs (asserts print it)
    std::string var;                       // ASSIGN lhs, WITH variable, GOTO target
    EP idx;                                // ASSIGN lhs index (x[i] := e), may be null
    std::string field;                     // ASSIGN lhs field path (r.f := e, r[i].f := e, r.f.g := e: "f.g"); gone after flatten_records
    std::string whole;                     // ASSIGN to a variable that flatten_records made from the record variable `whole`
    EP e;                                  // ASSIGN rhs; condition of IF / WHILE / AWAIT / ASSERT; WITH set or value
    bool with_eq = false;                  // with x = e
    std::vector<std::vector<SP>> blocks;   // IF: then, else; EITHER: branches; WHILE / WITH: body
    std::vector<SP> more;                  // ASSIGN: the other assignments of `a := e || b := f` (all right-hand sides
                                           // see the values before the statement)
    std::vector<EP> args;                  // CALL: the arguments (`var` = the procedure's name); gone after expand_procedures
};

struct VarDecl {
    std::string name;
    bool in_set = false;   // `x \in S` instead of `x = e`
    bool no_init = false;  // `variable x;`: the translation initialises it to the model value defaultInitValue
    EP init;
    Pos pos;
};

struct Proc {
    std::string name;
    bool is_set = false;   // process P \in S   vs   process P = e
    EP id;
    std::vector<VarDecl> locals;
    std::vector<SP> body;
};

// procedure P(a, b = e) variables x = e; begin ... return; end procedure   (p-manual section 3.5)
struct Procedure {
    std::string name;
    std::vector<VarDecl> params;   // `a` (initially defaultInitValue) or `a = e`
    std::vector<VarDecl> locals;   // (re)initialised on every call
    std::vector<SP> body;
    Pos pos;
};

// A variable whose initial value is a record constructor (or a function to records): kept FIELD BY FIELD (flatten_records, pcal.cpp).
struct RecordVar {
    std::string name;                  // r: the variables of the translation are r_f, one per field
    std::vector<std::string> fields;
    bool array = false;                // r = [x \in S |-> [f |-> e, ...]]
    bool seq = false;                  // a SEQUENCE of records (q = <<>> ... Append(q, [f |-> e, ...]); with `array`: [x \in S |-> <<>>], the
                                       // channels of a message-passing algorithm): one sequence per field, all of the same length
    std::string bound;                 // array: x
    EP domain;                         // array: S
    bool set = false;                  // a SET of records (msgs = {} ... msgs := msgs \cup {[type |-> "1a", bal |-> b]}): stays ONE set-valued variable of the
                                       // translation (as pcal2tla keeps it); the compiled program keeps its elements sorted, field by field
    EP shape;                          // seq / set: the constructor that says which fields (and of which types) an element has
    int proc = -1;                     // index into Module::procs of the process it is local to, -1 = global
    std::vector<std::string> sub;      // the fields that are records themselves (nested records): r_f is a RecordVar of depth + 1
    int depth = 0;                     // 0 = a variable of the algorithm, k = a field of a record of depth k - 1
};

struct Definition {
    std::string name;
    std::vector<std::string> params;
    EP body;
    int line = 0;
    bool in_define = false;   // from the algorithm's `define` block (printed in the translation)
};
struct Macro {
    std::string name;
    std::vector<std::string> params;
    std::vector<SP> body;
};

struct Module {
    std::string name;
    std::string algorithm;
    std::vector<std::string> constants;     // CONSTANT(S) declared by the module
    std::vector<VarDecl> globals;
    std::vector<Proc> procs;                // a uniprocess algorithm is one Proc with an empty name
    std::vector<Definition> defs;           // definitions of the `define` block and of the module around the algorithm
    std::vector<Macro> macros;
    std::vector<Procedure> procedures;      // as written; expand_procedures() has inlined them into the processes' bodies
    bool had_procedures = false;
    std::vector<RecordVar> records;         // record variables, replaced by their fields' variables in globals / locals
    int alg_first_line = 0, alg_last_line = 0;   // lines of "(* --algorithm" and "end algorithm *)"
    bool has_translation = false;
    int tr_first_line = 0, tr_last_line = 0;     // "\* BEGIN TRANSLATION" .. "\* END TRANSLATION"
};

// Parse the PlusCal algorithm (and the definitions around it) of a module text.  Returns "" or an error message.
// PROCEDURES (round 4): parsed and then EXPANDED into the calling processes — every call site gets its own copy of the procedure's
// body (labels `Label_pK`, K = the number of the copy), `call` becomes "parameters := arguments || locals := their initial values;
// goto the copy's first label" and `return` "parameters, locals := their initial values; goto the label after the call", all in
// the step the statement was in.  For NON-RECURSIVE procedures this is a bijection on states with the translation pcal2tla gives
// (pc + the contents of `stack` <-> the copy's pc; a frame's saved values are the initial values, which `return` restores), so
// distinct / generated / depth and every verdict agree (tests/test_pcal.py checks it against hand-written stack translations);
// what differs is the TEXT of the translation and of a printed state: no `stack` variable, the copies' label names.
// `call P(..); return` (pcal2tla's tail call) is refused with a message.
// RECURSIVE procedures (round 5; a procedure that can reach itself through `call`, mutual recursion included) get ONE copy of the body per
// calling process and a BOUNDED call stack per procedure, kept as plain variables: the depth counter P_sp, and for every level
// K = 1 .. D (D = $TLAMC_PCAL_STACK, default 4) the return-site code P_retK and one slot vK per parameter / procedure variable v — what a
// frame of pcal2tla's `stack` holds (the return pc and the values the procedure's variables had BEFORE the call).  `call`: assert
// P_sp < D (a run that needs a deeper stack fails THERE, with the call's position, instead of being cut short silently), fill level
// P_sp + 1, parameters := arguments, locals := initial values, goto the body's first label; `return`: branch on the top frame's
// return-site code, restore the variables from it, put the frame back to its initial values (two states must not differ in what lies
// above the stack), pop, goto the site.  pc + `stack` <-> pc + (P_sp, frames) is again a bijection on states — the return sites tell
// how the per-procedure stacks interleave — so counts, depth and verdicts equal those of pcal2tla's translation for every run that
// stays within D frames (tests/test_pcal.py: the hand-written stack translation tests/golden/pcal_recursion/RecursiveSumStack.tla).
// RECORDS (round 4): a variable initialised with `[f |-> e, g |-> h]` (or `[x \in S |-> [f |-> e, ...]]`) is kept field by field —
// variables r_f, r_g; `r.f` / `r[i].f` read them, `r.f := e` / `r[i].f := e` assign one, `r := [f |-> a, g |-> b]`, `r := s`
// (s another record variable or an element of a record array) assign all fields at once (`||`), `r = s` / `r # s` compare
// fieldwise.  A record value <-> the tuple of its fields is a bijection on states, so counts, depth and verdicts equal those of
// pcal2tla's translation (which keeps r one record-valued variable: tests/test_pcal.py checks it against hand-written record
// translations); the translation's TEXT differs: it declares r_f, r_g and defines `r == [f |-> r_f, g |-> r_g]` after them, so
// that the text around the algorithm (invariants written with r.f) keeps its meaning.  NESTED records (round 5): a field may itself
// be a record; the flattener works one level per pass (r -> r_f -> r_f_g), `r.f.g := e` carries its path in Stmt::field ("f.g"), an
// inner record is assigned / compared leaf by leaf, the translation defines every level, inner before outer.  Refused with a message:
// a record as a whole value anywhere else (`with` over a record, procedure arguments).  SEQUENCES of records (and arrays of them: channels)
// are kept as one sequence per field; a SET of records (a message soup) stays one set-valued variable of the translation and is kept as
// sorted cells by the compiled program (RecordVar::seq / ::set below).
std::string parse_module(const std::string &text, Module &out);

// The text `pcal2tla` inserts: from "\* BEGIN TRANSLATION" to "\* END TRANSLATION" inclusive, '\n' terminated.
std::string translate(const Module &m);

// The whole module with the translation inserted after the algorithm comment (replacing an existing one).
std::string transpile_text(const std::string &text, const Module &m);

// ------------------------------------------------------------------------------------------------
// Compilation to the stack machine of spec_vm.h.
struct ConstVal {
    enum K { INT, STR, SET } k = INT;
    long long i = 0;
    std::string s;                 // STR: a string or a model value (both become interned strings)
    std::vector<ConstVal> elems;   // SET
};
struct Config {
    std::vector<std::string> invariants;                       // INVARIANT names, in cfg order
    std::vector<std::string> constraints;                      // CONSTRAINT names
    std::vector<std::pair<std::string, ConstVal>> constants;   // CONSTANT name = value
};

struct VarInfo {
    std::string name;
    bool array = false;       // a function over `ids` (flattened to consecutive variables)
    int base = 0;             // index of the first 32-bit variable
    std::vector<long long> ids;  // domain of the array (process ids for pc and process locals)
    char type = 'i';          // 'i' integer, 'b' boolean, 's' interned string (of the elements for array / seq)
    bool set = false;         // a set of naturals 0..31 (or of interned strings), one cell holding the 32-bit mask
    bool seq = false;         // a bounded sequence: cell `base` = Len, then `cap` element cells
    bool defval = false;      // declared without an initial value: a cell holding VM_DEFAULT_INIT prints as defaultInitValue
    int cap = 0;
    bool rset = false;        // a SET of records: cell `base` = the number of elements, field f of element i at base + 1 + f * cap + i (sorted, spec_vm.h VM_RSADD)
    std::vector<std::string> fields;   // rset: the field names, in the order of the cells
    std::string ftypes;                // rset: their types ('i' / 'b' / 's')
};

struct Program {
    int magic = 0x70634c31;
    std::vector<int> image;               // what the device reads (header, tables, code)
    std::vector<VarInfo> vars;            // VARIABLES order: globals, pc, process locals
    std::vector<std::string> strings;     // interned strings; the first `nlabels` are the labels, "Done" included
    int nlabels = 0, nv = 0, ninst = 0, maxch = 1, pc_base = 0;
    bool multi = true;
    unsigned long long num_init = 1;
    std::vector<std::string> invariants;
    struct AssertPos { int line, col; };
    std::vector<AssertPos> asserts;
    std::string module, translated;       // module name; the module text with the translation inserted
};

// Returns "" or an error message.
std::string compile(const Module &m, const std::string &module_text, const Config &cfg, Program &out);

}  // namespace pcal
