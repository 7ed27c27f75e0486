"""oracle/tlaplus.py — TEST INFRASTRUCTURE, not product code.

An explicit-state checker that evaluates TLA+ MODULE TEXT the way TLC does, written so that the reference's own spec
files (/root/reference/examples/raft.tla, serializableSnapshotIsolation.tla, textbookSnapshotIsolation.tla, read where they
lie, never copied) can be run directly under the model wrappers of specs/.  Its job is to PIN the hand-written C oracle
(oracle/spec_raft.c, oracle/spec_ssi.c) and through it the HIP lowerings: both of those are restatements by one author of
one reading of the spec; this evaluator shares no code and no reading with them — it only knows TLA+.

What "the way TLC does" means here (TLC is the external Java tool of reference Makefile:6-7; p-manual.pdf section 4,
Specifying Systems ch. 14 as quoted in SURVEY.md App. B):
  * an action is evaluated left to right; `x' = e` ASSIGNS x' when x' has no value yet and is an equality TEST otherwise
    (this is what makes raft.tla:392-393 + :402 enable the "already done" branch only when m.mcommitIndex =
    commitIndex[i]); `x' \\in S` enumerates; UNCHANGED <<a, b>> is a' = a /\\ b' = b; `\\/`, `\\E`, IF, CASE and LET
    bodies branch; operator applications are expanded with lazily evaluated arguments; a successor with an unassigned
    variable is an error;
  * breadth-first search with exact de-duplication on whole states; counters as TLC prints them (README.md:319-321):
    generated = initial states + every successor produced (duplicates and out-of-CONSTRAINT ones included), distinct,
    queue, depth; a successor outside the CONSTRAINT is invariant-checked (TLC checks every unseen successor) but neither
    stored nor expanded; deadlock = no successor at all;
  * CHOOSE takes the first satisfying element in a fixed total order on values (model values in cfg order).

Pure Python with the expression tree compiled to closures: meant for models of 10^3..10^5 states.
Only tests/, tests/golden/*.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
"""
import itertools
import re
import sys
from pathlib import Path

sys.setrecursionlimit(20000)


class TLAError(Exception):
    """an evaluation error TLC would report (function applied outside its domain, CHOOSE without witness, ...)"""


class AssertFail(TLAError):
    pass


# =============================================================================================== lexer
class Tok:
    __slots__ = ("k", "s", "line", "col")

    def __init__(self, k, s, line, col):
        self.k, self.s, self.line, self.col = k, s, line, col

    def __repr__(self):
        return f"{self.k}:{self.s}@{self.line}:{self.col}"


SYMS = ["<=>", "|->", "-+->", "::=", ":=", "==", "=>", "=<", "<=", ">=", "/=", "/\\", "\\/", "..", "->", "<-", "<<", ">>", ":>",
        "@@", "[]", "<>", "~>", "||", "(", ")", "[", "]", "{", "}", ",", ";", ":", "+", "-", "*", "/", "%", "=", "<", ">", "#",
        "~", "'", "!", "@", ".", "^", "|", "&", "\\"]


def lex(text):
    toks, i, line, col, n = [], 0, 1, 1, len(text)

    def adv(k):
        nonlocal i, line, col
        for _ in range(k):
            if text[i] == "\n":
                line, col = line + 1, 1
            else:
                col += 1
            i += 1

    while i < n:
        c = text[i]
        if c in " \t\r\n":
            adv(1)
        elif text.startswith("\\*", i):
            while i < n and text[i] != "\n":
                adv(1)
        elif text.startswith("(*", i):
            depth = 1
            adv(2)
            while i < n and depth:
                if text.startswith("(*", i):
                    depth += 1
                    adv(2)
                elif text.startswith("*)", i):
                    depth -= 1
                    adv(2)
                else:
                    adv(1)
        elif text.startswith("----", i) or text.startswith("====", i):
            j = i
            while j < n and text[j] == c:
                j += 1
            toks.append(Tok("sep", text[i:j], line, col))
            adv(j - i)
        elif c.isalnum() or c == "_":
            j = i
            while j < n and (text[j].isalnum() or text[j] == "_"):
                j += 1
            s = text[i:j]
            toks.append(Tok("num" if s.isdigit() else "id", s, line, col))
            adv(j - i)
        elif c == '"':
            j = i + 1
            buf = []
            while text[j] != '"':
                if text[j] == "\\":
                    j += 1
                buf.append(text[j])
                j += 1
            toks.append(Tok("str", "".join(buf), line, col))
            adv(j + 1 - i)
        elif c == "\\" and i + 1 < n and text[i + 1].isalpha():
            j = i + 1
            while j < n and text[j].isalpha():
                j += 1
            toks.append(Tok("sym", text[i:j], line, col))
            adv(j - i)
        else:
            for s in SYMS:
                if text.startswith(s, i):
                    toks.append(Tok("sym", s, line, col))
                    adv(len(s))
                    break
            else:
                raise SyntaxError(f"unexpected character {c!r} at line {line}, column {col}")
    toks.append(Tok("end", "", line, 0))
    return toks


# =============================================================================================== parser
# (low, high) precedence ranges of Specifying Systems table 6 reduced to one number; all infix operators parse
# left-associatively except =>
PREC = {"=>": 1, "<=>": 2, "\\equiv": 2, "~>": 2, "-+->": 2, "\\/": 3, "\\lor": 3, "/\\": 3, "\\land": 3,
        "=": 5, "#": 5, "/=": 5, "<": 5, ">": 5, "<=": 5, "=<": 5, ">=": 5, "\\leq": 5, "\\geq": 5, "\\in": 5, "\\notin": 5,
        "\\subseteq": 5, "\\subset": 5, "@@": 6, ":>": 7, "\\cup": 8, "\\union": 8, "\\cap": 8, "\\intersect": 8, "\\": 8,
        "..": 9, "+": 10, "-": 10, "%": 11, "\\X": 11, "\\times": 11, "*": 13, "/": 13, "\\div": 13, "\\o": 13, "\\circ": 13,
        "\\cdot": 5, "^": 14}
CANON = {"=<": "<=", "\\leq": "<=", "\\geq": ">=", "/=": "#", "\\union": "\\cup", "\\intersect": "\\cap", "\\lor": "\\/",
         "\\land": "/\\", "\\equiv": "<=>", "\\times": "\\X", "\\circ": "\\o"}
KEYWORDS = {"MODULE", "EXTENDS", "CONSTANT", "CONSTANTS", "VARIABLE", "VARIABLES", "ASSUME", "ASSUMPTION", "AXIOM", "THEOREM",
            "LEMMA", "PROPOSITION", "COROLLARY", "RECURSIVE", "INSTANCE", "WITH", "LOCAL", "IF", "THEN", "ELSE", "CASE", "OTHER",
            "LET", "IN", "CHOOSE", "LAMBDA", "EXCEPT", "UNCHANGED", "ENABLED", "SUBSET", "UNION", "DOMAIN", "PROOF", "BY", "OBVIOUS",
            "OMITTED", "QED"}
UNIT_STARTS = {"EXTENDS", "CONSTANT", "CONSTANTS", "VARIABLE", "VARIABLES", "ASSUME", "ASSUMPTION", "AXIOM", "THEOREM", "LEMMA",
               "PROPOSITION", "COROLLARY", "RECURSIVE", "INSTANCE", "LOCAL"}


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.jstack = [0]  # columns of the enclosing junction-list bullets: a token at or left of the top ends the item

    def cur(self):
        return self.t[self.i]

    def peek(self, k=1):
        return self.t[min(self.i + k, len(self.t) - 1)]

    def is_sym(self, s):
        c = self.t[self.i]
        return c.k == "sym" and c.s == s

    def is_id(self, s):
        c = self.t[self.i]
        return c.k == "id" and c.s == s

    def fail(self, what):
        c = self.cur()
        raise SyntaxError(f"{what} at line {c.line}, column {c.col} (near {c.s!r})")

    def expect(self, s):
        c = self.cur()
        if c.s != s or c.k not in ("sym", "id"):
            self.fail(f"expected {s!r}")
        self.i += 1

    def ended(self):
        c = self.cur()
        return c.k in ("end", "sep") or c.col <= self.jstack[-1]

    def ident(self):
        c = self.cur()
        if c.k != "id" or c.s in KEYWORDS:
            self.fail("expected an identifier")
        self.i += 1
        return c.s

    # ---- expressions
    def expr(self, minprec=0):
        lhs = self.prefix()
        while not self.ended():
            c = self.cur()
            if c.k != "sym" or c.s not in PREC:
                break
            p = PREC[c.s]
            if p < minprec:
                break
            self.i += 1
            rhs = self.expr(p if c.s == "=>" else p + 1)
            op = CANON.get(c.s, c.s)
            if op == "/\\":
                lhs = ("conj", [lhs, rhs])
            elif op == "\\/":
                lhs = ("disj", [lhs, rhs])
            else:
                lhs = ("op", op, lhs, rhs)
        return lhs

    def junction(self, bullet):
        col = self.cur().col
        items = []
        while self.cur().k == "sym" and self.cur().s == bullet and self.cur().col == col:
            self.i += 1
            self.jstack.append(col)
            items.append(self.expr(0))
            self.jstack.pop()
        return ("conj" if bullet == "/\\" else "disj", items)

    def bounds(self):
        """x \\in S | x, y \\in S | x \\in S, y \\in T | <<a, b>> \\in S  ->  [(pattern, set expr)]; pattern = name or tuple of names"""
        out = []
        while True:
            pats = []
            while True:
                if self.is_sym("<<"):
                    self.i += 1
                    names = [self.ident()]
                    while self.is_sym(","):
                        self.i += 1
                        names.append(self.ident())
                    self.expect(">>")
                    pats.append(tuple(names))
                else:
                    pats.append(self.ident())
                if self.is_sym(","):
                    self.i += 1
                    continue
                break
            self.expect("\\in")
            dom = self.expr(6)
            out.extend((p, dom) for p in pats)
            if self.is_sym(","):
                self.i += 1
                continue
            return out

    def definitions(self, stop_in):
        """operator definitions of a LET (stop_in) or of the module body; returns [(name, params, body)]"""
        defs = []
        while True:
            c = self.cur()
            if stop_in and self.is_id("IN"):
                return defs
            if c.k == "id" and c.s == "RECURSIVE":
                self.i += 1
                while True:
                    self.ident()
                    if self.is_sym("("):
                        while not self.is_sym(")"):
                            self.i += 1
                        self.i += 1
                    if self.is_sym(","):
                        self.i += 1
                        continue
                    break
                continue
            if c.k != "id" or c.s in KEYWORDS:
                return defs
            defs.append(self.definition())

    def definition(self):
        line = self.cur().line
        name = self.ident()
        params = []
        if self.is_sym("("):
            self.i += 1
            while True:
                p = self.ident()
                arity = 0
                if self.is_sym("("):
                    self.i += 1
                    while not self.is_sym(")"):
                        if self.is_id("_"):
                            arity += 1
                        self.i += 1
                    self.i += 1
                params.append((p, arity))
                if self.is_sym(","):
                    self.i += 1
                    continue
                break
            self.expect(")")
            self.expect("==")
            body = self.expr(0)
        elif self.is_sym("["):  # f[x \in S] == e  is  f == [x \in S |-> e] (recursive reference allowed)
            self.i += 1
            bs = self.bounds()
            self.expect("]")
            self.expect("==")
            body = ("fndef", bs, self.expr(0), name)
        else:
            self.expect("==")
            if self.is_id("INSTANCE"):
                self.i += 1
                mod = self.ident()
                subst = []
                if self.is_id("WITH"):
                    self.i += 1
                    while True:
                        a = self.ident()
                        self.expect("<-")
                        subst.append((a, self.expr(0)))
                        if self.is_sym(","):
                            self.i += 1
                            continue
                        break
                body = ("instance", mod, subst)
            else:
                body = self.expr(0)
        return (name, params, body, line)

    def prefix(self):
        c = self.cur()
        if c.k == "sym":
            if c.s in ("/\\", "\\/"):
                return self.junction(c.s)
            if c.s in ("~", "\\lnot", "\\neg"):
                self.i += 1
                return ("not", self.expr(4))
            if c.s == "-":
                self.i += 1
                return ("neg", self.expr(12))
            if c.s in ("\\A", "\\E"):
                self.i += 1
                bs = self.bounds()
                self.expect(":")
                return ("quant", c.s[1], bs, self.expr(0))
            if c.s in ("[]", "<>"):  # temporal formulas are parsed and never evaluated
                self.i += 1
                return ("temporal", c.s, self.expr(4))
        if c.k == "id":
            if c.s in ("WF_", "SF_") and self.t[self.i + 1].s == "<<":  # WF_<<v1, v2>>(A): fairness, parsed and never evaluated
                self.i += 1
                sub = self.expr(16)
                self.expect("(")
                act = self.expr(0)
                self.expect(")")
                return ("temporal", c.s, ("tuple", [sub, act]))
            if c.s == "CHOOSE":
                self.i += 1
                if self.cur().k == "id" and self.t[self.i + 1].s == ":":
                    # unbounded CHOOSE x : P (serializableSnapshotIsolation.tla:24 NoLock): TLC cannot evaluate it either — the cfg
                    # replaces the definition by a model value (CachingMemory/MCInternalMemory.cfg:24-27); parsed, never evaluated
                    name = self.ident()
                    self.expect(":")
                    return ("choose_unbounded", name, self.expr(0))
                bs = self.bounds()
                self.expect(":")
                return ("choose", bs, self.expr(0))
            if c.s == "IF":
                self.i += 1
                cond = self.expr(0)
                self.expect("THEN")
                a = self.expr(0)
                self.expect("ELSE")
                return ("if", cond, a, self.expr(0))
            if c.s == "CASE":
                self.i += 1
                arms, other = [], None
                while True:
                    if self.is_id("OTHER"):
                        self.i += 1
                        self.expect("->")
                        other = self.expr(0)
                    else:
                        g = self.expr(0)
                        self.expect("->")
                        arms.append((g, self.expr(0)))
                    if self.is_sym("[]") and not self.ended():
                        self.i += 1
                        continue
                    break
                return ("case", arms, other)
            if c.s == "LET":
                self.i += 1
                saved = self.jstack
                self.jstack = [0]  # LET definitions may start left of an enclosing bullet's item text? no: keep simple
                self.jstack = saved
                defs = self.definitions(stop_in=True)
                self.expect("IN")
                return ("let", defs, self.expr(0))
            if c.s == "LAMBDA":
                self.i += 1
                names = [self.ident()]
                while self.is_sym(","):
                    self.i += 1
                    names.append(self.ident())
                self.expect(":")
                return ("lambda", names, self.expr(0))
            if c.s == "UNCHANGED":
                self.i += 1
                return ("unchanged", self.expr(14))
            if c.s == "ENABLED":
                self.i += 1
                return ("enabled", self.expr(14))
            if c.s in ("SUBSET", "UNION", "DOMAIN"):
                self.i += 1
                return ("pre", c.s, self.expr(9))
        return self.postfix(self.atom())

    def exprlist(self, close):
        items = []
        if not self.is_sym(close):
            items.append(self.expr(0))
            while self.is_sym(","):
                self.i += 1
                items.append(self.expr(0))
        self.expect(close)
        return items

    def atom(self):
        c = self.cur()
        self.i += 1
        if c.k == "num":
            return ("num", int(c.s))
        if c.k == "str":
            return ("str", c.s)
        if c.k == "id" and c.s not in KEYWORDS:
            if c.s == "TRUE" or c.s == "FALSE":
                return ("bool", c.s == "TRUE")
            name = c.s
            while self.is_sym("!") and self.peek().k == "id":  # Inst!Op
                self.i += 1
                name = name + "!" + self.ident()
            if self.is_sym("!") and self.peek().k == "sym" and self.peek().s == ":":  # Thm!: = the statement of Thm
                self.i += 2
                return ("id", name)
            if self.is_sym("!") and self.peek().k == "num":  # Def!k = the k-th conjunct of Def's body (MCPaxos.tla:44-46)
                self.i += 1
                k = int(self.cur().s)
                self.i += 1
                return ("nth", name, k)
            if self.is_sym("(") and not self.ended():
                self.i += 1
                return ("call", name, self.exprlist(")"))
            return ("id", name)
        if c.k == "sym":
            if c.s == "(":
                e = self.expr(0)
                self.expect(")")
                return ("paren", e)
            if c.s == "@":
                return ("at",)
            if c.s == "{":
                if self.is_sym("}"):
                    self.i += 1
                    return ("setenum", [])
                # {x \in S : P}  |  {<<a, b>> \in S : P}
                save = self.i
                if self.cur().k == "id" and self.peek().k == "sym" and self.peek().s == "\\in" or self.is_sym("<<"):
                    try:
                        bs = self.bounds()
                        if len(bs) == 1 and self.is_sym(":"):
                            self.i += 1
                            pred = self.expr(0)
                            self.expect("}")
                            return ("setfilter", bs[0][0], bs[0][1], pred)
                    except SyntaxError:
                        pass
                    self.i = save
                first = self.expr(0)
                if self.is_sym(":"):  # {e : x \in S, y \in T}
                    self.i += 1
                    bs = self.bounds()
                    self.expect("}")
                    return ("setmap", first, bs)
                items = [first]
                while self.is_sym(","):
                    self.i += 1
                    items.append(self.expr(0))
                self.expect("}")
                return ("setenum", items)
            if c.s == "<<":
                items = self.exprlist(">>")
                if self.cur().k == "id" and self.cur().s.startswith("_") and not self.ended():  # <<A>>_v
                    bare = self.cur().s == "_"
                    sub = ("id", self.cur().s[1:])
                    self.i += 1
                    if bare:
                        sub = self.expr(16)   # <<A>>_<<v1, v2>>: the subscript is a tuple
                    if len(items) != 1:
                        self.fail("<<A>>_v takes one action")
                    return ("temporal", "<<>>_", items[0], sub)
                return ("tuple", items)
            if c.s == "[":
                n1 = self.peek()
                if self.cur().k == "id" and n1.k == "sym" and n1.s in ("|->", ":") and self.cur().s not in KEYWORDS:
                    kind = "record" if n1.s == "|->" else "recordset"
                    fields = []
                    while True:
                        f = self.ident()
                        self.expect(n1.s)
                        fields.append((f, self.expr(0)))
                        if self.is_sym(","):
                            self.i += 1
                            continue
                        break
                    self.expect("]")
                    return (kind, fields)
                # [x \in S, y \in T |-> e]
                save = self.i
                if (self.cur().k == "id" and n1.k == "sym" and n1.s in ("\\in", ",")) or self.is_sym("<<"):
                    try:
                        bs = self.bounds()
                        if self.is_sym("|->"):
                            self.i += 1
                            body = self.expr(0)
                            self.expect("]")
                            return ("fndef", bs, body, None)
                    except SyntaxError:
                        pass
                    self.i = save
                first = self.expr(0)
                if self.is_id("EXCEPT"):
                    self.i += 1
                    ups = []
                    while True:
                        self.expect("!")
                        path = []
                        while True:
                            if self.is_sym("["):
                                self.i += 1
                                idx = self.exprlist("]")
                                path.append(idx[0] if len(idx) == 1 else ("tuple", idx))
                            elif self.is_sym("."):
                                self.i += 1
                                path.append(("str", self.ident()))
                            else:
                                break
                        self.expect("=")
                        ups.append((path, self.expr(0)))
                        if self.is_sym(","):
                            self.i += 1
                            continue
                        break
                    self.expect("]")
                    return ("except", first, ups)
                if self.is_sym("->"):
                    self.i += 1
                    rng = self.expr(0)
                    self.expect("]")
                    return ("fnset", first, rng)
                self.expect("]")
                if self.cur().k == "id" and self.cur().s.startswith("_"):  # [A]_v
                    bare = self.cur().s == "_"
                    sub = ("id", self.cur().s[1:])
                    self.i += 1
                    if bare:
                        sub = self.expr(16)   # [A]_<<v1, v2>>: the subscript is a tuple
                    return ("temporal", "[]_", first, sub)
                self.fail("unsupported bracket expression")
        self.i -= 1
        self.fail("expected an expression")

    def postfix(self, e):
        while not self.ended():
            if self.is_sym("["):
                self.i += 1
                idx = self.exprlist("]")
                e = ("idx", e, idx[0] if len(idx) == 1 else ("tuple", idx))
            elif self.is_sym("."):
                self.i += 1
                e = ("idx", e, ("str", self.ident()))
            elif self.is_sym("'"):
                self.i += 1
                e = ("prime", e)
            else:
                break
        return e


class Module:
    """parsed MODULE text: name, extends, constants (name -> arity), variables (ordered), definitions"""

    PROOF_LINE = re.compile(r"^\s*(<\d+>\w*\.?|BY\b|QED\b|OBVIOUS\b|OMITTED\b|PROOF\b).*$", re.M)

    def __init__(self, text):
        # structured proofs (examples/Paxos/Voting.tla:187-196, Consensus.tla:40-47) are not evaluated: their lines are blanked
        toks = lex(self.PROOF_LINE.sub("", text))
        p = Parser(toks)
        # skip to  ---- MODULE name ----
        while not (p.cur().k == "id" and p.cur().s == "MODULE"):
            if p.cur().k == "end":
                raise SyntaxError("no MODULE header")
            p.i += 1
        p.i += 1
        self.name = p.cur().s
        p.i += 1
        if p.cur().k == "sep":
            p.i += 1
        self.extends, self.constants, self.variables, self.defs, self.assumes, self.instances = [], {}, [], {}, [], []
        self.def_order = []
        while True:
            c = p.cur()
            if c.k == "end" or (c.k == "sep" and c.s.startswith("=")):
                break
            if c.k == "sep":
                p.i += 1
                continue
            if c.k != "id":
                p.fail("expected a module unit")
            if c.s == "LOCAL":
                p.i += 1
                continue
            if c.s == "EXTENDS":
                p.i += 1
                self.extends.append(p.ident())
                while p.is_sym(","):
                    p.i += 1
                    self.extends.append(p.ident())
            elif c.s in ("CONSTANT", "CONSTANTS"):
                p.i += 1
                while True:
                    nm = p.ident()
                    arity = 0
                    if p.is_sym("("):
                        p.i += 1
                        while not p.is_sym(")"):
                            if p.is_id("_"):
                                arity += 1
                            p.i += 1
                        p.i += 1
                    self.constants[nm] = arity
                    if p.is_sym(","):
                        p.i += 1
                        continue
                    break
            elif c.s in ("VARIABLE", "VARIABLES"):
                p.i += 1
                self.variables.append(p.ident())
                while p.is_sym(","):
                    p.i += 1
                    self.variables.append(p.ident())
            elif c.s in ("ASSUME", "ASSUMPTION", "AXIOM"):
                p.i += 1
                nm = None
                if p.cur().k == "id" and p.peek().k == "sym" and p.peek().s == "==":
                    nm = p.cur().s
                    p.i += 2
                e = p.expr(0)
                self.assumes.append(e)
                if nm:  # a named assumption is a definition (Name!: is its body, MCVoting.tla:32)
                    self.defs[nm] = (nm, [], e, c.line)
                    self.def_order.append(nm)
            elif c.s in ("THEOREM", "LEMMA", "PROPOSITION", "COROLLARY"):
                p.i += 1
                nm = None
                if p.cur().k == "id" and p.peek().k == "sym" and p.peek().s == "==":
                    nm = p.cur().s
                    p.i += 2
                e = p.expr(0)  # never evaluated unless a model refers to it as Name!: (MCVoting.tla:42-46)
                if nm:
                    self.defs[nm] = (nm, [], e, c.line)
                    self.def_order.append(nm)
                while p.cur().k == "id" and p.cur().s in ("PROOF", "BY", "OBVIOUS", "OMITTED", "QED"):
                    p.i += 1
            elif c.s == "INSTANCE":
                p.i += 1
                self.instances.append(p.ident())
            else:
                for d in p.definitions(stop_in=False):
                    self.defs[d[0]] = d
                    self.def_order.append(d[0])
                if p.cur() is c:
                    p.fail("cannot parse module unit")


# =============================================================================================== values
class MV:
    """a model value (cfg `c = c`, or an element of `S = {a, b}`): equal only to itself, printed bare, ordered by creation"""
    __slots__ = ("name", "idx")
    _all = {}

    def __new__(cls, name):
        v = cls._all.get(name)
        if v is None:
            v = object.__new__(cls)
            v.name, v.idx = name, len(cls._all)
            cls._all[name] = v
        return v

    def __repr__(self):
        return self.name

    def __reduce__(self):  # a model value crosses a process boundary by name (tests/golden/make_tlc_log_golden.py)
        return (MV, (self.name,))


class Fn:
    """a function whose domain is NOT 1..n (those are Python tuples; the empty function is ())"""
    __slots__ = ("d", "_h")

    def __init__(self, d):
        self.d = d
        self._h = None

    def __hash__(self):
        if self._h is None:
            self._h = hash(frozenset(self.d.items()))
        return self._h

    def __eq__(self, o):
        return isinstance(o, Fn) and self.d == o.d

    def __ne__(self, o):
        return not self.__eq__(o)

    def __repr__(self):
        return fmt(self)

    def __reduce__(self):
        return (Fn, (self.d,))


def mk_fn(d):
    n = len(d)
    if n == 0:
        return ()
    if all(type(k) is int for k in d) and min(d) == 1 and max(d) == n:
        return tuple(d[i] for i in range(1, n + 1))
    return Fn(d)


def is_fn(v):
    return isinstance(v, (tuple, Fn))


def fn_domain(f):
    if isinstance(f, tuple):
        return frozenset(range(1, len(f) + 1))
    if isinstance(f, Fn):
        return frozenset(f.d)
    raise TLAError(f"DOMAIN of a non-function {fmt(f)}")


def fn_apply(f, a):
    if isinstance(f, tuple):
        if type(a) is int and 1 <= a <= len(f):
            return f[a - 1]
        raise TLAError(f"sequence of length {len(f)} applied to {fmt(a)}")
    if isinstance(f, Fn):
        try:
            return f.d[a]
        except KeyError:
            raise TLAError(f"function applied outside its domain: {fmt(a)} not in DOMAIN {fmt(f)}") from None
    raise TLAError(f"applying a non-function {fmt(f)} to {fmt(a)}")


def fn_items(f):
    if isinstance(f, tuple):
        return {i + 1: v for i, v in enumerate(f)}
    return dict(f.d)


# ---- sets that are not enumerated eagerly
class LazySet:
    def contains(self, v):
        raise NotImplementedError

    def __iter__(self):
        raise TLAError(f"cannot enumerate {self!r}")


class NatSet(LazySet):
    def __init__(self, lo=0):
        self.lo = lo

    def contains(self, v):
        return type(v) is int and (self.lo is None or v >= self.lo)

    def __repr__(self):
        return "Nat" if self.lo == 0 else "Int"


class StringSet(LazySet):
    def contains(self, v):
        return isinstance(v, str)


class SeqSet(LazySet):
    def __init__(self, s):
        self.s = s

    def contains(self, v):
        return isinstance(v, tuple) and all(set_in(x, self.s) for x in v)


class Powerset(LazySet):
    def __init__(self, s):
        self.s = s

    def contains(self, v):
        return isinstance(v, frozenset) and all(set_in(x, self.s) for x in v)

    def __iter__(self):
        el = sorted_vals(iter_set(self.s))
        for r in range(len(el) + 1):
            for c in itertools.combinations(el, r):
                yield frozenset(c)


class FnSet(LazySet):
    def __init__(self, s, t):
        self.s, self.t = s, t

    def contains(self, v):
        return is_fn(v) and fn_domain(v) == to_frozen(self.s) and all(set_in(x, self.t) for x in fn_items(v).values())

    def __iter__(self):
        dom = sorted_vals(iter_set(self.s))
        rng = sorted_vals(iter_set(self.t))
        for vals in itertools.product(rng, repeat=len(dom)):
            yield mk_fn(dict(zip(dom, vals)))


class RecSet(LazySet):
    def __init__(self, fields):
        self.fields = fields  # [(name, set)]

    def contains(self, v):
        return isinstance(v, Fn) and set(v.d) == {f for f, _ in self.fields} and all(set_in(v.d[f], s) for f, s in self.fields)

    def __iter__(self):
        names = [f for f, _ in self.fields]
        for vals in itertools.product(*[sorted_vals(iter_set(s)) for _, s in self.fields]):
            yield Fn(dict(zip(names, vals)))


def set_in(v, s):
    if isinstance(s, frozenset):
        return v in s
    if isinstance(s, LazySet):
        return s.contains(v)
    raise TLAError(f"\\in applied to a non-set {fmt(s)}")


def iter_set(s):
    if isinstance(s, (frozenset, LazySet)):
        return s
    raise TLAError(f"enumerating a non-set {fmt(s)}")


def to_frozen(s):
    if isinstance(s, frozenset):
        return s
    if isinstance(s, LazySet):
        return frozenset(s)
    raise TLAError(f"not a set: {fmt(s)}")


def vkey(v):
    """a total order on values (CHOOSE takes the first satisfying element in it)"""
    if v is True or v is False:
        return (0, v)
    if type(v) is int:
        return (1, v)
    if isinstance(v, str):
        return (2, v)
    if isinstance(v, MV):
        return (3, v.idx)
    if isinstance(v, tuple):
        return (4, len(v), tuple(vkey(x) for x in v))
    if isinstance(v, Fn):
        return (5, len(v.d), tuple(sorted((vkey(k), vkey(x)) for k, x in v.d.items())))
    if isinstance(v, frozenset):
        return (6, len(v), tuple(sorted(vkey(x) for x in v)))
    return (7, repr(v))


def sorted_vals(it):
    return sorted(it, key=vkey)


def fmt(v):
    """canonical TLA+ text of a value — the format of oracle/spec_raft.c raft_print / mc_state_format: records with fields in
    alphabetical order, functions as (k :> v @@ ...) and sets sorted by text, sequences <<...>>, the empty function <<>>"""
    if v is True:
        return "TRUE"
    if v is False:
        return "FALSE"
    if type(v) is int:
        return str(v)
    if isinstance(v, str):
        return '"' + v + '"'
    if isinstance(v, MV):
        return v.name
    if isinstance(v, tuple):
        return "<<" + ", ".join(fmt(x) for x in v) + ">>"
    if isinstance(v, Fn):
        if all(isinstance(k, str) for k in v.d):
            return "[" + ", ".join(f"{k} |-> {fmt(v.d[k])}" for k in sorted(v.d)) + "]"
        return "(" + " @@ ".join(sorted(f"{fmt(k)} :> {fmt(x)}" for k, x in v.d.items())) + ")"
    if isinstance(v, frozenset):
        return "{" + ", ".join(sorted(fmt(x) for x in v)) + "}"
    return repr(v)


# =============================================================================================== evaluator
class Thunk:
    """a lazily evaluated operator argument / LET definition without parameters"""
    __slots__ = ("f", "env", "st", "memo", "val", "done", "pvar")

    def __init__(self, f, env, st, memo):
        self.f, self.env, self.st, self.memo, self.done = f, env, st, memo, False
        self.val = None
        self.pvar = None

    def force(self, nx):
        if self.done:
            return self.val
        v = self.f(self.env, self.st, nx)
        if self.memo:
            self.val, self.done = v, True
        return v


class RecFn:
    """a recursively defined function while it is being built"""
    __slots__ = ("dom", "cache", "apply")

    def __init__(self, dom):
        self.dom, self.cache, self.apply = dom, {}, None


class OpVal:
    """an operator as a value: LAMBDA, an operator passed by name, a LET operator with parameters"""
    __slots__ = ("params", "cv", "ca", "env", "name")

    def __init__(self, params, cv, ca, env, name="LAMBDA"):
        self.params, self.cv, self.ca, self.env, self.name = params, cv, ca, env, name


class GDef:
    """a global (module-level) definition, compiled on first use"""

    def __init__(self, name, params, body, module, line):
        self.name, self.params, self.body, self.module, self.line = name, params, body, module, line
        self.cv = self.ca = None
        self.primed = None
        self.const_val = None
        self.is_const = None


UNASSIGNED = object()


class Spec:
    """a root module + everything it EXTENDS, constants bound by a cfg: compiles and evaluates expressions and actions"""

    BUILTIN_MODULES = {"Naturals", "Integers", "Reals", "FiniteSets", "Sequences", "TLC", "Bags", "RealTime", "TLAPS"}

    def __init__(self, path, search=(), constants=None, overrides=None, clash="test", scoped=None):
        self.search = [Path(path).parent] + [Path(s) for s in search]
        self.scoped = dict(scoped or {})  # (module, name) -> name: the cfg's `Id <-[Module] Id` (examples/Paxos/MCPaxos.cfg:9)
        self.scoped_overrides = {}
        self.clash = clash  # "test": TLC (a second x' = e is an equality test); "ignore": naive (first assignment wins, later ones TRUE)
        self.modules = []
        self.variables, self.constants, self.defs = [], {}, {}
        self._load(Path(path))
        self.varidx = {v: i for i, v in enumerate(self.variables)}
        self.const_vals = dict(constants or {})
        self.overrides = dict(overrides or {})  # name <- name (cfg substitution)
        self.overrides.update(self.scoped_overrides)
        for k in self.constants:
            if k not in self.const_vals and k not in self.overrides:
                raise TLAError(f"CONSTANT {k} has no value in the configuration")

    def _load(self, path):
        m = Module(path.read_text(errors="replace"))  # the reference's SSI spec has cp1252 quotes inside comments
        for e in m.extends:
            if e in self.BUILTIN_MODULES or any(x.name == e for x in self.modules):
                continue
            for d in self.search:
                if (d / f"{e}.tla").exists():
                    self._load(d / f"{e}.tla")
                    break
            else:
                raise TLAError(f"module {e} not found (search path: {[str(s) for s in self.search]})")
        self.modules.append(m)
        self.constants.update(m.constants)
        for v in m.variables:
            if v not in self.variables:
                self.variables.append(v)
        for name, (nm, params, body, line) in m.defs.items():
            if body[0] == "instance":
                self._import_instance(name + "!", body[1], body[2])
            else:
                self.defs[name] = GDef(nm, params, body, m.name, line)
        for inst in m.instances:  # INSTANCE M without a name (TLC/MCAlternatingBit.tla:4): M's definitions under their own names
            if inst not in self.BUILTIN_MODULES and not any(x.name == inst for x in self.modules):
                self._import_instance("", inst, [])

    def _find_module(self, name):
        for d in self.search:
            if (d / f"{name}.tla").exists():
                return d / f"{name}.tla"
        raise TLAError(f"module {name} not found (search path: {[str(s) for s in self.search]})")

    def _import_instance(self, prefix, modname, subst):
        """I == INSTANCE M WITH c <- e, ...: every definition d of M (and of what M EXTENDS) becomes the global definition
        I!d, in which M's own definition names are prefixed and the substituted constants / variables are replaced by their
        expressions; a constant or variable of M without a WITH clause stands for the instantiating module's identifier of the
        same name (Voting.tla:185 `C == INSTANCE Consensus`: Consensus's variable `chosen` IS Voting's state function
        `chosen`).  Priming such an identifier is priming a state function (v_prime)."""
        mods = []

        def gather(nm):
            m = Module(self._find_module(nm).read_text(errors="replace"))
            for e in m.extends:
                if e not in self.BUILTIN_MODULES:
                    gather(e)
            mods.append(m)
        gather(modname)
        names = set()
        for m in mods:
            names.update(m.defs)
        sub = dict(subst)

        def rn(x):
            if isinstance(x, tuple):
                if x and isinstance(x[0], str) and x[0] in ("id", "call", "nth") and len(x) > 1 and isinstance(x[1], str):
                    if x[0] == "id" and x[1] in sub:
                        return sub[x[1]]
                    if x[1].split("!")[0] in names:
                        return (x[0], prefix + x[1]) + tuple(rn(y) for y in x[2:])
                return tuple(rn(y) for y in x)
            if isinstance(x, list):
                return [rn(y) for y in x]
            return x
        for m in mods:
            for name, (nm, params, body, line) in m.defs.items():
                if body[0] == "instance":
                    self._import_instance(prefix + name + "!", body[1], [(a, rn(e)) for a, e in body[2]])
                    continue
                if not prefix and name in self.defs:
                    continue  # already there through EXTENDS
                self.defs[prefix + name] = GDef(prefix + nm, params, rn(body), m.name, line)
                if (m.name, name) in self.scoped:
                    self.scoped_overrides[prefix + name] = self.scoped[(m.name, name)]

    # ------------------------------------------------------------------ static analysis
    def primed(self, node, scope):
        """does evaluating `node` look at primed variables?  (decides whether a thunk's value may be memoised)"""
        k = node[0]
        if k in ("prime", "unchanged", "enabled"):
            return True
        if k in ("num", "str", "bool", "at"):
            return False
        if k == "nth":
            return self.primed(self._nth_node(node), {})
        if k == "id" or k == "call":
            name = node[1]
            r = False
            if name in scope:
                r = scope[name] is True
            elif name in self.defs and name not in self.overrides:
                d = self.defs[name]
                if d.primed is None:
                    d.primed = False  # recursion guard
                    d.primed = self.primed(d.body, {p: False for p, _ in d.params})
                r = d.primed
            if k == "call":
                r = r or any(self.primed(a, scope) for a in node[2])
            return r
        if k == "let":
            sc = dict(scope)
            any_p = False
            for name, params, body, _ in node[1]:
                sc2 = dict(sc)
                sc2.update({p: False for p, _ in params})
                sc[name] = False
                pr = self.primed(body, sc2)
                sc[name] = pr
                any_p = any_p or pr
            return self.primed(node[2], sc)  # a primed definition matters only where it is used
        if k in ("quant", "choose"):
            sc = dict(scope)
            r = False
            for pat, dom in node[2] if k == "quant" else node[1]:
                r = r or self.primed(dom, scope)
                for nm in (pat if isinstance(pat, tuple) else (pat,)):
                    sc[nm] = False
            return r or self.primed(node[3] if k == "quant" else node[2], sc)
        if k == "setfilter":
            sc = dict(scope)
            for nm in (node[1] if isinstance(node[1], tuple) else (node[1],)):
                sc[nm] = False
            return self.primed(node[2], scope) or self.primed(node[3], sc)
        if k == "setmap" or k == "fndef":
            bs = node[2] if k == "setmap" else node[1]
            body = node[1] if k == "setmap" else node[2]
            sc = dict(scope)
            r = False
            for pat, dom in bs:
                r = r or self.primed(dom, scope)
                for nm in (pat if isinstance(pat, tuple) else (pat,)):
                    sc[nm] = False
            return r or self.primed(body, sc)
        if k == "lambda":
            sc = dict(scope)
            sc.update({p: False for p in node[1]})
            return self.primed(node[2], sc)
        # generic: any child node
        for ch in node[1:]:
            if self._primed_any(ch, scope):
                return True
        return False

    def _primed_any(self, x, scope):
        if isinstance(x, tuple) and x and isinstance(x[0], str) and x[0] in NODE_KINDS:
            return self.primed(x, scope)
        if isinstance(x, (list, tuple)):
            return any(self._primed_any(y, scope) for y in x)
        return False

    # ------------------------------------------------------------------ compilation: VALUE mode
    # a compiled expression is f(env, st, nx) -> value;  env: dict of local names;  st: tuple of the current state's values;
    # nx: dict var index -> value of the (partial) next state, or None outside actions
    def cv(self, node, scope):
        m = getattr(self, "v_" + node[0], None)
        if m is None:
            raise TLAError(f"cannot evaluate a {node[0]!r} expression")
        return m(node, scope)

    def v_num(self, node, scope):
        v = node[1]
        return lambda env, st, nx: v

    v_str = v_num
    v_bool = v_num

    def v_paren(self, node, scope):
        return self.cv(node[1], scope)

    def v_at(self, node, scope):
        return lambda env, st, nx: env["@"]

    def _nth_node(self, node):
        name = self.overrides.get(node[1], node[1])
        if name not in self.defs:
            raise TLAError(f"{node[1]}!{node[2]}: {node[1]} is not a definition")
        body = self.defs[name].body
        while body[0] == "paren":
            body = body[1]
        if body[0] not in ("conj", "disj") or not 1 <= node[2] <= len(body[1]):
            raise TLAError(f"{node[1]}!{node[2]}: the definition has no such conjunct")
        return body[1][node[2] - 1]

    def v_nth(self, node, scope):
        return self.cv(self._nth_node(node), {})

    def v_temporal(self, node, scope):
        if node[1] in ("[]_", "<<>>_"):  # [A]_v = A \/ UNCHANGED v;  <<A>>_v = A /\ ~UNCHANGED v
            a, u = self.cv(node[2], scope), self.cv(("unchanged", node[3]), scope)
            if node[1] == "[]_":
                return lambda env, st, nx: _bool(a(env, st, nx), "[A]_v") or _bool(u(env, st, nx), "UNCHANGED")
            return lambda env, st, nx: _bool(a(env, st, nx), "<<A>>_v") and not _bool(u(env, st, nx), "UNCHANGED")

        def f(env, st, nx):
            raise TLAError("temporal formula evaluated")
        return f

    def a_temporal(self, node, scope):
        if node[1] == "[]_":
            a, u = self.ca(node[2], scope), self.ca(("unchanged", node[3]), scope)

            def g(env, st, nx):
                yield from a(env, st, nx)
                yield from u(env, st, nx)
            return g
        if node[1] == "<<>>_":
            a, u = self.ca(node[2], scope), self.cv(("unchanged", node[3]), scope)

            def g2(env, st, nx):
                for nx2 in a(env, st, nx):
                    if u(env, st, nx2) is False:
                        yield nx2
            return g2
        return self._a_test(node, scope)

    def v_id(self, node, scope):
        name = node[1]
        if name in scope:
            def f(env, st, nx):
                v = env[name]
                if type(v) is Thunk:
                    return v.force(nx)
                if type(v) is OpVal and not v.params:
                    return v.cv(v.env, st, nx)
                return v
            return f
        if name in self.overrides:
            return self.v_id(("id", self.overrides[name]), scope)
        if name in self.varidx:
            i = self.varidx[name]

            def fvar(env, st, nx):
                v = st[i]
                if v is UNASSIGNED and nx:  # inside Init a later conjunct reads what an earlier one assigned (MCConsensus.tla:19-20)
                    v = nx.get(i, UNASSIGNED)
                    if v is UNASSIGNED:
                        raise TLAError(f"{name} is read before the initial predicate gives it a value")
                return v
            return fvar
        if name in self.const_vals:
            v = self.const_vals[name]
            return lambda env, st, nx: v
        if name in self.defs:
            return self._global_apply(self.defs[name], [], scope)
        b = BUILTIN_CONSTS.get(name)
        if b is not None:
            return lambda env, st, nx: b
        raise TLAError(f"unknown identifier {name}")

    def _compile_def(self, d):
        if d.cv is None:
            sc = {p: False for p, _ in d.params}
            d.cv = "pending"
            cvf = self.cv(d.body, sc)
            d.cv = cvf
        return d

    def _global_apply(self, d, argnodes, scope):
        if len(argnodes) != len(d.params):
            raise TLAError(f"operator {d.name} takes {len(d.params)} arguments, {len(argnodes)} given")
        args = [self._arg(a, scope, arity) for a, (_, arity) in zip(argnodes, d.params)]
        pnames = [p for p, _ in d.params]
        spec = self
        if not pnames:
            # constant-level zero-argument definitions are evaluated once
            state = {}

            def f0(env, st, nx):
                if "v" in state:
                    return state["v"]
                spec._compile_def(d)
                if d.is_const is None:
                    d.is_const = not spec._mentions_state(d.body, set())
                v = d.cv({}, st, nx)
                if d.is_const:
                    state["v"] = v
                return v
            return f0

        def f(env, st, nx):
            if d.cv is None or d.cv == "pending":
                spec._compile_def(d)
            return d.cv({p: a(env, st, nx) for p, a in zip(pnames, args)}, st, nx)
        return f

    def _mentions_state(self, node, seen):
        """does the expression (transitively) mention a VARIABLE?"""
        if isinstance(node, tuple) and node and isinstance(node[0], str) and node[0] in NODE_KINDS:
            if node[0] == "nth":
                return self._mentions_state(self._nth_node(node), seen)
            if node[0] in ("id", "call"):
                nm = self.overrides.get(node[1], node[1])
                if nm in self.varidx:
                    return True
                if nm in self.defs and nm not in seen:
                    seen.add(nm)
                    if self._mentions_state(self.defs[nm].body, seen):
                        return True
            if node[0] in ("prime", "unchanged", "enabled"):
                return True
            return any(self._mentions_state(ch, seen) for ch in node[1:])
        if isinstance(node, (list, tuple)):
            return any(self._mentions_state(y, seen) for y in node)
        return False

    def _arg(self, node, scope, arity=0):
        """an argument as a function (env, st, nx) -> Thunk or OpVal"""
        if node[0] == "lambda":
            params = node[1]
            sc = dict(scope)
            sc.update({p: False for p in params})
            cvf = self.cv(node[2], sc)
            return lambda env, st, nx: OpVal(params, cvf, None, env)
        if arity > 0 and node[0] == "id":  # an operator passed by name
            name = node[1]
            if name in scope:
                return lambda env, st, nx: env[name]
            if name in self.defs:
                d = self.defs[name]
                pn = [p for p, _ in d.params]
                spec = self

                def mk(env, st, nx):
                    spec._compile_def(d)
                    return OpVal(pn, d.cv, None, {}, name)
                return mk
            bi = BUILTIN_OPS.get(name)
            if bi is not None:
                def mkb(env, st, nx):
                    pn = [f"_{i}" for i in range(arity)]
                    return OpVal(pn, lambda e, s, n: bi(*[e[p] if type(e[p]) is not Thunk else e[p].force(n) for p in pn]), None, {}, name)
                return mkb
        f = self.cv(node, scope)
        memo = not self.primed(node, scope)
        # an argument that IS a primed variable (InternalMemory.tla:17 Send(p, req, memInt, memInt')) can be assigned through the
        # parameter (MCSend(p, d, old, new) == new = <<p, d>>): the thunk remembers which variable
        inner = node
        while inner[0] == "paren":
            inner = inner[1]
        pvar = None
        if inner[0] == "prime" and inner[1][0] == "id" and inner[1][1] not in scope and self.overrides.get(inner[1][1], inner[1][1]) in self.varidx:
            pvar = self.varidx[self.overrides.get(inner[1][1], inner[1][1])]
        if inner[0] == "id" and inner[1] in scope:
            name = inner[1]

            def mkp(env, st, nx):
                t = Thunk(f, env, st, memo)
                src = env[name]
                t.pvar = src.pvar if type(src) is Thunk else None
                return t
            return mkp

        def mkt(env, st, nx):
            t = Thunk(f, env, st, memo)
            t.pvar = pvar
            return t
        return mkt

    def v_call(self, node, scope):
        name, argnodes = node[1], node[2]
        if name in scope:  # operator parameter / LET operator
            args = [self._arg(a, scope) for a in argnodes]

            def f(env, st, nx):
                op = env[name]
                if type(op) is not OpVal:
                    raise TLAError(f"{name} is not an operator")
                e2 = dict(op.env)
                for p, a in zip(op.params, args):
                    e2[p] = a(env, st, nx)
                return op.cv(e2, st, nx)
            return f
        if name in self.overrides:
            return self.v_call(("call", self.overrides[name], argnodes), scope)
        if name in self.defs:
            return self._global_apply(self.defs[name], argnodes, scope)
        if name in self.const_vals and isinstance(self.const_vals[name], OpVal):
            op = self.const_vals[name]
            args = [self._arg(a, scope) for a in argnodes]
            return lambda env, st, nx: op.cv({p: a(env, st, nx) for p, a in zip(op.params, args)}, st, nx)
        bi = BUILTIN_OPS.get(name)
        if name in BUILTIN_OPS:
            if name == "SelectSeq":
                seq = self.cv(argnodes[0], scope)
                test = self._arg(argnodes[1], scope, 1)

                def fsel(env, st, nx):
                    op = test(env, st, nx)
                    out = []
                    for x in seq(env, st, nx):
                        e2 = dict(op.env)
                        e2[op.params[0]] = x
                        if op.cv(e2, st, nx) is True:
                            out.append(x)
                    return tuple(out)
                return fsel
            if name == "Assert":
                cond, msg = self.cv(argnodes[0], scope), self.cv(argnodes[1], scope)

                def fas(env, st, nx):
                    if cond(env, st, nx) is not True:
                        raise AssertFail(str(msg(env, st, nx)))
                    return True
                return fas
            args = [self.cv(a, scope) for a in argnodes]
            if len(args) == 1:
                a0 = args[0]
                return lambda env, st, nx: bi(a0(env, st, nx))
            if len(args) == 2:
                a0, a1 = args
                return lambda env, st, nx: bi(a0(env, st, nx), a1(env, st, nx))
            return lambda env, st, nx: bi(*[a(env, st, nx) for a in args])
        raise TLAError(f"unknown operator {name}")

    def v_prime(self, node, scope):
        inner = node[1]
        if inner[0] == "id" and inner[1] not in scope and self.overrides.get(inner[1], inner[1]) in self.varidx:
            i = self.varidx[self.overrides.get(inner[1], inner[1])]
            name = inner[1]

            def f(env, st, nx):
                if nx is None:
                    raise TLAError(f"{name}' evaluated outside an action")
                v = nx.get(i, UNASSIGNED)
                if v is UNASSIGNED:
                    raise TLAError(f"{name}' is read before it is assigned")
                return v
            return f
        # e' for a state function e: evaluate e in the next state (all its variables must be assigned)
        f0 = self.cv(inner, scope)
        nvars = len(self.variables)

        def g(env, st, nx):
            if nx is None:
                raise TLAError("priming outside an action")
            st2 = tuple(nx.get(i, UNASSIGNED) for i in range(nvars))
            return f0(env, st2, None)
        return g

    def v_not(self, node, scope):
        f = self.cv(node[1], scope)

        def g(env, st, nx):
            v = f(env, st, nx)
            if v is True:
                return False
            if v is False:
                return True
            raise TLAError(f"~ applied to the non-boolean {fmt(v)}")
        return g

    def v_neg(self, node, scope):
        f = self.cv(node[1], scope)
        return lambda env, st, nx: -f(env, st, nx)

    def v_conj(self, node, scope):
        fs = [self.cv(x, scope) for x in node[1]]

        def g(env, st, nx):
            for f in fs:
                v = f(env, st, nx)
                if v is False:
                    return False
                if v is not True:
                    raise TLAError(f"/\\ applied to the non-boolean {fmt(v)}")
            return True
        return g

    def v_disj(self, node, scope):
        fs = [self.cv(x, scope) for x in node[1]]

        def g(env, st, nx):
            for f in fs:
                v = f(env, st, nx)
                if v is True:
                    return True
                if v is not False:
                    raise TLAError(f"\\/ applied to the non-boolean {fmt(v)}")
            return False
        return g

    def v_if(self, node, scope):
        c, a, b = self.cv(node[1], scope), self.cv(node[2], scope), self.cv(node[3], scope)

        def g(env, st, nx):
            v = c(env, st, nx)
            if v is True:
                return a(env, st, nx)
            if v is False:
                return b(env, st, nx)
            raise TLAError(f"IF condition is the non-boolean {fmt(v)}")
        return g

    def v_case(self, node, scope):
        arms = [(self.cv(g, scope), self.cv(e, scope)) for g, e in node[1]]
        other = self.cv(node[2], scope) if node[2] is not None else None

        def f(env, st, nx):
            for g, e in arms:
                if g(env, st, nx) is True:
                    return e(env, st, nx)
            if other is None:
                raise TLAError("CASE: no arm is true and there is no OTHER")
            return other(env, st, nx)
        return f

    def _bind(self, pat, val, env):
        if isinstance(pat, tuple):
            if not isinstance(val, tuple) or len(val) != len(pat):
                raise TLAError(f"tuple pattern {pat} bound to {fmt(val)}")
            for p, v in zip(pat, val):
                env[p] = v
        else:
            env[pat] = val

    def _scope_with(self, scope, bs):
        sc = dict(scope)
        for pat, _ in bs:
            for nm in (pat if isinstance(pat, tuple) else (pat,)):
                sc[nm] = False
        return sc

    def _iter_bounds(self, bs, scope):
        """compiled bounds -> generator of environments"""
        # each bound's set may mention earlier bound names only in setmap/fndef/quant per TLA+? (it may not); evaluate in outer env
        doms = [(pat, self.cv(dom, scope)) for pat, dom in bs]
        bind = self._bind
        if len(doms) == 1:
            pat, dom = doms[0]
            if not isinstance(pat, tuple):
                def it1(env, st, nx):
                    for v in iter_set(dom(env, st, nx)):
                        e2 = dict(env)
                        e2[pat] = v
                        yield e2
                return it1

        def it(env, st, nx):
            sets = [list(iter_set(dom(env, st, nx))) for _, dom in doms]
            for combo in itertools.product(*sets):
                e2 = dict(env)
                for (pat, _), v in zip(doms, combo):
                    bind(pat, v, e2)
                yield e2
        return it

    def v_quant(self, node, scope):
        kind, bs, body = node[1], node[2], node[3]
        it = self._iter_bounds(bs, scope)
        f = self.cv(body, self._scope_with(scope, bs))
        if kind == "E":
            def g(env, st, nx):
                for e2 in it(env, st, nx):
                    v = f(e2, st, nx)
                    if v is True:
                        return True
                    if v is not False:
                        raise TLAError(f"\\E body is the non-boolean {fmt(v)}")
                return False
        else:
            def g(env, st, nx):
                for e2 in it(env, st, nx):
                    v = f(e2, st, nx)
                    if v is False:
                        return False
                    if v is not True:
                        raise TLAError(f"\\A body is the non-boolean {fmt(v)}")
                return True
        return g

    def v_choose_unbounded(self, node, scope):
        def f(env, st, nx):
            raise TLAError(f"TLC cannot evaluate the unbounded CHOOSE {node[1]} : ... (give the defined symbol a model value in the cfg)")
        return f

    def v_choose(self, node, scope):
        bs, body = node[1], node[2]
        if len(bs) != 1:
            raise TLAError("CHOOSE with several bounds")
        pat, domn = bs[0]
        dom = self.cv(domn, scope)
        f = self.cv(body, self._scope_with(scope, bs))
        bind = self._bind

        def g(env, st, nx):
            for v in sorted_vals(iter_set(dom(env, st, nx))):
                e2 = dict(env)
                bind(pat, v, e2)
                if f(e2, st, nx) is True:
                    return v
            raise TLAError("CHOOSE: no element satisfies the predicate")
        return g

    def v_setenum(self, node, scope):
        fs = [self.cv(x, scope) for x in node[1]]
        return lambda env, st, nx: frozenset(f(env, st, nx) for f in fs)

    def v_tuple(self, node, scope):
        fs = [self.cv(x, scope) for x in node[1]]
        return lambda env, st, nx: tuple(f(env, st, nx) for f in fs)

    def v_setfilter(self, node, scope):
        pat, dom, pred = node[1], self.cv(node[2], scope), None
        pred = self.cv(node[3], self._scope_with(scope, [(pat, None)]))
        bind = self._bind

        def g(env, st, nx):
            out = []
            for v in iter_set(dom(env, st, nx)):
                e2 = dict(env)
                bind(pat, v, e2)
                r = pred(e2, st, nx)
                if r is True:
                    out.append(v)
                elif r is not False:
                    raise TLAError(f"set filter predicate is the non-boolean {fmt(r)}")
            return frozenset(out)
        return g

    def v_setmap(self, node, scope):
        it = self._iter_bounds(node[2], scope)
        f = self.cv(node[1], self._scope_with(scope, node[2]))
        return lambda env, st, nx: frozenset(f(e2, st, nx) for e2 in it(env, st, nx))

    def v_fndef(self, node, scope):
        bs, bodyn, selfname = node[1], node[2], node[3]
        sc = self._scope_with(scope, bs)
        if selfname:
            sc[selfname] = False
        body = self.cv(bodyn, sc)
        doms = [(pat, self.cv(dom, scope)) for pat, dom in bs]
        bind = self._bind
        if selfname:
            # f[x \in S] == e with f inside e (WriteThroughCache.tla:55-60): while the function is built, f[a] evaluates e for x = a
            if len(doms) != 1:
                raise TLAError("recursive function definitions with several bounds are not supported")
            pat0, dom0 = doms[0]

            def grec(env, st, nx):
                d = dom0(env, st, nx)
                rec = RecFn(d)
                e1 = dict(env)
                e1[selfname] = rec

                def apply(a):
                    if a in rec.cache:
                        return rec.cache[a]
                    if not set_in(a, d):
                        raise TLAError(f"function applied outside its domain: {fmt(a)}")
                    e2 = dict(e1)
                    bind(pat0, a, e2)
                    v = body(e2, st, nx)
                    rec.cache[a] = v
                    return v
                rec.apply = apply
                return mk_fn({x: apply(x) for x in iter_set(d)})
            return grec
        if len(doms) == 1 and not isinstance(doms[0][0], tuple):
            pat, dom = doms[0]

            def g1(env, st, nx):
                d = {}
                for v in iter_set(dom(env, st, nx)):
                    e2 = dict(env)
                    e2[pat] = v
                    d[v] = body(e2, st, nx)
                return mk_fn(d)
            return g1

        def g(env, st, nx):
            sets = [list(iter_set(dom(env, st, nx))) for _, dom in doms]
            d = {}
            for combo in itertools.product(*sets):
                e2 = dict(env)
                for (pat, _), v in zip(doms, combo):
                    bind(pat, v, e2)
                d[combo if len(combo) > 1 else combo[0]] = body(e2, st, nx)
            return mk_fn(d)
        return g

    def v_record(self, node, scope):
        fs = [(k, self.cv(e, scope)) for k, e in node[1]]
        return lambda env, st, nx: Fn({k: f(env, st, nx) for k, f in fs})

    def v_recordset(self, node, scope):
        fs = [(k, self.cv(e, scope)) for k, e in node[1]]
        return lambda env, st, nx: RecSet([(k, f(env, st, nx)) for k, f in fs])

    def v_fnset(self, node, scope):
        a, b = self.cv(node[1], scope), self.cv(node[2], scope)
        return lambda env, st, nx: FnSet(a(env, st, nx), b(env, st, nx))

    def v_except(self, node, scope):
        base = self.cv(node[1], scope)
        sc = dict(scope)
        sc["@"] = False
        ups = [([self.cv(p, scope) for p in path], self.cv(e, sc)) for path, e in node[2]]

        def upd(f, path, k, efn, env, st, nx):
            key = path[k](env, st, nx)
            old = fn_apply(f, key)
            if k + 1 == len(path):
                e2 = dict(env)
                e2["@"] = old
                new = efn(e2, st, nx)
            else:
                new = upd(old, path, k + 1, efn, env, st, nx)
            if isinstance(f, tuple):
                return f[:key - 1] + (new,) + f[key:]
            d = dict(f.d)
            d[key] = new
            return Fn(d)

        def g(env, st, nx):
            f = base(env, st, nx)
            for path, efn in ups:
                f = upd(f, path, 0, efn, env, st, nx)
            return f
        return g

    def v_idx(self, node, scope):
        f, a = self.cv(node[1], scope), self.cv(node[2], scope)

        def g(env, st, nx):
            fv = f(env, st, nx)
            if type(fv) is RecFn:
                return fv.apply(a(env, st, nx))
            return fn_apply(fv, a(env, st, nx))
        return g

    def v_pre(self, node, scope):
        f = self.cv(node[2], scope)
        if node[1] == "DOMAIN":
            return lambda env, st, nx: fn_domain(f(env, st, nx))
        if node[1] == "SUBSET":
            return lambda env, st, nx: Powerset(f(env, st, nx))

        def un(env, st, nx):
            out = set()
            for s in iter_set(f(env, st, nx)):
                out.update(iter_set(s))
            return frozenset(out)
        return un

    def v_let(self, node, scope):
        binder, sc = self._let_binder(node[1], scope)
        body = self.cv(node[2], sc)
        return lambda env, st, nx: body(binder(env, st, nx), st, nx)

    def _let_binder(self, defs, scope):
        sc = dict(scope)
        comp = []
        for name, params, bodyn, _ in defs:
            sc[name] = False  # visible to itself (RECURSIVE) and to later definitions
        for name, params, bodyn, _ in defs:
            sc2 = dict(sc)
            sc2.update({p: False for p, _ in params})
            pr = self.primed(bodyn, sc2)
            sc[name] = pr
            cvf = self.cv(bodyn, sc2)
            try:
                caf = self.ca(bodyn, sc2)
            except TLAError:
                caf = None
            comp.append((name, [p for p, _ in params], cvf, caf, not pr))

        def binder(env, st, nx):
            e2 = dict(env)
            for name, params, cvf, caf, memo in comp:
                if params:
                    e2[name] = OpVal(params, cvf, caf, e2, name)
                else:
                    t = Thunk(cvf, e2, st, memo)
                    e2[name] = t if caf is None or memo else OpVal([], cvf, caf, e2, name)
            return e2
        return binder, sc

    def v_lambda(self, node, scope):
        raise TLAError("LAMBDA outside an argument position")

    def v_unchanged(self, node, scope):
        try:
            idxs = self._unchanged_vars(node[1], scope)
        except TLAError:  # UNCHANGED e for a state function e (an instantiated module's variable, Paxos.tla:201): e' = e
            return self.cv(("op", "=", ("prime", node[1]), node[1]), scope)

        def g(env, st, nx):
            if nx is None:
                raise TLAError("UNCHANGED outside an action")
            for i in idxs:
                v = nx.get(i, UNASSIGNED)
                if v is UNASSIGNED:
                    raise TLAError(f"UNCHANGED {self.variables[i]} read before it is assigned")
                if v != st[i]:
                    return False
            return True
        return g

    def _unchanged_vars(self, node, scope):
        """flatten UNCHANGED <<a, b, vars>> to variable indices (definitions that are tuples of variables are expanded)"""
        k = node[0]
        if k == "paren":
            return self._unchanged_vars(node[1], scope)
        if k == "tuple":
            out = []
            for x in node[1]:
                out.extend(self._unchanged_vars(x, scope))
            return out
        if k == "id":
            name = self.overrides.get(node[1], node[1])
            if name in self.varidx and name not in scope:
                return [self.varidx[name]]
            if name in self.defs and name not in scope:
                return self._unchanged_vars(self.defs[name].body, {})
        raise TLAError(f"UNCHANGED of something that is not a tuple of variables: {node}")

    def v_op(self, node, scope):
        op = node[1]
        a, b = self.cv(node[2], scope), self.cv(node[3], scope)
        if op == "=":
            return lambda env, st, nx: a(env, st, nx) == b(env, st, nx)
        if op == "#":
            return lambda env, st, nx: a(env, st, nx) != b(env, st, nx)
        if op == "\\in":
            return lambda env, st, nx: set_in(a(env, st, nx), b(env, st, nx))
        if op == "\\notin":
            return lambda env, st, nx: not set_in(a(env, st, nx), b(env, st, nx))
        if op == "=>":
            def imp(env, st, nx):
                v = a(env, st, nx)
                if v is False:
                    return True
                return b(env, st, nx)
            return imp
        if op == "<=>":
            return lambda env, st, nx: a(env, st, nx) is b(env, st, nx)
        fn = BINOPS.get(op)
        if fn is None:
            raise TLAError(f"operator {op} is not supported")
        return lambda env, st, nx: fn(a(env, st, nx), b(env, st, nx))

    # ------------------------------------------------------------------ compilation: ACTION mode
    # a compiled action is g(env, st, nx) -> iterator of next-state dicts (nx extended, never mutated)
    def ca(self, node, scope):
        m = getattr(self, "a_" + node[0], None)
        if m is not None:
            return m(node, scope)
        return self._a_test(node, scope)

    def _a_test(self, node, scope):
        f = self.cv(node, scope)

        def g(env, st, nx):
            v = f(env, st, nx)
            if v is True:
                yield nx
            elif v is not False:
                raise TLAError(f"action conjunct is the non-boolean {fmt(v)}")
        return g

    def a_paren(self, node, scope):
        return self.ca(node[1], scope)

    def a_conj(self, node, scope):
        parts = [self.ca(x, scope) for x in node[1]]
        n = len(parts)

        def run(env, st, nx, i):
            if i == n - 1:
                yield from parts[i](env, st, nx)
                return
            for nx2 in parts[i](env, st, nx):
                yield from run(env, st, nx2, i + 1)
        return lambda env, st, nx: run(env, st, nx, 0)

    def a_disj(self, node, scope):
        parts = [self.ca(x, scope) for x in node[1]]

        def g(env, st, nx):
            for p in parts:
                yield from p(env, st, nx)
        return g

    def a_if(self, node, scope):
        c, a, b = self.cv(node[1], scope), self.ca(node[2], scope), self.ca(node[3], scope)

        def g(env, st, nx):
            v = c(env, st, nx)
            if v is True:
                return a(env, st, nx)
            if v is False:
                return b(env, st, nx)
            raise TLAError(f"IF condition is the non-boolean {fmt(v)}")
        return g

    def a_case(self, node, scope):
        arms = [(self.cv(g, scope), self.ca(e, scope)) for g, e in node[1]]
        other = self.ca(node[2], scope) if node[2] is not None else None

        def f(env, st, nx):
            for g, e in arms:
                if g(env, st, nx) is True:
                    return e(env, st, nx)
            if other is None:
                raise TLAError("CASE: no arm is true and there is no OTHER")
            return other(env, st, nx)
        return f

    def a_quant(self, node, scope):
        kind, bs, body = node[1], node[2], node[3]
        if kind != "E":
            return self._a_test(node, scope)
        it = self._iter_bounds(bs, scope)
        f = self.ca(body, self._scope_with(scope, bs))

        def g(env, st, nx):
            for e2 in it(env, st, nx):
                yield from f(e2, st, nx)
        return g

    def a_let(self, node, scope):
        binder, sc = self._let_binder(node[1], scope)
        body = self.ca(node[2], sc)
        return lambda env, st, nx: body(binder(env, st, nx), st, nx)

    def a_unchanged(self, node, scope):
        try:
            idxs = self._unchanged_vars(node[1], scope)
        except TLAError:
            return self._a_test(node, scope)
        clash_test = self.clash == "test"

        def g(env, st, nx):
            new = None
            for i in idxs:
                v = nx.get(i, UNASSIGNED)
                if v is UNASSIGNED:
                    if new is None:
                        new = dict(nx)
                    new[i] = st[i]
                elif clash_test and v != st[i]:
                    return
            yield nx if new is None else new
        return g

    def a_op(self, node, scope):
        op = node[1]
        lhs = node[2]
        while lhs[0] == "paren":
            lhs = lhs[1]
        if op in ("=", "\\in") and lhs[0] == "prime" and lhs[1][0] == "id" and lhs[1][1] not in scope \
                and self.overrides.get(lhs[1][1], lhs[1][1]) in self.varidx:
            i = self.varidx[self.overrides.get(lhs[1][1], lhs[1][1])]
            rhs = self.cv(node[3], scope)
            clash_test = self.clash == "test"
            if op == "=":
                def g(env, st, nx):
                    cur = nx.get(i, UNASSIGNED)
                    if cur is UNASSIGNED:
                        new = dict(nx)
                        new[i] = rhs(env, st, nx)
                        yield new
                    elif not clash_test or cur == rhs(env, st, nx):
                        yield nx
                return g

            def gin(env, st, nx):
                cur = nx.get(i, UNASSIGNED)
                s = rhs(env, st, nx)
                if cur is UNASSIGNED:
                    for v in iter_set(s):
                        new = dict(nx)
                        new[i] = v
                        yield new
                elif not clash_test or set_in(cur, s):
                    yield nx
            return gin
        if op in ("=", "\\in") and lhs[0] == "id" and lhs[1] in scope:
            pname = lhs[1]
            rhs = self.cv(node[3], scope)
            test = self._a_test(node, scope)

            def gpar(env, st, nx):
                t = env[pname]
                i = t.pvar if type(t) is Thunk else None
                if i is None:
                    yield from test(env, st, nx)
                    return
                cur = nx.get(i, UNASSIGNED)
                r = rhs(env, st, nx)
                if cur is UNASSIGNED:
                    for v in ([r] if op == "=" else iter_set(r)):
                        new = dict(nx)
                        new[i] = v
                        yield new
                elif (cur == r) if op == "=" else set_in(cur, r):
                    yield nx
            return gpar
        return self._a_test(node, scope)

    def a_id(self, node, scope):
        return self.a_call(("call", node[1], []), scope, from_id=node)

    def a_call(self, node, scope, from_id=None):
        name, argnodes = node[1], node[2]
        name = self.overrides.get(name, name) if name not in scope else name
        if name in scope:
            args = [self._arg(a, scope) for a in argnodes]
            fallback = self.cv(from_id or node, scope)

            def g(env, st, nx):
                op = env[name]
                if type(op) is OpVal and op.ca is not None:
                    e2 = dict(op.env)
                    for p, a in zip(op.params, args):
                        e2[p] = a(env, st, nx)
                    yield from op.ca(e2, st, nx)
                    return
                v = fallback(env, st, nx)
                if v is True:
                    yield nx
                elif v is not False:
                    raise TLAError(f"action conjunct {name} is the non-boolean {fmt(v)}")
            return g
        if name in self.defs:
            d = self.defs[name]
            if len(argnodes) != len(d.params):
                raise TLAError(f"operator {d.name} takes {len(d.params)} arguments")
            args = [self._arg(a, scope, arity) for a, (_, arity) in zip(argnodes, d.params)]
            pnames = [p for p, _ in d.params]
            spec = self

            def g(env, st, nx):
                if d.ca is None:
                    d.ca = "pending"
                    d.ca = spec.ca(d.body, {p: False for p in pnames})
                return d.ca({p: a(env, st, nx) for p, a in zip(pnames, args)}, st, nx)
            return g
        return self._a_test(from_id or node, scope)

    # ------------------------------------------------------------------ entry points
    def value_of(self, expr_text_or_name, state=None):
        f = self.compile_value(expr_text_or_name)
        return f({}, state, None)

    def compile_value(self, name):
        node = ("id", name) if isinstance(name, str) else name
        return self.cv(node, {})

    def compile_action(self, name):
        node = ("id", name) if isinstance(name, str) else name
        return self.ca(node, {})

    def init_states(self, init="Init"):
        g = self.compile_action_as_init(init)
        nv = len(self.variables)
        out = []
        for nx in g({}, tuple([UNASSIGNED] * nv), {}):
            if len(nx) != nv:
                missing = [v for v, i in self.varidx.items() if i not in nx]
                raise TLAError(f"initial state leaves {missing} unassigned")
            out.append(tuple(nx[i] for i in range(nv)))
        return out

    def compile_action_as_init(self, init):
        """Init is evaluated like an action in which every UNPRIMED variable is assigned: reuse the action compiler on a copy of
        the tree where `x = e` / `x \\in S` for a variable x stands for x' = e / x' \\in S"""
        def rw(node):
            if isinstance(node, tuple) and node and node[0] == "op" and node[1] in ("=", "\\in"):
                l = node[2]
                while l[0] == "paren":
                    l = l[1]
                if l[0] == "id" and self.overrides.get(l[1], l[1]) in self.varidx:
                    return ("op", node[1], ("prime", l), node[3])
                return node
            if isinstance(node, tuple) and node and node[0] in ("conj", "disj"):
                return (node[0], [rw(x) for x in node[1]])
            if isinstance(node, tuple) and node and node[0] == "paren":
                return ("paren", rw(node[1]))
            if isinstance(node, tuple) and node and node[0] == "quant" and node[1] == "E":
                return ("quant", "E", node[2], rw(node[3]))
            if isinstance(node, tuple) and node and node[0] == "id" and node[1] in self.defs and not self.defs[node[1]].params:
                return rw(self.defs[node[1]].body)
            return node
        return self.ca(rw(("id", init)), {})

    def successors(self, st, nextf):
        nv = len(self.variables)
        for nx in nextf({}, st, {}):
            if len(nx) != nv:
                missing = [v for v, i in self.varidx.items() if i not in nx]
                raise TLAError(f"a successor leaves {missing} unassigned")
            yield tuple(nx[i] for i in range(nv))

    def state_text(self, st, order=None, sep=" "):
        names = order or self.variables
        return sep.join(f"/\\ {v} = {fmt(st[self.varidx[v]])}" for v in names)


NODE_KINDS = {"num", "str", "bool", "id", "call", "paren", "at", "conj", "disj", "op", "not", "neg", "quant", "choose", "choose_unbounded", "if", "case",
              "let", "lambda", "unchanged", "enabled", "pre", "setenum", "setfilter", "setmap", "tuple", "record", "recordset",
              "fndef", "fnset", "except", "idx", "prime", "temporal", "instance", "nth"}


# ---- built-in operators
def _int(v, what):
    if type(v) is not int:
        raise TLAError(f"{what} applied to the non-integer {fmt(v)}")
    return v


def _seq(v, what):
    if not isinstance(v, tuple):
        raise TLAError(f"{what} applied to the non-sequence {fmt(v)}")
    return v


def _bool(v, what):
    if v is not True and v is not False:
        raise TLAError(f"{what} applied to the non-boolean {fmt(v)}")
    return v


def op_div(a, b):
    _int(a, "\\div"), _int(b, "\\div")
    if b == 0:
        raise TLAError("division by zero")
    return a // b


def op_mod(a, b):
    _int(a, "%"), _int(b, "%")
    if b <= 0:
        raise TLAError("% with a non-positive modulus")
    return a % b


def op_range(a, b):
    return frozenset(range(_int(a, ".."), _int(b, "..") + 1))


def op_atat(f, g):  # f @@ g: union of the domains, f wins (TLC.tla:11-12)
    d = fn_items(g)
    d.update(fn_items(f))
    return mk_fn(d)


def op_cup(a, b):
    if isinstance(a, frozenset) and isinstance(b, frozenset):
        return a | b
    return to_frozen(a) | to_frozen(b)


def op_setminus(a, b):
    if isinstance(b, frozenset):
        return to_frozen(a) - b
    return frozenset(x for x in iter_set(a) if not set_in(x, b))


def op_subseteq(a, b):
    if isinstance(a, frozenset) and isinstance(b, frozenset):
        return a <= b
    return all(set_in(x, b) for x in iter_set(a))


def op_lt(a, b):
    return _int(a, "<") < _int(b, "<")


BINOPS = {
    "+": lambda a, b: _int(a, "+") + _int(b, "+"), "-": lambda a, b: _int(a, "-") - _int(b, "-"),
    "*": lambda a, b: _int(a, "*") * _int(b, "*"), "\\div": op_div, "%": op_mod, "^": lambda a, b: _int(a, "^") ** _int(b, "^"),
    "<": op_lt, ">": lambda a, b: _int(a, ">") > _int(b, ">"), "<=": lambda a, b: _int(a, "<=") <= _int(b, "<="),
    ">=": lambda a, b: _int(a, ">=") >= _int(b, ">="), "..": op_range,
    "\\cup": op_cup, "\\cap": lambda a, b: frozenset(x for x in iter_set(a) if set_in(x, b)), "\\": op_setminus,
    "\\subseteq": op_subseteq, "\\o": lambda a, b: _seq(a, "\\o") + _seq(b, "\\o"),
    ":>": lambda a, b: mk_fn({a: b}), "@@": op_atat,
    "\\X": lambda a, b: frozenset((x, y) for x in iter_set(a) for y in iter_set(b)),
}


def bi_subseq(s, m, n):
    _seq(s, "SubSeq")
    if m > n:
        return ()
    if m < 1 or n > len(s):
        raise TLAError(f"SubSeq({fmt(s)}, {m}, {n}) out of range")
    return s[m - 1:n]


def bi_head(s):
    if not _seq(s, "Head"):
        raise TLAError("Head of the empty sequence")
    return s[0]


def bi_tail(s):
    if not _seq(s, "Tail"):
        raise TLAError("Tail of the empty sequence")
    return s[1:]


def bi_permutations(s):
    el = sorted_vals(iter_set(s))
    return frozenset(mk_fn(dict(zip(el, p))) for p in itertools.permutations(el))


def bi_print(v, r=None):
    return True if r is None else r


BUILTIN_OPS = {
    "Cardinality": lambda s: len(to_frozen(s)), "IsFiniteSet": lambda s: isinstance(s, frozenset),
    "Len": lambda s: len(_seq(s, "Len")), "Append": lambda s, e: _seq(s, "Append") + (e,), "Head": bi_head, "Tail": bi_tail,
    "SubSeq": bi_subseq, "Seq": lambda s: SeqSet(s), "SelectSeq": None, "Assert": None, "Permutations": bi_permutations,
    "Print": bi_print, "PrintT": lambda v: True, "ToString": lambda v: fmt(v),
}
BUILTIN_CONSTS = {"Nat": NatSet(0), "Int": NatSet(None), "BOOLEAN": frozenset([True, False]), "STRING": StringSet()}


# =============================================================================================== cfg
def permute(v, g):
    """the value with every model value m replaced by g[m] (TLC's symmetry reduction applies a permutation to a whole state)"""
    if isinstance(v, MV):
        return g.get(v, v)
    if isinstance(v, tuple):
        return tuple(permute(x, g) for x in v)
    if isinstance(v, frozenset):
        return frozenset(permute(x, g) for x in v)
    if isinstance(v, Fn):
        return mk_fn({permute(k, g): permute(x, g) for k, x in v.d.items()})
    return v


def parse_cfg(text):
    """TLC configuration file (grammar: examples/SpecifyingSystems/TLC/ConfigFileGrammar.tla:4-32) ->
    dict(spec, init, next, invariants, constraints, constants {name: value}, overrides {name: name}, symmetry)"""
    toks = [t for t in lex(text) if t.k != "end"]
    out = dict(spec=None, init=None, next=None, invariants=[], constraints=[], constants={}, overrides={}, symmetry=None,
               properties=[], scoped={})
    KW = {"SPECIFICATION", "INIT", "NEXT", "INVARIANT", "INVARIANTS", "CONSTRAINT", "CONSTRAINTS", "CONSTANT", "CONSTANTS", "SYMMETRY",
          "PROPERTY", "PROPERTIES", "ACTION_CONSTRAINT", "ACTION_CONSTRAINTS", "VIEW", "CHECK_DEADLOCK"}
    i = 0

    def value(j):
        t = toks[j]
        if t.k == "num":
            return int(t.s), j + 1
        if t.k == "str":
            return t.s, j + 1
        if t.k == "sym" and t.s == "-" and toks[j + 1].k == "num":
            return -int(toks[j + 1].s), j + 2
        if t.k == "sym" and t.s == "{":
            j += 1
            items = []
            while not (toks[j].k == "sym" and toks[j].s == "}"):
                if toks[j].k == "sym" and toks[j].s == ",":
                    j += 1
                    continue
                v, j = value(j)
                items.append(v)
            return frozenset(items), j + 1
        if t.k == "id":
            if t.s in ("TRUE", "FALSE"):
                return t.s == "TRUE", j + 1
            return MV(t.s), j + 1
        raise SyntaxError(f"cfg: unexpected {t.s!r} at line {t.line}")

    while i < len(toks):
        t = toks[i]
        if t.k != "id" or t.s not in KW:
            raise SyntaxError(f"cfg: expected a statement keyword at line {t.line}, found {t.s!r}")
        kw = t.s
        i += 1
        if kw == "CHECK_DEADLOCK":   # TLC2's statement (not in the 2001 grammar): TRUE | FALSE
            out["check_deadlock"] = toks[i].s == "TRUE"
            i += 1
        elif kw in ("SPECIFICATION", "INIT", "NEXT", "SYMMETRY", "VIEW"):
            key = {"SPECIFICATION": "spec", "INIT": "init", "NEXT": "next", "SYMMETRY": "symmetry", "VIEW": "view"}[kw]
            out[key] = toks[i].s
            i += 1
        elif kw in ("INVARIANT", "INVARIANTS", "CONSTRAINT", "CONSTRAINTS", "PROPERTY", "PROPERTIES", "ACTION_CONSTRAINT", "ACTION_CONSTRAINTS"):
            key = "invariants" if kw.startswith("INV") else "constraints" if kw.startswith("CONSTRAINT") else "properties"
            while i < len(toks) and not (toks[i].k == "id" and toks[i].s in KW):
                out[key].append(toks[i].s)
                i += 1
        else:
            while i < len(toks) and not (toks[i].k == "id" and toks[i].s in KW):
                name = toks[i].s
                i += 1
                if toks[i].k == "sym" and toks[i].s == "=":
                    v, i = value(i + 1)
                    out["constants"][name] = v
                elif toks[i].k == "sym" and toks[i].s == "<-":
                    i += 1
                    if toks[i].k == "sym" and toks[i].s == "[":  # <-[Module] Id: only inside that module (MCPaxos.cfg:9)
                        out["scoped"][(toks[i + 1].s, name)] = toks[i + 3].s
                        i += 4
                        continue
                    out["overrides"][name] = toks[i].s
                    i += 1
                else:
                    raise SyntaxError(f"cfg: expected = or <- after {name} at line {toks[i].line}")
    return out


# =============================================================================================== checker
class Checker:
    """TLC's breadth-first search over a Spec"""

    def __init__(self, tla_path, cfg_text=None, cfg_path=None, search=(), clash="test", constants=None, symmetry=True):
        if cfg_text is None:
            cfg_text = Path(cfg_path or str(tla_path)[:-4] + ".cfg").read_text()
        self.cfg = parse_cfg(cfg_text)
        consts = dict(self.cfg["constants"])
        consts.update(constants or {})
        self.spec = Spec(tla_path, search=search, constants=consts, overrides=self.cfg["overrides"], clash=clash,
                         scoped=self.cfg["scoped"])
        init, nxt = self.cfg["init"], self.cfg["next"]
        if self.cfg["spec"]:
            init, nxt = self._split_spec(self.cfg["spec"])
        self.init_name, self.next_name = init, nxt
        self.nextf = self.spec.compile_action(nxt)
        self.invs = [(n, self.spec.compile_value(n)) for n in self.cfg["invariants"]]
        self.cons = [(n, self.spec.compile_value(n)) for n in self.cfg["constraints"]]
        self.props = [self._compile_property(n) for n in self.cfg["properties"]]
        self.group = self._symmetry_group(self.cfg["symmetry"]) if self.cfg["symmetry"] and symmetry else None

    # ------------------------------------------------------------------ PROPERTY (safety part) and SYMMETRY
    def _compile_property(self, name):
        """PROPERTY P with P == I /\\ [][A]_v (/\\ fairness): what TLC checks of it without liveness — I on the initial states,
        A \\/ v' = v on every transition it generates (MCVoting.cfg:9 ConsensusSpecBar == C!Spec, MCPaxos.cfg:12)"""
        sp = self.spec
        flat = []

        def walk(n):
            if n[0] == "conj":
                for x in n[1]:
                    walk(x)
            elif n[0] == "paren":
                walk(n[1])
            elif n[0] == "id" and n[1] not in sp.varidx and sp.overrides.get(n[1], n[1]) in sp.defs and \
                    not sp.defs[sp.overrides.get(n[1], n[1])].params and self._is_temporal(sp.defs[sp.overrides.get(n[1], n[1])].body):
                walk(sp.defs[sp.overrides.get(n[1], n[1])].body)
            else:
                flat.append(n)
        walk(("id", name))
        inits, steps = [], []
        for n in flat:
            if n[0] == "temporal" and n[1] == "[]" and n[2][0] == "temporal" and n[2][1] == "[]_":
                steps.append((sp.ca(n[2][2], {}), sp.cv(n[2][3], {})))
            elif self._is_temporal(n) or (n[0] == "call" and n[1][:3] in ("WF_", "SF_")):
                continue  # liveness (<>, ~>, fairness): not checked
            else:
                inits.append(sp.cv(n, {}))
        return name, inits, steps

    def _is_temporal(self, n, seen=None):
        seen = seen if seen is not None else set()
        if isinstance(n, tuple) and n and (n[0] == "temporal" or (n[0] == "op" and n[1] == "~>")
                                           or (n[0] == "call" and isinstance(n[1], str) and n[1][:3] in ("WF_", "SF_"))):
            return True
        if isinstance(n, tuple) and n and n[0] == "id":
            nm = self.spec.overrides.get(n[1], n[1])
            if nm in self.spec.defs and nm not in seen:
                seen.add(nm)
                return self._is_temporal(self.spec.defs[nm].body, seen)
            return False
        if isinstance(n, (tuple, list)):
            return any(self._is_temporal(x, seen) for x in n)
        return False

    def property_violated_init(self, st):
        for k, (_, inits, _) in enumerate(self.props):
            if any(f({}, st, None) is not True for f in inits):
                return k
        return -1

    def property_violated_step(self, st, s2):
        nx = dict(enumerate(s2))
        for k, (_, _, steps) in enumerate(self.props):
            for act, sub in steps:
                if sub({}, st, None) == sub({}, s2, None):
                    continue
                if not any(True for _ in act({}, st, nx)):
                    return k
        return -1

    def _symmetry_group(self, name):
        """the group generated by the cfg's SYMMETRY set of permutations (functions on model values), each as a dict"""
        gens = [dict(fn_items(f)) for f in iter_set(self.spec.compile_value(name)({}, None, None))]
        dom = sorted({k for g in gens for k in g}, key=vkey)
        ident = tuple(dom)
        gens = [tuple(g.get(k, k) for k in dom) for g in gens]
        pos = {k: i for i, k in enumerate(dom)}
        group, todo = {ident}, [ident]
        while todo:
            a = todo.pop()
            for g in gens:
                c = tuple(g[pos[x]] for x in a)
                if c not in group:
                    group.add(c)
                    todo.append(c)
        return [dict(zip(dom, g)) for g in sorted(group, key=lambda g: [vkey(x) for x in g])]

    def canon(self, st):
        """the orbit's key: the least image of the state under the group (any fixed choice gives the same orbit counts)"""
        if not self.group:
            return st
        return min((permute(st, g) for g in self.group), key=vkey)

    def _split_spec(self, name):
        """Spec == Init /\\ [][Next]_vars (/\\ fairness): the first non-temporal conjunct is Init, [][N]_v gives Next"""
        body = self.spec.defs[name].body
        items = body[1] if body[0] == "conj" else [body]
        flat = []

        sp = self.spec

        def walk(n):
            if n[0] == "conj":
                for x in n[1]:
                    walk(x)
            elif n[0] == "paren":
                walk(n[1])
            elif n[0] == "id" and n[1] not in sp.varidx and sp.overrides.get(n[1], n[1]) in sp.defs and \
                    not sp.defs[sp.overrides.get(n[1], n[1])].params and self._is_temporal(sp.defs[sp.overrides.get(n[1], n[1])].body):
                walk(sp.defs[sp.overrides.get(n[1], n[1])].body)  # LSpec == HC /\ WF_hr(HCnxt) with HC == HCini /\ [][HCnxt]_hr
            else:
                flat.append(n)
        for it in items:
            walk(it)
        init = nxt = None
        for n in flat:
            if n[0] == "temporal" and n[1] == "[]" and n[2][0] == "temporal" and n[2][1] == "[]_":
                a = n[2][2]
                nxt = a[1] if a[0] == "id" else a
            elif init is None and not self._is_temporal(n) and not (n[0] == "call" and n[1][:3] in ("WF_", "SF_")):
                init = n[1] if n[0] == "id" else n
        if init is None or nxt is None:
            raise TLAError(f"cannot split {name} into Init and Next")
        return init, nxt

    def in_model(self, st):
        return all(f({}, st, None) is True for _, f in self.cons)

    def violated(self, st):
        for k, (n, f) in enumerate(self.invs):
            if f({}, st, None) is not True:
                return k
        return -1

    def run_levels(self, max_levels=0, max_distinct=0, check_deadlock=True, stop_on_violation=True, keep_states=True, progress=None):
        """returns dict(distinct, generated, depth, verdict, violated_invariant, levels [new states per level], queue_left,
        trace_len, level_states [[state]] when keep_states)"""
        sp = self.spec
        seen = {}
        levels, level_states = [], []
        generated = 0
        verdict, viol_inv, trace_len = "ok", -1, 0
        frontier = []
        err_msg = None

        def note(kind, inv, depth_len):
            nonlocal verdict, viol_inv, trace_len
            if verdict == "ok":
                verdict, viol_inv, trace_len = kind, inv, depth_len
        keys = set() if self.group else None  # SYMMETRY: the orbit decides whether a state is new; the state met first is kept
        for st in sp.init_states(self.init_name):
            generated += 1
            if keys is not None:
                key = self.canon(st)
                if key in keys:
                    continue
            elif st in seen:
                continue
            k = self.violated(st)
            if k >= 0:
                note("invariant", k, 1)
            if self.props and self.property_violated_init(st) >= 0:
                note("property", self.property_violated_init(st), 1)
            if not self.in_model(st):
                continue
            if keys is not None:
                keys.add(key)
            seen[st] = None
            frontier.append(st)
        levels.append(len(frontier))
        if keep_states:
            level_states.append(list(frontier))
        depth = 1
        budget = False
        while frontier and not (verdict != "ok" and stop_on_violation):
            if max_levels and depth >= max_levels:
                budget = True
                break
            if max_distinct and len(seen) >= max_distinct:
                budget = True
                break
            new = []
            for st in frontier:
                nsucc = 0
                try:
                    for s2 in sp.successors(st, self.nextf):
                        nsucc += 1
                        generated += 1
                        if self.props:
                            k = self.property_violated_step(st, s2)
                            if k >= 0:
                                note("property", k, depth + 1)
                        if keys is not None:
                            key = self.canon(s2)
                            if key in keys:
                                continue
                        elif s2 in seen:
                            continue
                        inm = self.in_model(s2)
                        k = self.violated(s2)
                        if k >= 0:
                            note("invariant", k, depth + 1)
                        if inm:
                            if keys is not None:
                                keys.add(key)
                            seen[s2] = st
                            new.append(s2)
                except AssertFail as e:
                    note("assert", -1, depth)
                    err_msg = str(e)
                except TLAError as e:
                    note("spec-error", -1, depth)
                    err_msg = str(e)
                if nsucc == 0 and check_deadlock and self.cfg.get("check_deadlock", True):
                    note("deadlock", -1, depth)
                if verdict != "ok" and stop_on_violation:
                    break
            if verdict != "ok" and stop_on_violation:
                frontier = new
                break
            frontier = new
            if new:
                levels.append(len(new))
                if keep_states:
                    level_states.append(new)
                depth += 1
            if progress:
                progress(depth, len(seen), generated)
        if verdict == "ok" and budget:
            verdict = "budget"
        return dict(distinct=len(seen), generated=generated, depth=depth, verdict=verdict, violated_invariant=viol_inv, levels=levels,
                    queue_left=len(frontier) if verdict != "ok" else 0, trace_len=trace_len, level_states=level_states, error=err_msg,
                    parents=seen)


if __name__ == "__main__":
    import argparse
    import json
    import time
    ap = argparse.ArgumentParser(description="evaluate a TLA+ module the way TLC does (test infrastructure)")
    ap.add_argument("tla")
    ap.add_argument("-config")
    ap.add_argument("-I", action="append", default=[])
    ap.add_argument("-levels", type=int, default=0)
    ap.add_argument("-deadlock", action="store_true", help="do not report deadlock (TLC's -deadlock)")
    ap.add_argument("-nosymmetry", action="store_true", help="ignore the cfg's SYMMETRY")
    ap.add_argument("-naive", action="store_true", help="negative control: a second x' = e is ignored instead of tested")
    a = ap.parse_args()
    t0 = time.time()
    c = Checker(a.tla, cfg_path=a.config, search=a.I, clash="ignore" if a.naive else "test", symmetry=not a.nosymmetry)
    r = c.run_levels(max_levels=a.levels, keep_states=False, check_deadlock=not a.deadlock, progress=lambda d, n, g: print(f"  level {d}: {n} distinct, {g} generated", file=sys.stderr))
    r.pop("level_states")
    r.pop("parents")
    r["seconds"] = time.time() - t0
    print(json.dumps(r))
