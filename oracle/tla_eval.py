"""oracle/tla_eval.py — TEST INFRASTRUCTURE, not product code.

A small explicit-state checker that evaluates the TLA+ text a PlusCal translation consists of, the way TLC
does (p-manual.pdf §4; the reference's workflow is `pcal2tla X.tla; tlc X.tla`, Makefile:3-7):

  * Init / Next are evaluated left to right; `v' = e` (or `v' \\in S`) ASSIGNS v' when it is still unassigned and
    is a test otherwise; UNCHANGED v is v' = v; `\\/`, `\\E` and IF branch the evaluation; Assert(FALSE, msg) is
    an error raised while the action is being evaluated (README.md:268-269).
  * breadth-first search, FIFO, exact de-duplication on the full state (no fingerprints), INVARIANTs checked on
    every new distinct state, deadlock = a state without successor (the PlusCal terminating disjunct counts),
    TLC's counters: generated = initial states + every successor produced, distinct, states left on queue,
    depth of the state graph (README.md:319-321).

Pure Python: for small models only (the pinned one, README.md:267-321, has 6 164 distinct states).
Supported TLA+ subset = what tla_rust_amd/csrc/pcal.cpp emits plus simple invariants: junction lists by
alignment, IF/THEN/ELSE, CASE, \\E \\A, functions ([x \\in S |-> e], [f EXCEPT ![i] = e], f[i], [S -> T]),
tuples, finite sets, .., \\cup \\cap \\ \\in \\notin \\subseteq, integers, strings, booleans, LET (one definition),
operator definitions with parameters, Assert, Len/Append/Head/Tail/Cardinality, UNCHANGED.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under oracle/.
"""
import re
import sys
from collections import deque


# ------------------------------------------------------------------------------------------------ lexer
class Tok:
    __slots__ = ("k", "s", "line", "col")

    def __init__(self, k, s, line, col):
        self.k, self.s, self.line, self.col = k, s, line, col

    def __repr__(self):
        return f"{self.k}:{self.s}@{self.line}:{self.col}"


SYMS = ["|->", "<=>", "<>", "[]", ":=", "||", "==", "=>", "<=", ">=", "=<", "/=", "/\\", "\\/", "..", "->", "<<", ">>", ":>", "@@",
        "(", ")", "[", "]", "{", "}", ",", ";", ":", "+", "-", "*", "%", "=", "<", ">", "#", "~", "'", "!", "@", ".", "^", "_", "\\"]


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
        elif c.isalnum() or c == "_" and i + 1 < n and (text[i + 1].isalnum() or text[i + 1] == "_"):
            j = i
            while j < n and (text[j].isalnum() or text[j] == "_"):
                j += 1
            s = text[i:j]
            toks.append(Tok("num" if s.isdigit() else "id", s, line, col))
            adv(j - i)
        elif c == '"':
            j = text.index('"', i + 1)
            toks.append(Tok("str", text[i + 1:j], line, col))
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


# ------------------------------------------------------------------------------------------------ parser
PREC = {"=>": 1, "<=>": 1, "\\equiv": 1, "\\/": 3, "/\\": 3, "=": 5, "#": 5, "/=": 5, "<": 5, ">": 5, "<=": 5, "=<": 5, ">=": 5, "\\leq": 5, "\\geq": 5,
        "\\in": 5, "\\notin": 5, "\\subseteq": 5, "\\cup": 8, "\\union": 8, "\\cap": 8, "\\intersect": 8, "\\": 8, "..": 9,
        "+": 10, "-": 10, "%": 11, "*": 13, "\\div": 13, "\\o": 13, ":>": 7, "@@": 6}
CANON = {"\\equiv": "<=>", "=<": "<=", "\\leq": "<=", "\\geq": ">=", "/=": "#", "\\union": "\\cup", "\\intersect": "\\cap"}


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0
        self.jstack = [0]  # columns of the enclosing junction-list bullets: a token at or left of the top ends the item

    def cur(self):
        return self.t[self.i]

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
            rhs = self.expr(p + 1)
            lhs = ("op", CANON.get(c.s, c.s), lhs, rhs)
        return lhs

    def junction(self, bullet):
        col = self.cur().col
        items = []
        while self.is_sym(bullet) and self.cur().col == col:
            self.i += 1
            self.jstack.append(col)
            items.append(self.expr(0))
            self.jstack.pop()
        return ("conj" if bullet == "/\\" else "disj", items)

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
                var = self.ident()
                self.expect("\\in")
                dom = self.expr(6)
                self.expect(":")
                return ("quant", c.s[1], var, dom, self.expr(0))
            if c.s in ("[]", "<>"):  # temporal formulas (Spec, Termination) are parsed and never evaluated
                self.i += 1
                return ("temporal", self.expr(4))
        if c.k == "id":
            if c.s == "DOMAIN":
                self.i += 1
                return ("domain", self.postfix(self.atom()))
            if c.s == "CHOOSE":
                self.i += 1
                var = self.ident()
                self.expect("\\in")
                dom = self.expr(6)
                self.expect(":")
                return ("quant", "C", var, dom, self.expr(0))
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
            if c.s == "LET":   # LET a == e  f(x, y) == g  ... IN body
                self.i += 1
                defs = []
                while True:
                    name = self.ident()
                    params = []
                    if self.is_sym("("):
                        self.i += 1
                        while True:
                            params.append(self.ident())
                            if self.is_sym(","):
                                self.i += 1
                                continue
                            break
                        self.expect(")")
                    self.expect("==")
                    defs.append((name, params, self.expr(0)))
                    if self.cur().k == "id" and self.cur().s == "IN":
                        break
                self.expect("IN")
                body = self.expr(0)
                # a definition without parameters is a binding evaluated where the LET stands; one with parameters is substituted into
                # the definitions after it and into the body, at parse time (its body sees the earlier bindings of this LET)
                for k in range(len(defs) - 1, -1, -1):
                    name, params, val = defs[k]
                    if params:
                        rest = [(n, ps, self._subst_call(v, name, params, val)) for n, ps, v in defs[k + 1:]]
                        defs[k + 1:] = rest
                        body = self._subst_call(body, name, params, val)
                for name, params, val in reversed(defs):
                    if not params:
                        body = ("let", name, val, body)
                return body
            if c.s == "UNCHANGED":
                self.i += 1
                return ("unchanged", self.expr(14))
        return self.postfix(self.atom())

    @classmethod
    def _subst_ids(cls, e, m):
        if isinstance(e, tuple):
            if len(e) == 2 and e[0] == "id" and e[1] in m:
                return m[e[1]]
            return tuple(cls._subst_ids(x, m) for x in e)
        if isinstance(e, list):
            return [cls._subst_ids(x, m) for x in e]
        return e

    @classmethod
    def _subst_call(cls, e, name, params, body):
        """every call name(args) inside e replaced by body with params := args"""
        if isinstance(e, tuple):
            if len(e) == 3 and e[0] == "call" and e[1] == name and len(e[2]) == len(params):
                args = [cls._subst_call(a, name, params, body) for a in e[2]]
                return cls._subst_ids(body, dict(zip(params, args)))
            return tuple(cls._subst_call(x, name, params, body) for x in e)
        if isinstance(e, list):
            return [cls._subst_call(x, name, params, body) for x in e]
        return e

    def ident(self):
        c = self.cur()
        if c.k != "id":
            self.fail("expected an identifier")
        self.i += 1
        return c.s

    def atom(self):
        c = self.cur()
        self.i += 1
        if c.k == "num":
            return ("num", int(c.s))
        if c.k == "str":
            return ("str", c.s)
        if c.k == "id":
            if c.s in ("TRUE", "FALSE"):
                return ("bool", c.s == "TRUE")
            if self.is_sym("(") and not self.ended():
                self.i += 1
                args = [self.expr(0)]
                while self.is_sym(","):
                    self.i += 1
                    args.append(self.expr(0))
                self.expect(")")
                return ("call", c.s, args)
            return ("id", c.s)
        if c.k == "sym":
            if c.s == "(":
                e = self.expr(0)
                self.expect(")")
                return e
            if c.s == "{":
                items = []
                if not self.is_sym("}"):
                    items.append(self.expr(0))
                    if self.is_sym(":"):   # {x \\in S : P}   {e : x \\in S}
                        self.i += 1
                        first = items[0]
                        if first[0] == "op" and first[1] == "\\in" and first[2][0] == "id":
                            pred = self.expr(0)
                            self.expect("}")
                            return ("setfilter", first[2][1], first[3], pred)
                        var = self.ident()
                        self.expect("\\in")
                        dom = self.expr(6)
                        self.expect("}")
                        return ("setmap", var, dom, first)
                    while self.is_sym(","):
                        self.i += 1
                        items.append(self.expr(0))
                self.expect("}")
                return ("setenum", items)
            if c.s == "<<":
                items = []
                if not self.is_sym(">>"):
                    items.append(self.expr(0))
                    while self.is_sym(","):
                        self.i += 1
                        items.append(self.expr(0))
                self.expect(">>")
                return ("tuple", items)
            if c.s == "[":
                # [x \in S |-> e]   [f EXCEPT ![i] = e, ![j] = e]   [S -> T]   [][Next]_vars
                if self.cur().k == "id" and self.t[self.i + 1].k == "sym" and self.t[self.i + 1].s == "\\in":
                    var = self.ident()
                    self.i += 1
                    dom = self.expr(6)
                    self.expect("|->")
                    body = self.expr(0)
                    self.expect("]")
                    return ("funcdef", var, dom, body)
                if self.cur().k == "id" and self.t[self.i + 1].k == "sym" and self.t[self.i + 1].s == "|->":
                    # [f |-> e, g |-> h]: a record = a function on the field names (the translator's view of a record variable
                    # it keeps field by field, tla_rust_amd/csrc/pcal.h RECORDS)
                    fields = []
                    while True:
                        name = self.ident()
                        self.expect("|->")
                        fields.append((name, self.expr(0)))
                        if self.is_sym(","):
                            self.i += 1
                            continue
                        break
                    self.expect("]")
                    return ("record", fields)
                first = self.expr(0)
                if self.is_id("EXCEPT"):
                    self.i += 1
                    ups = []
                    while True:
                        self.expect("!")
                        path = []
                        while self.is_sym("["):
                            self.i += 1
                            path.append(self.expr(0))
                            self.expect("]")
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
                    return ("funcset", first, rng)
                self.expect("]")
                if self.is_sym("_"):
                    self.i += 1
                    self.atom()
                return ("temporal", first)
        self.i -= 1
        self.fail("expected an expression")

    def postfix(self, e):
        while not self.ended():
            if self.is_sym("["):
                self.i += 1
                idx = self.expr(0)
                self.expect("]")
                e = ("idx", e, idx)
            elif self.is_sym(".") and self.t[self.i + 1].k == "id":
                self.i += 1
                e = ("idx", e, ("str", self.ident()))
            elif self.is_sym("'"):
                self.i += 1
                e = ("prime", e)
            else:
                break
        return e


class Module:
    def __init__(self, text):
        self.text = text
        toks = lex(text)
        self.variables, self.defs, self.constants = [], {}, []
        self.def_pos = {}
        p = Parser(toks)
        while p.cur().k != "end":
            c = p.cur()
            if c.k == "id" and c.s in ("VARIABLE", "VARIABLES"):
                p.i += 1
                self.variables.append(p.ident())
                while p.is_sym(","):
                    p.i += 1
                    self.variables.append(p.ident())
            elif c.k == "id" and c.s in ("CONSTANT", "CONSTANTS"):
                p.i += 1
                self.constants.append(p.ident())
                while p.is_sym(","):
                    p.i += 1
                    self.constants.append(p.ident())
            elif c.k == "id" and c.col == 1 and c.s not in ("EXTENDS", "MODULE") and self._is_def(p):
                name = p.ident()
                params = []
                if p.is_sym("("):
                    p.i += 1
                    params.append(p.ident())
                    while p.is_sym(","):
                        p.i += 1
                        params.append(p.ident())
                    p.expect(")")
                p.expect("==")
                p.jstack = [1]
                self.defs[name] = (params, p.expr(0))
                self.def_pos[name] = c.line
                p.jstack = [0]
            else:
                p.i += 1

    @staticmethod
    def _is_def(p):
        j = p.i + 1
        if p.t[j].k == "sym" and p.t[j].s == "(":
            while p.t[j].k != "end" and not (p.t[j].k == "sym" and p.t[j].s == ")"):
                j += 1
            j += 1
        return p.t[j].k == "sym" and p.t[j].s == "=="


# ------------------------------------------------------------------------------------------------ values
class Fn:
    """A TLA+ function with a finite domain; tuples are functions on 1..n."""
    __slots__ = ("items", "_h")

    def __init__(self, d):
        self.items = tuple(sorted(d.items(), key=lambda kv: sort_key(kv[0])))
        self._h = hash(self.items)

    def __hash__(self):
        return self._h

    def __eq__(self, o):
        return isinstance(o, Fn) and self.items == o.items

    def get(self, k):
        for a, b in self.items:
            if a == k:
                return b
        raise EvalError(f"function applied outside its domain: {fmt(k)}")

    def domain(self):
        return frozenset(a for a, _ in self.items)

    def is_seq(self):
        return [a for a, _ in self.items] == list(range(1, len(self.items) + 1))


class MV(str):
    """a model value (cfg `c = c`): equal only to itself, printed bare"""
    def __eq__(self, o):
        return isinstance(o, MV) and str.__eq__(self, o)

    def __ne__(self, o):
        return not self.__eq__(o)

    __hash__ = str.__hash__


def sort_key(v):
    if isinstance(v, bool):
        return (0, int(v))
    if isinstance(v, int):
        return (1, v)
    if isinstance(v, str):
        return (2, v)
    if isinstance(v, Fn):
        return (3, tuple(sort_key(b) for _, b in v.items))
    return (4, tuple(sorted(sort_key(x) for x in v)))


def fmt(v):
    if isinstance(v, MV):
        return str(v)
    if isinstance(v, bool):
        return "TRUE" if v else "FALSE"
    if isinstance(v, int):
        return str(v)
    if isinstance(v, str):
        return f'"{v}"'
    if isinstance(v, Fn):
        if v.is_seq():
            return "<<" + ", ".join(fmt(b) for _, b in v.items) + ">>"
        if v.items and all(isinstance(a, str) and not isinstance(a, MV) for a, _ in v.items):   # a record (fields in name order)
            return "[" + ", ".join(f"{a} |-> {fmt(b)}" for a, b in v.items) + "]"
        return "(" + " @@ ".join(f"{fmt(a)} :> {fmt(b)}" for a, b in v.items) + ")"
    return "{" + ", ".join(fmt(x) for x in sorted(v, key=sort_key)) + "}"


class EvalError(Exception):
    pass


class AssertViolation(Exception):
    pass


# ------------------------------------------------------------------------------------------------ evaluator
class Checker:
    def __init__(self, text, constants=None):
        self.engine_mode = False  # True: a failing Assert is one generated successor and the other successors go on
        self.m = Module(text)
        self.vars = self.m.variables
        self.defs = self.m.defs
        self.consts = dict(constants or {})
        if "defaultInitValue" in self.m.constants and "defaultInitValue" not in self.consts:
            self.consts["defaultInitValue"] = MV("defaultInitValue")   # the cfg line TLC needs: defaultInitValue = defaultInitValue

    # ---- value expressions.  env = (state dict, next dict or None, bound dict)
    def ev(self, e, st, nx, bd):
        k = e[0]
        if k == "num" or k == "str" or k == "bool":
            return e[1]
        if k == "id":
            n = e[1]
            if n in bd:
                return bd[n]
            if n in st:
                return st[n]
            if n in self.consts:
                return self.consts[n]
            if n in self.defs and not self.defs[n][0]:
                return self.ev(self.defs[n][1], st, nx, bd)
            raise EvalError(f"unknown identifier {n}")
        if k == "prime":
            if e[1][0] != "id" or nx is None or e[1][1] not in nx:
                raise EvalError(f"primed expression read before it is assigned: {e[1]}")
            return nx[e[1][1]]
        if k == "op":
            o = e[1]
            if o == "/\\":
                return self.ev(e[2], st, nx, bd) and self.ev(e[3], st, nx, bd)
            if o == "\\/":
                return self.ev(e[2], st, nx, bd) or self.ev(e[3], st, nx, bd)
            if o == "=>":
                return (not self.ev(e[2], st, nx, bd)) or self.ev(e[3], st, nx, bd)
            if o in ("\\in", "\\notin") and e[3] in (("id", "Nat"), ("id", "Int")):    # TypeOK conjuncts: x \\in Nat
                a = self.ev(e[2], st, nx, bd)
                yes = isinstance(a, int) and not isinstance(a, bool) and (a >= 0 or e[3][1] == "Int")
                return yes if o == "\\in" else not yes
            a, b = self.ev(e[2], st, nx, bd), self.ev(e[3], st, nx, bd)
            if o == "<=>":
                return bool(a) == bool(b)
            if o == "=":
                return a == b
            if o == "#":
                return a != b
            if o == "+":
                return a + b
            if o == "-":
                return a - b
            if o == "*":
                return a * b
            if o == "\\div":
                return a // b
            if o == "%":
                return a % b
            if o == "<":
                return a < b
            if o == ">":
                return a > b
            if o == "<=":
                return a <= b
            if o == ">=":
                return a >= b
            if o == "..":
                return frozenset(range(a, b + 1))
            if o == "\\in":
                return a in b
            if o == "\\notin":
                return a not in b
            if o == "\\cup":
                return a | b
            if o == "\\cap":
                return a & b
            if o == "\\":
                return a - b
            if o == "\\subseteq":
                return a <= b
            if o == ":>":
                return Fn({a: b})
            if o == "@@":
                d = dict(b.items)
                d.update(dict(a.items))
                return Fn(d)
            if o == "\\o":
                xs = [v for _, v in a.items] + [v for _, v in b.items]
                return Fn({i + 1: v for i, v in enumerate(xs)})
            raise EvalError(f"operator {o} not supported")
        if k == "not":
            return not self.ev(e[1], st, nx, bd)
        if k == "neg":
            return -self.ev(e[1], st, nx, bd)
        if k == "conj":
            return all(self.ev(x, st, nx, bd) for x in e[1])
        if k == "disj":
            return any(self.ev(x, st, nx, bd) for x in e[1])
        if k == "if":
            return self.ev(e[2] if self.ev(e[1], st, nx, bd) else e[3], st, nx, bd)
        if k == "case":
            for g, v in e[1]:
                if self.ev(g, st, nx, bd):
                    return self.ev(v, st, nx, bd)
            if e[2] is not None:
                return self.ev(e[2], st, nx, bd)
            raise EvalError("CASE without a matching arm")
        if k == "idx":
            f = self.ev(e[1], st, nx, bd)
            return f.get(self.ev(e[2], st, nx, bd))
        if k == "quant":
            dom = sorted(self.ev(e[3], st, nx, bd), key=sort_key)
            if e[1] == "C":   # CHOOSE: TLC takes the first element of its (sorted) enumeration that satisfies the body
                for v in dom:
                    if self.ev(e[4], st, nx, {**bd, e[2]: v}):
                        return v
                raise EvalError("CHOOSE: no element of the set satisfies the predicate")
            gen = (self.ev(e[4], st, nx, {**bd, e[2]: v}) for v in dom)
            return all(gen) if e[1] == "A" else any(gen)
        if k == "setenum":
            return frozenset(self.ev(x, st, nx, bd) for x in e[1])
        if k == "setfilter":
            return frozenset(v for v in self.ev(e[2], st, nx, bd) if self.ev(e[3], st, nx, {**bd, e[1]: v}))
        if k == "setmap":
            return frozenset(self.ev(e[3], st, nx, {**bd, e[1]: v}) for v in self.ev(e[2], st, nx, bd))
        if k == "domain":
            f = self.ev(e[1], st, nx, bd)
            return f.domain()
        if k == "tuple":
            return Fn({i + 1: self.ev(x, st, nx, bd) for i, x in enumerate(e[1])})
        if k == "record":
            return Fn({name: self.ev(x, st, nx, bd) for name, x in e[1]})
        if k == "funcdef":
            return Fn({v: self.ev(e[3], st, nx, {**bd, e[1]: v}) for v in self.ev(e[2], st, nx, bd)})
        if k == "funcset":
            dom = sorted(self.ev(e[1], st, nx, bd), key=sort_key)
            rng = sorted(self.ev(e[2], st, nx, bd), key=sort_key)
            out = [{}]
            for d in dom:
                out = [{**f, d: r} for f in out for r in rng]
            return frozenset(Fn(f) for f in out)
        if k == "except":
            f = self.ev(e[1], st, nx, bd)
            for path, val in e[2]:
                f = self._except(f, [self.ev(p, st, nx, bd) for p in path], val, st, nx, bd)
            return f
        if k == "let":
            return self.ev(e[3], st, nx, {**bd, e[1]: self.ev(e[2], st, nx, bd)})
        if k == "call":
            name, args = e[1], [self.ev(a, st, nx, bd) for a in e[2]]
            if name == "Assert":
                if not args[0]:
                    raise AssertViolation(args[1])
                return True
            if name == "Len":
                return len(args[0].items)
            if name == "Cardinality":
                return len(args[0])
            if name == "Head":
                return args[0].get(1)
            if name == "Tail":
                return Fn({i: v for i, (_, v) in enumerate(args[0].items[1:], 1)})
            if name == "Append":
                return Fn({**dict(args[0].items), len(args[0].items) + 1: args[1]})
            if name == "PrintT":
                return True
            if name in self.defs:
                params, body = self.defs[name]
                return self.ev(body, st, nx, {**bd, **dict(zip(params, args))})
            raise EvalError(f"unknown operator {name}")
        if k == "unchanged":
            return all(nx[v] == st[v] for v in self._vars_of(e[1]))
        raise EvalError(f"cannot evaluate {k}")

    def _except(self, f, path, val, st, nx, bd):
        if not path:
            return self.ev(val, st, nx, bd)
        d = dict(f.items)
        if path[0] not in d:
            raise EvalError("EXCEPT outside the domain")
        d[path[0]] = self._except(d[path[0]], path[1:], val, st, nx, bd)
        return Fn(d)

    def _vars_of(self, e):
        if e[0] == "id":
            if e[1] in self.vars:
                return [e[1]]
            if e[1] in self.defs:
                return self._vars_of(self.defs[e[1]][1])
        if e[0] == "tuple":
            return [v for x in e[1] for v in self._vars_of(x)]
        raise EvalError(f"UNCHANGED of {e}")

    # ---- action evaluation: yields completed/extended next-state dicts.  `target` = "next" (v' is assigned) or
    # "init" (the unprimed variable is assigned: evaluating Init)
    def act(self, e, st, nx, bd, init=False):
        k = e[0]
        if k == "conj" or (k == "op" and e[1] == "/\\"):
            items = e[1] if k == "conj" else [e[2], e[3]]

            def rec(i, cur):
                if i == len(items):
                    yield cur
                    return
                for n2 in self.act(items[i], cur if init else st, cur, bd, init):
                    if "__assert__" in n2:
                        yield n2
                    else:
                        yield from rec(i + 1, n2)

            yield from rec(0, nx)
            return
        if k == "disj" or (k == "op" and e[1] == "\\/"):
            for x in (e[1] if k == "disj" else [e[2], e[3]]):
                yield from self.act(x, st, nx, bd, init)
            return
        if k == "quant" and e[1] == "E":
            for v in sorted(self.ev(e[3], st, nx, bd), key=sort_key):
                yield from self.act(e[4], st, nx, {**bd, e[2]: v}, init)
            return
        if k == "if":
            yield from self.act(e[2] if self.ev(e[1], st, nx, bd) else e[3], st, nx, bd, init)
            return
        if k == "let":
            yield from self.act(e[3], st, nx, {**bd, e[1]: self.ev(e[2], st, nx, bd)}, init)
            return
        if k == "unchanged":
            cur = nx
            for v in self._vars_of(e[1]):
                if v in cur:
                    if cur[v] != st[v]:
                        return
                else:
                    cur = {**cur, v: st[v]}
            yield cur
            return
        if k == "op" and e[1] in ("=", "\\in"):
            tgt = None
            if not init and e[2][0] == "prime" and e[2][1][0] == "id":
                tgt = e[2][1][1]
            if init and e[2][0] == "id" and e[2][1] in self.vars:
                tgt = e[2][1]
            if tgt is not None and tgt not in nx:
                val = self.ev(e[3], st, nx, bd)
                if e[1] == "=":
                    yield {**nx, tgt: val}
                else:
                    for v in sorted(val, key=sort_key):
                        yield {**nx, tgt: v}
                return
        if k == "call" and e[1] in self.defs:
            params, body = self.defs[e[1]]
            args = [self.ev(a, st, nx, bd) for a in e[2]]
            yield from self.act(body, st, nx, {**bd, **dict(zip(params, args))}, init)
            return
        if k == "id" and e[1] in self.defs and not self.defs[e[1]][0] and e[1] not in bd:
            yield from self.act(self.defs[e[1]][1], st, nx, bd, init)
            return
        if self.engine_mode and k == "call" and e[1] == "Assert":
            if not self.ev(e[2][0], st, nx, bd):
                yield {"__assert__": self.ev(e[2][1], st, nx, bd)}
                return
            yield nx
            return
        if self.ev(e, st, nx, bd):
            yield nx

    def initial_states(self, init="Init"):
        for s in self.act(self.defs[init][1], {}, {}, {}, init=True):
            missing = [v for v in self.vars if v not in s]
            if missing:
                raise EvalError(f"Init leaves {missing} unassigned")
            yield s

    def successors(self, st, nxt="Next"):
        for n in self.act(self.defs[nxt][1], st, {}, {}):
            if "__assert__" in n:
                yield n
                continue
            missing = [v for v in self.vars if v not in n]
            if missing:
                raise EvalError(f"Next leaves {missing} unassigned")
            yield n

    def key(self, s):
        return tuple(s[v] for v in self.vars)

    def fmt_state(self, s):
        return "\n".join(f"/\\ {v} = {fmt(s[v])}" for v in self.vars)

    def in_model(self, s, constraints):
        """cfg CONSTRAINT (FIFO/MCInnerFIFO.cfg:23-26, p-manual section 4.3 p.36): a state that does not satisfy it is generated
        and checked, but neither stored nor expanded."""
        return all(self.ev(self.defs[name][1], s, None, {}) for name in constraints)

    def run_levels(self, invariants=(), check_deadlock=True, max_distinct=0, init="Init", nxt="Next", constraints=()):
        """The engine's convention (include/tlamc.h): a violation does not cut the level short — every state of the
        level is expanded, every enabled successor counted (a failing Assert is one generated successor that is
        not stored), and the search stops at the end of that level.  Also returns the state texts per level."""
        self.engine_mode = True
        seen, levels_states = {}, []
        res = dict(distinct=0, generated=0, queue_left=0, depth=0, verdict="ok", violated=None, trace_len=0, levels=[], states=levels_states)
        # `errors`: EVERY (verdict, violated invariant, trace length) the failing pass holds.  verdict / violated / trace_len are the first
        # one in this evaluator's order; an engine that expands a level in parallel may report any of them (as TLC does with several workers)
        errors = set()

        def first_violated(st):
            for name in invariants:
                if not self.ev(self.defs[name][1], st, None, {}):
                    return name
            return None
        cur = []
        for s in self.initial_states(init):
            res["generated"] += 1
            k = self.key(s)
            if k not in seen:
                bad = first_violated(s)
                if bad:
                    errors.add(("invariant", bad, 1))
                    if res["verdict"] == "ok":
                        res.update(verdict="invariant", violated=bad, trace_len=1)
                if not self.in_model(s, constraints):
                    continue
                seen[k] = 1
                cur.append(s)
        level = 1
        while cur:
            res["levels"].append(len(cur))
            levels_states.append(sorted(self.fmt_state(s).replace("\n", " ") for s in cur))
            if res["verdict"] != "ok" or (max_distinct and len(seen) >= max_distinct):
                break
            nxt_level = []
            for s in cur:
                nsucc = 0
                for n in self.successors(s, nxt):
                    nsucc += 1
                    res["generated"] += 1
                    if "__assert__" in n:
                        errors.add(("assert", None, level))
                        if res["verdict"] == "ok":
                            res.update(verdict="assert", trace_len=level, message=n["__assert__"])
                        continue
                    k = self.key(n)
                    if k in seen:
                        continue
                    bad = first_violated(n)
                    if bad:
                        errors.add(("invariant", bad, level + 1))
                        if res["verdict"] == "ok":
                            res.update(verdict="invariant", violated=bad, trace_len=level + 1)
                    if not self.in_model(n, constraints):
                        continue
                    seen[k] = 1
                    nxt_level.append(n)
                if nsucc == 0 and check_deadlock:
                    errors.add(("deadlock", None, level))
                    if res["verdict"] == "ok":
                        res.update(verdict="deadlock", trace_len=level)
            cur = nxt_level
            if cur:
                level += 1
        if res["verdict"] != "ok" and cur and len(res["levels"]) < level:
            res["levels"].append(len(cur))
            levels_states.append(sorted(self.fmt_state(s).replace("\n", " ") for s in cur))
        res["errors"] = sorted(errors, key=lambda e: (e[2], e[0], e[1] or ""))
        res.update(distinct=len(seen), depth=level, queue_left=len(cur) if res["verdict"] != "ok" else 0)
        self.engine_mode = False
        return res

    def run(self, invariants=(), check_deadlock=True, max_distinct=0, init="Init", nxt="Next", constraints=()):
        """TLC-like BFS.  Returns dict(distinct, generated, queue_left, depth, verdict, violated, trace, levels)."""
        seen, order, parent = {}, [], []
        res = dict(distinct=0, generated=0, queue_left=0, depth=0, verdict="ok", violated=None, trace=[], levels=[], message=None)

        def trace_to(i):
            out = []
            while i is not None:
                out.append(self.fmt_state(order[i]))
                i = parent[i]
            return out[::-1]

        def add(s, par):
            k = self.key(s)
            if k in seen:
                return None
            if not self.in_model(s, constraints):   # generated and checked, not stored (returns -1: check it, do not keep it)
                return -1
            seen[k] = len(order)
            order.append(s)
            parent.append(par)
            return len(order) - 1

        def bad_inv(s):
            for name in invariants:
                if not self.ev(self.defs[name][1], s, None, {}):
                    return name
            return None

        for s in self.initial_states(init):
            res["generated"] += 1
            i = add(s, None)
            if i is not None:
                b = bad_inv(s)
                if b:
                    res.update(verdict="invariant", violated=b, trace=trace_to(i) if i >= 0 else [self.fmt_state(s)], distinct=len(order), depth=1,
                               queue_left=max(0, len(order) - 1))
                    return res
        lo, hi, level = 0, len(order), 1
        res["levels"].append(hi)
        while hi > lo:
            if max_distinct and len(order) >= max_distinct:
                res["verdict"] = "budget"
                break
            for i in range(lo, hi):
                s = order[i]
                nsucc = 0
                stop = None
                try:
                    for n in self.successors(s, nxt):
                        nsucc += 1
                        res["generated"] += 1
                        j = add(n, i)
                        if j is not None:
                            b = bad_inv(n)
                            if b:
                                stop = ("invariant", b, trace_to(j) if j >= 0 else trace_to(i) + [self.fmt_state(n)], None)
                                break
                except AssertViolation as a:
                    stop = ("assert", None, trace_to(i), str(a))
                if stop is None and nsucc == 0 and check_deadlock:
                    stop = ("deadlock", None, trace_to(i), None)
                if stop:
                    res.update(verdict=stop[0], violated=stop[1], trace=stop[2], message=stop[3], distinct=len(order),
                               depth=level + (1 if len(order) > hi else 0), queue_left=len(order) - (i + 1))
                    return res
            lo, hi = hi, len(order)
            if hi > lo:
                level += 1
                res["levels"].append(hi - lo)
        res.update(distinct=len(order), depth=level, queue_left=hi - lo)
        return res


def main(argv):
    import json
    text = open(argv[1]).read()
    invs = argv[2:]
    r = Checker(text).run(invariants=invs)
    print(json.dumps({k: v for k, v in r.items() if k != "trace"}))
    for k, s in enumerate(r["trace"], 1):
        print(f"State {k}:\n{s}\n")


if __name__ == "__main__":
    main(sys.argv)
