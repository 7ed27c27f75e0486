"""Differential fuzzing of the PlusCal front-end: seeded random algorithms — assignments to scalars, a function variable, a RECORD and its
fields, a set and a sequence variable, `||`, if / either / with / await / assert / while / goto, a macro, a PROCEDURE with a parameter — go through both of the
product's back-ends and must describe the same state graph:

  * the translator's TLA+ text (tla_rust_amd/csrc/pcal.cpp: translate), evaluated by the oracle's evaluator (oracle/tla_eval.py), and
  * the compiled bytecode program (tla_rust_amd/csrc/pcal_compile.cpp), run by the host build of the engine's interpreter (tests/_shim),
  * and the same translation evaluated by the product's host evaluator (tla_rust_amd/csrc/tlaeval.cpp), counters only.

Counters, depth, verdict (ok / invariant / assert / deadlock), trace length, per-level counts and the SET of states of every level.
The hand-written specs of specs/pluscal/ cover the constructs one by one; this covers their combinations (the shapes p-manual
section 3 allows inside one step, App. B's translation of each)."""
import random
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent))
import helpers  # noqa: E402
import test_pcal  # noqa: E402
import test_tlaeval  # noqa: E402

MAX_STATES = 3000


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.use_fn = self.r.random() < 0.7
        self.use_rec = self.r.random() < 0.7
        self.use_macro = self.r.random() < 0.5
        self.use_proc = self.r.random() < 0.5
        self.single = self.r.random() < 0.25          # one `process q = 3` beside the process set
        self.use_set = self.r.random() < 0.4          # a set variable: \cup, \, \in, Cardinality, `with` over it
        self.use_seq = self.r.random() < 0.4          # a sequence variable: Append, Head, Tail, Len, q[i]
        self.local = "t"                              # the current process's own variable

    # ---- expressions (integers stay in 0..2)
    def atom(self, env):
        c = ["x", "y", self.local, "self", str(self.r.randrange(3))] + list(env)
        if self.use_fn:
            c += ["f[self]", "f[3 - self]"]
        if self.use_rec:
            c += ["r.a"]
        if self.use_set:
            c += ["Cardinality(s)"]
        if self.use_seq:
            c += ["Len(q)", "(IF q # <<>> THEN Head(q) ELSE 0)", "(IF Len(q) = 2 THEN q[2] ELSE 1)"]
        return self.r.choice(c)

    def iexpr(self, env, depth=0):
        k = self.r.random()
        if depth >= 2 or k < 0.45:
            a = self.atom(env)
            return f"({a} % 3)" if a == "self" else a
        if k < 0.85:
            return f"(({self.iexpr(env, depth + 1)} + {self.iexpr(env, depth + 1)}) % 3)"
        return f"(IF {self.cond(env, depth + 1)} THEN {self.iexpr(env, depth + 1)} ELSE {self.iexpr(env, depth + 1)})"

    def cond(self, env, depth=0):
        k = self.r.random()
        if depth >= 2 or k < 0.6:
            if self.use_rec and self.r.random() < 0.15:
                return self.r.choice(["r.b", "~r.b"])
            if self.use_set and self.r.random() < 0.15:
                return self.r.choice([f"{self.iexpr(env, 2)} \\in s", f"{self.iexpr(env, 2)} \\notin s", "s = {}", "s \\subseteq {0, 1}"])
            if self.use_seq and self.r.random() < 0.12:
                return self.r.choice(["q = <<>>", "Len(q) < 2", "q # <<>>"])
            return f"{self.iexpr(env, 2)} {self.r.choice(['=', '#', '<', '<=', '>'])} {self.iexpr(env, 2)}"
        op = self.r.choice(["/\\", "\\/"])
        return f"({self.cond(env, depth + 1)} {op} {self.cond(env, depth + 1)})"

    # ---- statements of one step; `done` = variables already assigned on this path
    def assign(self, env, done):
        targets = [v for v in ["x", "y", self.local] if v not in done]
        if self.use_fn and "f" not in done:
            targets.append("f")
        if self.use_rec and "r" not in done:
            targets += ["r", "r.a", "r.b", "r.ab"]
        if self.use_set and "s" not in done:
            targets.append("s")
        if self.use_seq and "q" not in done:
            targets.append("q")
        if not targets:
            return "skip"
        v = self.r.choice(targets)
        if v == "f":
            done.add("f")
            return f"f[self] := {self.iexpr(env)}"
        if v == "s":
            done.add("s")
            return self.r.choice([f"s := s \\cup {{{self.iexpr(env)}}}", f"s := s \\ {{{self.iexpr(env)}}}", "s := {}", f"s := {{{self.iexpr(env)}, {self.iexpr(env)}}}"])
        if v == "q":
            done.add("q")
            return self.r.choice([f"if Len(q) < 2 then q := Append(q, {self.iexpr(env)}); end if", "if q # <<>> then q := Tail(q); end if", "q := <<>>"])
        if v == "r":
            done.add("r")
            return f"r := [a |-> {self.iexpr(env)}, b |-> {self.cond(env)}]"
        if v == "r.a":
            done.add("r")
            return f"r.a := {self.iexpr(env)}"
        if v == "r.b":
            done.add("r")
            return f"r.b := {self.cond(env)}"
        if v == "r.ab":
            done.add("r")
            return f"r.a := {self.iexpr(env)} || r.b := {self.cond(env)}"
        done.add(v)
        if self.r.random() < 0.2:
            w = [u for u in ["x", "y", self.local] if u not in done]
            if w:
                u = self.r.choice(w)
                done.add(u)
                return f"{v} := {self.iexpr(env)} || {u} := {self.iexpr(env)}"
        return f"{v} := {self.iexpr(env)}"

    def simple(self, env, done):
        k = self.r.random()
        if k < 0.62:
            return self.assign(env, done)
        if k < 0.72:
            return f"await {self.cond(env)}"
        if k < 0.77:
            return f"assert {self.cond(env)} \\/ x < 3"          # mostly true; sometimes x < 3 is dropped below
        if k < 0.82:
            return "skip"
        if k < 0.9 and self.use_macro:
            v = self.r.choice([u for u in ["x", "y", self.local] if u not in done] or [None])
            if v:
                done.add(v)
                return f"bump({v}, {self.iexpr(env)})"
        return self.assign(env, done)

    def block(self, env, done, n):
        return "; ".join(self.simple(env, done) for _ in range(n)) + ";"

    def compound(self, env, done, labels, last):
        """one statement that may hold branches; `last` = it ends the step (a goto may close a branch)"""
        k = self.r.random()

        def branch():
            d = set(done)
            s = self.block(env, d, self.r.randrange(1, 3))
            if last and labels and self.r.random() < 0.3:
                s += f" goto {self.r.choice(labels)};"
            return s, d
        if k < 0.4:
            a, da = branch()
            if self.r.random() < 0.6:
                b, db = branch()
                done |= da | db
                return f"if {self.cond(env)} then {a} else {b} end if;"
            done |= da
            return f"if {self.cond(env)} then {a} end if;"
        if k < 0.6:
            a, da = branch()
            b, db = branch()
            done |= da | db
            return f"either {a} or {b} end either;"
        if k < 0.8:
            v = "v" if "v" not in env else "w"
            d = set(done)
            body = self.block(list(env) + [v], d, self.r.randrange(1, 3))
            done |= d
            dom = self.r.choice(["{0, 1, 2}", "0..1", "{1, 2}"] + (["s", "s \\cup {2}"] if self.use_set else []))
            if self.r.random() < 0.3:
                return f"with {v} = {self.iexpr(env)} do {body} end with;"
            return f"with {v} \\in {dom} do {body} end with;"
        return self.block(env, done, 1)

    def step(self, label, labels, may_call, is_last):
        done = set()
        parts = []
        if self.r.random() < 0.18:
            body = self.block([], set(), self.r.randrange(1, 3))
            return f"  {label}: while t < 2 do t := t + 1; {body.replace('t :=', 'y :=') if 't :=' in body else body} end while;"
        n = self.r.randrange(1, 4)
        for i in range(n):
            last = i == n - 1
            if self.r.random() < 0.35:
                parts.append(self.compound([], done, labels, last and not may_call))
            else:
                parts.append(self.simple([], done) + ";")
        if may_call and self.use_proc and not is_last and self.r.random() < 0.5:
            parts.append(f"call inc({self.iexpr([])});")
        elif labels and self.r.random() < 0.15 and "goto" not in parts[-1]:
            parts.append(f"goto {self.r.choice(labels)};")
        return f"  {label}: " + " ".join(parts)

    def program(self):
        n = self.r.randrange(2, 5)
        labels = [f"L{k}" for k in range(1, n + 1)]
        steps = [self.step(lb, labels, True, k == n - 1) for k, lb in enumerate(labels)]
        text = "---- MODULE Fz ----\nEXTENDS Naturals, Sequences, FiniteSets, TLC\n(* --algorithm Fz\nvariables x = 0, y = 1"
        if self.use_fn:
            text += ", f = [i \\in 1..2 |-> 0]"
        if self.use_rec:
            text += ", r = [a |-> 0, b |-> FALSE]"
        if self.use_set:
            text += ", s = {}"
        if self.use_seq:
            text += ", q = <<>>"
        text += ";\n"
        if self.use_macro:
            text += "macro bump(v, d) begin v := (v + d + 1) % 3; end macro;\n"
        if self.use_proc:
            text += "procedure inc(d)\nvariables k = 0;\nbegin\n  I1: k := (d + 1) % 3;\n  I2: x := (x + k) % 3;\n      return;\nend procedure;\n"
        text += "process p \\in 1..2\nvariables t = 0;\nbegin\n" + "\n".join(steps) + "\nend process;\n"
        if self.single:
            fn, self.use_fn, self.local = self.use_fn, False, "u"      # (f is a function on the process SET: f[3] is outside its domain)
            q1, q2 = self.block([], set(), 2), self.block([], set(), 1)
            self.use_fn, self.local = fn, "t"
            text += "process q = 3\nvariables u = 1;\nbegin\n  Q1: " + q1 + "\n  Q2: " + q2 + "\nend process;\n"
        text += "end algorithm *)\n"
        text += "Small == x + y < 4\nTyped == x \\in 0..2 /\\ y \\in 0..2\n"
        if self.use_rec:
            text += "RecOk == r.a \\in 0..2 /\\ (r.b \\/ ~r.b)\n"
        text += "====\n"
        invs = ["Typed"] + (["RecOk"] if self.use_rec else []) + (["Small"] if self.r.random() < 0.5 else [])
        return text.replace(" \\/ x < 3", "" if self.r.random() < 0.3 else " \\/ x < 3"), invs


@pytest.mark.parametrize("seed", range(300))
def test_random_algorithm_translated_vs_compiled(seed, tmp_path):
    text, invs = Gen(seed).program()
    try:
        helpers.pcal_translate(text)
    except RuntimeError as e:   # a shape PlusCal forbids (e.g. a second assignment in a step through a macro): refused, with a message
        assert str(e).strip(), text
        pytest.skip(f"refused: {e}")
    prog = helpers.ShimProgram(text, invs, {})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    if r["distinct"] > MAX_STATES:
        pytest.skip(f"{r['distinct']} states: too many for the Python evaluator in a unit test")
    path = tmp_path / "Fz.tla"
    path.write_text(text)
    try:
        test_pcal.test_compiled_program_vs_tla_evaluator(path, invs, {})
        # ... and a third opinion: the product's host evaluator (tla_rust_amd/csrc/tlaeval.cpp) on the translation
        test_tlaeval.test_pluscal_translation_evaluated_vs_compiled_program(path, invs, {}, tmp_path)
    except AssertionError:
        print(text)
        raise


class ChanGen(Gen):
    """... plus the CHANNEL shapes of round 5's last part: `box`, an ARRAY of sequences of numbers, `m`, a SEQUENCE of RECORDS (kept as
    one sequence per field), and `ms`, a SET of RECORDS (sorted cells), read and assigned inside the same random steps (a subclass: the
    300 programs above stay what they were)"""

    def __init__(self, seed):
        super().__init__(seed + 100000)
        self.single = False          # (box is a function on the process SET)
        self.chan = True

    def atom(self, env):
        if getattr(self, "chan", False) and self.r.random() < 0.3:
            i = self.r.choice(["self", "3 - self"])
            return self.r.choice([f"Len(box[{i}])", f"(IF box[{i}] # <<>> THEN Head(box[{i}]) ELSE 0)", f"(IF Len(box[{i}]) = 2 THEN box[{i}][2] ELSE 1)",
                                  "Len(m)", "(IF m # <<>> THEN Head(m).a ELSE 0)", "(IF Len(m) = 2 THEN m[2].a ELSE 1)", "(Cardinality(ms) % 3)"])
        return super().atom(env)

    def cond(self, env, depth=0):
        if getattr(self, "chan", False) and self.r.random() < 0.15:
            return self.r.choice(["box[3 - self] = <<>>", "box[self] # <<>>", "m # <<>>", "(m # <<>> /\\ Head(m).b)", "(Len(m) = 2 /\\ ~m[2].b)", "Len(box[self]) < 2",
                                  "ms = {}", f"[a |-> {self.iexpr(env, 2)}, b |-> TRUE] \\in ms", f"[a |-> {self.iexpr(env, 2)}, b |-> FALSE] \\notin ms",
                                  f"(\\E e \\in ms : e.a = {self.iexpr(env, 2)})", "(\\A e \\in ms : e.b \\/ e.a < 2)"])
        return super().cond(env, depth)

    def assign(self, env, done):
        if getattr(self, "chan", False) and self.r.random() < 0.35:
            c = []
            if "box" not in done:
                i = self.r.choice(["self", "3 - self"])
                c += [("box", f"if Len(box[{i}]) < 2 then box[{i}] := Append(box[{i}], {self.iexpr(env)}); end if"),
                      ("box", "if box[self] # <<>> then box[self] := Tail(box[self]); end if"),
                      ("box", f"box[{i}] := <<>>"),
                      ("box", f"box[self] := <<{self.iexpr(env)}, {self.iexpr(env)}>>"),
                      ("box", f"if Len(box[{i}]) < 2 then box[{i}] := box[{i}] \\o <<{self.iexpr(env)}>>; end if")]
            if "m" not in done:
                rec = f"[a |-> {self.iexpr(env)}, b |-> {self.cond(env)}]"
                c += [("m", f"if Len(m) < 2 then m := Append(m, {rec}); end if"), ("m", "if m # <<>> then m := Tail(m); end if"), ("m", "m := <<>>"),
                      ("m", f"if m # <<>> then m[1] := {rec}; end if"), ("m", f"m := <<{rec}>>")]
                if self.use_rec:
                    c += [("m", "if Len(m) < 2 then m := Append(m, r); end if"), ("m", "if Len(m) = 2 then m[2] := r; end if")]
                    if "r" not in done:
                        c += [("r", "if m # <<>> then r := Head(m); end if"), ("mr", "if m # <<>> then r := Head(m) || m := Tail(m); end if")]
                if "box" not in done:
                    c += [("mbox", f"if Len(m) < 2 /\\ box[self] # <<>> then m := Append(m, [a |-> Head(box[self]), b |-> TRUE]) || box[self] := Tail(box[self]); end if")]
            if "ms" not in done:
                rec = f"[a |-> {self.iexpr(env)}, b |-> {self.cond(env)}]"
                c += [("ms", f"ms := ms \\cup {{{rec}}}"), ("ms", f"ms := ms \\ {{{rec}}}"), ("ms", "ms := {}"), ("ms", f"ms := (ms \\ {{{rec}}}) \\cup {{[a |-> 0, b |-> TRUE]}}")]
                free = [u for u in ["x", "y"] if u not in done]
                if free:
                    c += [("msw:" + free[0], f"with e \\in ms do ms := (ms \\ {{e}}) \\cup {{[a |-> (e.a + 1) % 3, b |-> ~e.b]}}; {free[0]} := e.a; end with")]
                if self.use_rec:
                    c += [("ms", "ms := ms \\cup {r}")]
                    if "r" not in done:
                        c += [("msr", "with e \\in ms do r := e; ms := ms \\ {e}; end with")]
            if c:
                what, stmt = self.r.choice(c)
                if what.startswith("msw:"):
                    done |= {"ms", what[4:]}
                else:
                    done |= {"mr": {"m", "r"}, "mbox": {"m", "box"}, "msr": {"ms", "r"}}.get(what, {what})
                return stmt
        return super().assign(env, done)

    def program(self):
        text, invs = super().program()
        text = text.replace("variables x = 0, y = 1", "variables x = 0, y = 1, box = [i \\in 1..2 |-> <<>>], m = << [a |-> 1, b |-> FALSE] >>, ms = {[a |-> 2, b |-> TRUE]}", 1)
        text = text.replace("====\n", "ChanOk == (\\A i \\in 1..2 : Len(box[i]) <= 2) /\\ (\\A k \\in 1..Len(m) : m[k].a \\in 0..2 /\\ (m[k].b \\/ ~m[k].b)) /\\ (\\A e \\in ms : e.a \\in 0..2) /\\ Cardinality(ms) <= 6\n====\n")
        return text, invs + ["ChanOk"]


@pytest.mark.parametrize("seed", range(120))
def test_random_algorithm_with_channels_translated_vs_compiled(seed, tmp_path):
    text, invs = ChanGen(seed).program()
    try:
        helpers.pcal_translate(text)
    except RuntimeError as e:
        assert str(e).strip(), text
        pytest.skip(f"refused: {e}")
    try:
        prog = helpers.ShimProgram(text, invs, {})
    except RuntimeError as e:   # a limit of the compiled program (choices per step), said so
        assert "too many alternatives" in str(e), text
        pytest.skip(f"refused: {e}")
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    if r["distinct"] > MAX_STATES:
        pytest.skip(f"{r['distinct']} states: too many for the Python evaluator in a unit test")
    path = tmp_path / "Fz.tla"
    path.write_text(text)
    try:
        test_pcal.test_compiled_program_vs_tla_evaluator(path, invs, {})
        test_tlaeval.test_pluscal_translation_evaluated_vs_compiled_program(path, invs, {}, tmp_path)
    except AssertionError:
        print(text)
        raise


@pytest.mark.parametrize("seed", range(80))
def test_random_uniprocess_algorithm_translated_vs_compiled(seed, tmp_path):
    """the same random steps as ONE process without a name (a uniprocess algorithm: `pc` a plain variable, no `self`, the procedure's
    parameter and variable plain variables of the translation) — the shape on which the translator crashed until round 5's last part"""
    import re
    g = ChanGen(seed) if seed % 2 else Gen(seed)
    g.single = False
    text, invs = g.program()
    text = text.replace("process p \\in 1..2\nvariables t = 0;\nbegin\n", "begin\n").replace("end process;\n", "")
    text = text.replace("variables x = 0, y = 1", "variables t = 0, x = 0, y = 1", 1)
    text = re.sub(r"\bself\b", "1", text)
    try:
        helpers.pcal_translate(text)
        prog = helpers.ShimProgram(text, invs, {})
    except RuntimeError as e:
        assert str(e).strip(), text
        pytest.skip(f"refused: {e}")
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    if r["distinct"] > MAX_STATES:
        pytest.skip(f"{r['distinct']} states: too many for the Python evaluator in a unit test")
    path = tmp_path / "Fz.tla"
    path.write_text(text)
    try:
        test_pcal.test_compiled_program_vs_tla_evaluator(path, invs, {})
        test_tlaeval.test_pluscal_translation_evaluated_vs_compiled_program(path, invs, {}, tmp_path)
    except AssertionError:
        print(text)
        raise


class ExprGen(ChanGen):
    """... plus the expression forms of the round's last hours inside the same random steps: CASE, LET, set filters and images, DOMAIN, CHOOSE over
    a set, <=>, `\\in Nat` (a subclass again: the programs above, some of which the GPU tests run, stay what they were)"""

    def __init__(self, seed):
        super().__init__(seed + 200000)

    def iexpr(self, env, depth=0):
        if depth < 2 and self.r.random() < 0.22:
            k = self.r.randrange(8)
            a, b = self.iexpr(env, depth + 1), self.iexpr(env, depth + 1)
            if k == 0:
                return f"(CASE {self.cond(env, depth + 1)} -> {a} [] {self.cond(env, depth + 1)} -> {b} [] OTHER -> 0)"
            if k == 1:
                return f"(LET z_ == {a} IN (z_ + {b}) % 3)"
            if k == 2:
                return f"(Cardinality({{k_ \\in 0..2 : k_ <= {a}}}) % 3)"
            if k == 3:
                return f"Cardinality({{(k_ + {a}) % 2 : k_ \\in 0..1}})"
            if k == 4 and self.use_set:
                return "(IF s # {} THEN (CHOOSE k_ \\in s : TRUE) % 3 ELSE 0)"
            if k == 5 and self.use_set:
                return "Cardinality({k_ \\in s : k_ > 0})"
            if k == 6 and self.use_fn:
                return "(Cardinality(DOMAIN f) % 3)"
            if k == 7 and self.use_seq:
                return "Cardinality({k_ \\in DOMAIN q : q[k_] > 0})"
        return super().iexpr(env, depth)

    def cond(self, env, depth=0):
        if depth < 2 and self.r.random() < 0.12:
            if self.r.random() < 0.5:
                return f"({self.cond(env, depth + 1)} <=> {self.cond(env, depth + 1)})"
            return f"({self.iexpr(env, 2)} \\in Nat)"
        return super().cond(env, depth)


@pytest.mark.parametrize("seed", range(150))
def test_random_algorithm_with_the_wider_expression_language(seed, tmp_path):
    text, invs = ExprGen(seed).program()
    try:
        helpers.pcal_translate(text)
        prog = helpers.ShimProgram(text, invs, {})
    except RuntimeError as e:
        assert str(e).strip(), text
        pytest.skip(f"refused: {e}")
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    if r["distinct"] > MAX_STATES:
        pytest.skip(f"{r['distinct']} states: too many for the Python evaluator in a unit test")
    path = tmp_path / "Fz.tla"
    path.write_text(text)
    try:
        test_pcal.test_compiled_program_vs_tla_evaluator(path, invs, {})
        test_tlaeval.test_pluscal_translation_evaluated_vs_compiled_program(path, invs, {}, tmp_path)
    except AssertionError:
        print(text)
        raise
