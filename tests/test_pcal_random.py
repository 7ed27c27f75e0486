"""Differential test of the PlusCal compiler: seeded random algorithms (labels, assignments to scalars / function
variables / process locals, if / elsif / else with and without labels inside, either, with, while, await, goto,
assert, skip, ||) are compiled to the bytecode program and run by the host build of the interpreter, and the
TRANSLATION of the same algorithm is evaluated by oracle/tla_eval.py: counters, verdict and the per-level sets of
states must be identical.  Two independent routes from one source text (compile vs translate + evaluate) — and a third: the
translation evaluated by the product's own host evaluator (C++, tests/_tlaeval door), counters and verdict."""
import os
import random
import sys
import tempfile
from pathlib import Path

import pytest

import helpers

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
from tla_eval import Checker  # noqa: E402

K = 3  # values live in 0..K-1


class Gen:
    def __init__(self, seed):
        self.r = random.Random(seed)
        self.nlabel = 0

    def label(self):
        self.nlabel += 1
        return f"L{self.nlabel}"

    def expr(self, local):
        r = self.r
        atoms = ["x", "y", "a[1]", "a[2]", "self", str(r.randrange(K)), "Len(q)", "Cardinality(s)", "Sum", "Bump(x)",
                 "Mix(y, a[1])"] + (["t"] if local else [])
        e = r.choice(atoms)
        if r.random() < 0.5:
            e = f"({e} + {r.choice(atoms)}) % {K}"
        return e

    def cond(self, local):
        r = self.r
        c = f"{self.expr(local)} {r.choice(['=', '#', '<', '>='])} {self.expr(local)}"
        k = r.random()
        if k < 0.05:
            c = r.choice(['m = "a"', 'm # "b"', "f", "~f", 'f /\\ m = "b"', "f = (x = y)"])
        elif k < 0.1:
            c = f"{self.expr(local)} \\in s"
        elif k < 0.15:
            c = f"q # <<>>"
        elif k < 0.2:
            c = f"s \\subseteq {{0, 1}} \\/ {c}"
        elif k < 0.25:
            c = f"\\E e \\in s : e # {r.randrange(K)}"
        elif k < 0.3:
            c = f"\\E e \\in 0..1 : e = x \\/ e = y"
        elif k < 0.33:
            c = f"\\E e \\in 0..x : e # y"
        if r.random() < 0.3:
            junct = r.choice(["/\\", "\\/"])
            c = f"{c} {junct} {self.expr(local)} # {r.randrange(K)}"
        return c

    def assign(self, free, local, ind):
        """one assignment to a variable not yet assigned in this step (free is updated)"""
        r = self.r
        targets = [v for v in ("x", "y", "a1", "a2", "as", "t", "m", "f") if v in free and (v != "t" or local)]
        if not targets:
            return ind + "skip;"
        v = r.choice(targets)
        if v in ("a1", "a2", "as"):
            for q in ("a1", "a2", "as"):
                free.discard(q)   # the whole function counts as one variable per step
            idx = {"a1": "1", "a2": "2", "as": "self"}[v]
            return f"{ind}a[{idx}] := {self.expr(local)};"
        free.discard(v)
        if v == "m":
            return ind + r.choice(['m := "a";', 'm := "b";', 'm := IF x = 0 THEN "a" ELSE "b";'])
        if v == "f":
            return ind + r.choice(["f := ~f;", "f := TRUE;", f"f := {self.expr(local)} < {self.expr(local)};", 'f := (m = "a") \\/ f;'])
        if v in ("x", "y") and r.random() < 0.15:
            return f"{ind}bump({v});"                     # macro call: v := Bump(v)
        if v in ("x", "y") and r.random() < 0.2:
            other = "y" if v == "x" else "x"
            if other in free:
                free.discard(other)
                return f"{ind}{v} := {self.expr(local)} || {other} := {self.expr(local)};"
        return f"{ind}{v} := {self.expr(local)};"

    def simple_block(self, free, local, ind, depth=0):
        """1-2 label-free statements"""
        r, out = self.r, []
        for _ in range(r.randint(1, 2)):
            k = r.random()
            if k < 0.45 or depth >= 2:
                out.append(self.assign(free, local, ind))
            elif k < 0.6:
                f1, f2 = set(free), set(free)
                out.append(f"{ind}if {self.cond(local)} then")
                out.append(self.simple_block(f1, local, ind + "  ", depth + 1))
                if r.random() < 0.5:
                    out.append(f"{ind}elsif {self.cond(local)} then")
                    out.append(self.simple_block(f2, local, ind + "  ", depth + 1))
                    f3 = set(free)
                    out.append(f"{ind}else")
                    out.append(self.simple_block(f3, local, ind + "  ", depth + 1))
                    free &= f1 & f2 & f3
                else:
                    out.append(f"{ind}else")
                    out.append(self.simple_block(f2, local, ind + "  ", depth + 1))
                    free &= f1 & f2
                out.append(f"{ind}end if;")
            elif k < 0.72:
                f1, f2 = set(free), set(free)
                out.append(f"{ind}either")
                out.append(self.simple_block(f1, local, ind + "  ", depth + 1))
                out.append(f"{ind}or")
                out.append(self.simple_block(f2, local, ind + "  ", depth + 1))
                out.append(f"{ind}end either;")
                free &= f1 & f2
            elif k < 0.82:
                f1 = set(free)
                dom = r.choice(["0..1", "{0, 2}", "1..2", "0..x", "y..2", "1..Len(q)"])
                if r.random() < 0.2:
                    out.append(f"{ind}with w = {self.expr(local)} do")
                else:
                    out.append(f"{ind}with w \\in {dom} do")
                body = self.simple_block(f1, local, ind + "  ", depth + 1)
                out.append(body.replace("self", "w", 1) if r.random() < 0.5 else body)
                out.append(f"{ind}end with;")
                free &= f1
            elif k < 0.86 and "q" in free:
                free.discard("q")
                if r.random() < 0.5:
                    out.append(f"{ind}if Len(q) < 2 then q := Append(q, {self.expr(local)}); end if;")
                else:
                    tgt = r.choice([v for v in ("x", "y") if v in free] or ["q"])
                    if tgt == "q":
                        out.append(f"{ind}if q # <<>> then q := Tail(q); end if;")
                    else:
                        free.discard(tgt)
                        out.append(f"{ind}if q # <<>> then {tgt} := Head(q); q := Tail(q); end if;")
            elif k < 0.9 and "s" in free:
                free.discard("s")
                op = r.choice(["\\cup", "\\"])
                out.append(f"{ind}s := s {op} {{{self.expr(local)}}};")
            elif k < 0.93:
                out.append(f"{ind}await {self.cond(local)};")
            elif k < 0.95:
                out.append(f"{ind}assert {self.expr(local)} < {K};")
            else:
                out.append(f"{ind}skip;")
        return "\n".join(out)

    def process_body(self, local):
        r = self.r
        labels = [self.label() for _ in range(r.randint(2, 4))]
        out = []
        for i, lab in enumerate(labels):
            free = {"x", "y", "a1", "a2", "as", "t", "q", "s", "m", "f"}
            k = r.random()
            if k < 0.1 and i + 1 < len(labels):    # an either with a label inside one branch
                inner = self.label()
                f1, f2 = set(free), set(free)
                out.append(f"  {lab}: either")
                out.append(self.simple_block(f1, local, "      "))
                out.append(f"    {inner}: " + self.assign(set(free), local, "").strip())
                out.append("  or")
                out.append(self.simple_block(f2, local, "      "))
                out.append("  end either;")
            elif k < 0.2 and i + 1 < len(labels):    # a while loop on a bounded counter (t), body with its own label
                inner = self.label()
                out.append(f"  {lab}: while {'t' if local else 'x'} < {K - 1} do")
                out.append(f"    {inner}: {'t := t + 1' if local else 'x := x + 1'};")
                f2 = {"y", "a1", "a2", "as", "q", "s", "m", "f"}
                out.append(self.simple_block(f2, local, "      "))
                out.append("  end while;")
            elif k < 0.4 and i + 1 < len(labels):  # an if with a label / goto inside: the next statement is labeled
                f1, f2 = set(free), set(free)
                inner = self.label()
                out.append(f"  {lab}: if {self.cond(local)} then")
                out.append(self.simple_block(f1, local, "      "))
                out.append(f"    {inner}: " + self.assign(set(free), local, "").strip())
                out.append("  else")
                if r.random() < 0.5:
                    out.append(self.simple_block(f2, local, "      "))
                    out.append(f"      goto {r.choice(labels[i + 1:])};")
                else:
                    out.append(self.simple_block(f2, local, "      "))
                out.append("  end if;")
            else:
                body = self.simple_block(free, local, "      ")
                out.append(f"  {lab}:\n{body}")
        return "\n".join(out)

    def module(self, name):
        r = self.r
        procs = []
        local = r.random() < 0.7
        procs.append(f"process P \\in 1..2\n" + ("  variables t = 0;\n" if local else "") + "begin\n" + self.process_body(local) + "\nend process")
        if r.random() < 0.5:
            body = self.process_body(False).replace("self", "0")
            procs.append("process Q = 0\nbegin\n" + body + "\nend process")
        alg = (f"variables x = 0, y \\in 0..1, a = [i \\in 0..2 |-> i % {K}], q = <<>>, s = {{}}, m \\in {{\"a\", \"b\"}}, f = FALSE;\n"
               f"define\n  Sum == (x + y) % {K}\n  Bump(v) == (v + 1) % {K}\n  Mix(u, v) == IF u < v THEN Bump(u) ELSE v\nend define;\n"
               f"macro bump(v) begin v := Bump(v); end macro;\n\n" + "\n\n".join(procs))
        return (f"---- MODULE {name} ----\nEXTENDS Naturals, Sequences, FiniteSets, TLC\n\n(* --algorithm {name}\n{alg}\n\nend algorithm *)\n\n"
                f"Small == x < {K} /\\ y < {K} /\\ (\\A i \\in 0..2 : a[i] < {K}) /\\ Len(q) <= 2 /\\ (\\A e \\in s : e < {K})\n====\n")


@pytest.mark.parametrize("block", range(6))
def test_random_algorithms_compiled_vs_evaluated(block):
    checked = 0
    for seed in range(block * 20, block * 20 + 20):
        text = Gen(seed).module(f"rnd{seed}")
        try:
            prog = helpers.ShimProgram(text, ["Small"], {})
        except RuntimeError as e:
            if "too many alternatives" in str(e):   # a documented capacity limit of the compiled path (254 slots per state)
                continue
            with pytest.raises(RuntimeError):       # the generator broke a PlusCal rule (e.g. a needed label): both routes refuse
                helpers.pcal_translate(text)
            assert "label" in str(e) or "assignment" in str(e), (seed, str(e), text)
            continue
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        try:
            r = helpers.shim_run("pcal", prog.params, dump=dump, check_deadlock=False)
            o = Checker(prog.translated()).run_levels(invariants=["Small"], check_deadlock=False)
            for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len"):
                assert r[k] == o[k], (seed, k, r[k], o[k], text)
            states = helpers.read_dump(dump)
            assert [states[l + 1] for l in range(len(states))] == o["states"], (seed, text)
            # third route: the product's host evaluator (tla_rust_amd/csrc/tlaeval.cpp, through the test door) on the translation
            with tempfile.TemporaryDirectory() as d:
                (Path(d) / f"rnd{seed}.tla").write_text(prog.translated())
                (Path(d) / "m.cfg").write_text("SPECIFICATION Spec\nINVARIANT Small\n")
                e = helpers.tlaeval_run(Path(d) / f"rnd{seed}.tla", Path(d) / "m.cfg", deadlock=False)
            assert e["rc"] == 0, (seed, e)
            for k in ("distinct", "generated", "queue_left", "depth", "trace_len"):
                assert r[k] == e[k], (seed, k, r[k], e[k], text)
            assert e["verdict"] == {"ok": 0, "invariant": 1, "assert": 2, "deadlock": 3, "spec-error": 4}[r["verdict"]], (seed, e, r["verdict"])
            checked += 1
        finally:
            os.unlink(dump)
            prog.close()
    assert checked >= 10


@pytest.mark.parametrize("block", range(2))
def test_random_algorithms_under_a_constraint(block):
    """the same two routes under cfg CONSTRAINT Tight (states outside are generated and invariant-checked, not stored)"""
    checked = pruned = 0
    for seed in range(300 + block * 20, 300 + block * 20 + 20):
        text = Gen(seed).module(f"rnd{seed}").replace("\n====\n", f"\nTight == x + y < {K} - 1 /\\ Len(q) <= 1 /\\ Cardinality(s) <= 1\n====\n")
        try:
            prog = helpers.ShimProgram(text, ["Small"], {}, constraints=["Tight"])
        except RuntimeError:
            continue            # refusals are the subject of the test above
        try:
            r = helpers.shim_run("pcal", prog.params, check_deadlock=False)
            o = Checker(prog.translated()).run_levels(invariants=["Small"], check_deadlock=False, constraints=["Tight"])
            free = Checker(prog.translated()).run_levels(invariants=["Small"], check_deadlock=False)
            for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
                assert r[k] == o[k], (seed, k, r[k], o[k], text)
            checked += 1
            pruned += o["distinct"] < free["distinct"]
        finally:
            prog.close()
    assert checked >= 10 and pruned >= 5


def test_random_algorithms_with_an_uninitialised_variable():
    """`variables u, ...` (defaultInitValue) threaded through random algorithms: an extra process assigns u, then reads it"""
    checked = 0
    for seed in range(400, 420):
        text = Gen(seed).module(f"rnd{seed}")
        text = text.replace("variables x = 0,", "variables u, x = 0,", 1)
        text = text.replace("\n\nend algorithm *)", f"\n\nprocess R = 7\nbegin\n  R0: u := x;\n  R1: u := (u + 1) % {K};\nend process\n\nend algorithm *)", 1)
        assert "process R = 7" in text and "variables u, x" in text
        try:
            prog = helpers.ShimProgram(text, ["Small"], {})
        except RuntimeError:
            continue
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        try:
            r = helpers.shim_run("pcal", prog.params, dump=dump, check_deadlock=False)
            o = Checker(prog.translated()).run_levels(invariants=["Small"], check_deadlock=False)
            for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
                assert r[k] == o[k], (seed, k, r[k], o[k], text)
            states = helpers.read_dump(dump)
            assert [states[l + 1] for l in range(len(states))] == o["states"], (seed, text)
            assert "u = defaultInitValue" in states[1][0]
            checked += 1
        finally:
            os.unlink(dump)
            prog.close()
    assert checked >= 8
