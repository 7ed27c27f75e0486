"""Differential fuzzing of the product's host evaluator (tla_rust_amd/csrc/tlaeval.cpp: `mc X.tla` for modules without a GPU lowering)
against the oracle's general TLA+ evaluator (oracle/tlaplus.py): seeded random modules whose Init / Next / invariant are built from
typed random expressions — integers, sets (enumerations, intervals, comprehensions of both kinds, UNION, SUBSET, DOMAIN), functions
(constructors, EXCEPT with @, application), records (constructors, EXCEPT !.f, field access), sequences (Append / Tail / Head / \\o /
SubSeq / SelectSeq with LAMBDA / Len), quantifiers, bounded CHOOSE, LET, IF, CASE — with nondeterminism by \\E and by disjunction.

Compared: distinct, generated, depth, verdict, per-level counts and the SET of states of every level as printed text (both print TLC's
way).  The two evaluators share a skeleton (the C++ one was written after the Python one), so agreement here
is evidence about the PORT — every operator on every shape of value — not about the semantics; those are pinned by the reference's
own fixtures (tests/test_reference_text_*.py, tests/test_tlaeval.py)."""
import hashlib
import random
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import helpers  # noqa: E402

ORDER = ["n", "s", "f", "r", "q", "c"]


class Gen:
    def __init__(self, seed):
        self.r = random.Random(7000 + seed)
        self.fresh = 0

    def var(self):
        self.fresh += 1
        return f"z{self.fresh}"

    def pick(self, options):
        return self.r.choice(options)()

    # ---- integers in 0..4
    def int_(self, env, d, state=True):
        leaf = [lambda: str(self.r.randrange(5))]
        if env:
            leaf.append(lambda: self.r.choice(env))
        if state:
            leaf += [lambda: "n", lambda: "r.a", lambda: "Len(q)", lambda: f"f[{self.r.randrange(1, 4)}]"]
        if d <= 0:
            return self.pick(leaf)
        a = lambda: self.int_(env, d - 1, state)      # noqa: E731
        b = lambda: self.bool_(env, d - 1, state)     # noqa: E731
        s = lambda: self.set_(env, d - 1, state)      # noqa: E731

        def let():
            z = self.var()
            return f"(LET {z} == {a()} IN ({z} + {self.int_(env + [z], d - 1, state)}) % 5)"

        def choose():
            z = self.var()
            return f"(CHOOSE {z} \\in ({s()} \\cup {{{a()}}}) : TRUE)"

        def choose_p():
            z = self.var()
            return f"(CHOOSE {z} \\in 0..4 : {z} >= {a()})"

        def case():
            return f"(CASE {b()} -> {a()} [] {b()} -> {a()} [] OTHER -> {a()})"
        opts = leaf + [lambda: f"(({a()} + {a()}) % 5)", lambda: f"(({a()} * {a()}) % 5)", lambda: f"(IF {b()} THEN {a()} ELSE {a()})",
                       lambda: f"(Cardinality({s()}) % 5)", lambda: f"({a()} \\div ({a()} + 1))", let, choose, choose_p, case,
                       lambda: f"(IF {a()} > {a()} THEN 1 ELSE 0)"]
        if state:
            opts += [lambda: f"f[({a()} % 3) + 1]", lambda: f"Head(Append(q, {a()}))", lambda: f"Len({self.seq_(env, d - 1)})",
                     lambda: f"{self.rec_(env, d - 1)}.a", lambda: f"{self.fn_(env, d - 1)}[{self.r.randrange(1, 4)}]"]
        return self.pick(opts)

    # ---- sets of integers
    def set_(self, env, d, state=True):
        leaf = [lambda: "{}", lambda: f"{{{self.r.randrange(5)}, {self.r.randrange(5)}}}", lambda: f"0..{self.r.randrange(4)}"]
        if state:
            leaf += [lambda: "s", lambda: "r.b"]
        if d <= 0:
            return self.pick(leaf)
        a = lambda: self.int_(env, d - 1, state)      # noqa: E731
        s = lambda: self.set_(env, d - 1, state)      # noqa: E731

        def filt():
            z = self.var()
            return f"{{{z} \\in {s()} : {self.bool_(env + [z], d - 1, state)}}}"

        def image():
            z = self.var()
            return f"{{{self.int_(env + [z], d - 1, state)} : {z} \\in {s()}}}"

        def big_union():
            z = self.var()
            return f"(UNION {{ {{{z}, ({z} + 1) % 5}} : {z} \\in {s()} }})"

        def subsets():
            z = self.var()
            return f"(UNION {{{z} \\in SUBSET ({s()} \\cap 0..2) : Cardinality({z}) < 2}})"
        opts = leaf + [lambda: f"({s()} \\cup {s()})", lambda: f"({s()} \\cap {s()})", lambda: f"({s()} \\ {s()})", lambda: f"{{{a()}, {a()}}}",
                       lambda: f"({a()}..{a()})", filt, image, big_union, subsets, lambda: f"(IF {self.bool_(env, d - 1, state)} THEN {s()} ELSE {s()})",
                       lambda: f"(UNION {{{s()}, {s()}}})"]
        if state:
            opts += [lambda: f"(DOMAIN {self.fn_(env, d - 1)})", lambda: f"{{f[{self.var_in_domain()}]}}"]
        return self.pick(opts)

    def var_in_domain(self):
        return str(self.r.randrange(1, 4))

    # ---- booleans
    def bool_(self, env, d, state=True):
        a = lambda: self.int_(env, max(d - 1, 0), state)   # noqa: E731
        s = lambda: self.set_(env, max(d - 1, 0), state)   # noqa: E731
        leaf = [lambda: f"{a()} {self.r.choice(['<', '<=', '=', '#', '>', '>='])} {a()}", lambda: f"{a()} \\in {s()}", lambda: self.r.choice(["TRUE", "FALSE"])]
        if d <= 0:
            return self.pick(leaf)
        b = lambda: self.bool_(env, d - 1, state)     # noqa: E731

        def quant():
            z = self.var()
            qf = self.r.choice(["\\A", "\\E"])
            return f"({qf} {z} \\in {s()} : {self.bool_(env + [z], d - 1, state)})"
        opts = leaf + [lambda: f"({b()} /\\ {b()})", lambda: f"({b()} \\/ {b()})", lambda: f"(~{b()})", lambda: f"({b()} => {b()})", lambda: f"({b()} <=> {b()})",
                       lambda: f"({s()} \\subseteq {s()})", lambda: f"({s()} = {s()})", lambda: f"({a()} \\notin {s()})", quant]
        if state:
            opts += [lambda: f"({self.seq_(env, d - 1)} = {self.seq_(env, d - 1)})", lambda: f"({self.rec_(env, d - 1)} = {self.rec_(env, d - 1)})",
                     lambda: f"({self.fn_(env, d - 1)} = {self.fn_(env, d - 1)})", lambda: "(q = <<>>)"]
        return self.pick(opts)

    # ---- functions on 1..3, records [a: int, b: set], sequences of at most 3 integers
    def fn_(self, env, d):
        if d <= 0:
            return "f"
        z = self.var()
        k = self.r.randrange(1, 4)
        return self.pick([lambda: "f", lambda: f"[{z} \\in 1..3 |-> {self.int_(env + [z], d - 1)}]", lambda: f"[f EXCEPT ![{k}] = {self.int_(env, d - 1)}]",
                          lambda: f"[f EXCEPT ![{k}] = (@ + {self.int_(env, d - 1)}) % 5]",
                          lambda: f"[f EXCEPT ![{k}] = {self.int_(env, d - 1)}, ![{k % 3 + 1}] = {self.int_(env, d - 1)}]",
                          lambda: f"(({k} :> {self.int_(env, d - 1)}) @@ f)"])

    def rec_(self, env, d):
        if d <= 0:
            return "r"
        return self.pick([lambda: "r", lambda: f"[a |-> {self.int_(env, d - 1)}, b |-> {self.set_(env, d - 1)}]", lambda: f"[r EXCEPT !.a = {self.int_(env, d - 1)}]",
                          lambda: f"[r EXCEPT !.b = {self.set_(env, d - 1)}]", lambda: f"[r EXCEPT !.a = (@ + 1) % 5, !.b = @ \\cup {{{self.int_(env, d - 1)}}}]"])

    def seq_(self, env, d):
        if d <= 0:
            return "q"
        a = lambda: self.int_(env, d - 1)             # noqa: E731
        z = self.var()
        return self.pick([lambda: "q", lambda: "<<>>", lambda: f"<<{a()}, {a()}>>", lambda: f"(IF Len(q) < 3 THEN Append(q, {a()}) ELSE Tail(q))",
                          lambda: "(IF q # <<>> THEN Tail(q) ELSE q)", lambda: f"(IF Len(q) < 3 THEN q \\o <<{a()}>> ELSE <<>>)",
                          lambda: f"SubSeq(q, 1, IF Len(q) > 1 THEN Len(q) - 1 ELSE Len(q))", lambda: f"SelectSeq(q, LAMBDA {z} : {z} < {a()})",
                          lambda: f"(IF Len(q) < 3 THEN <<{a()}>> \\o q ELSE q)"])

    def module(self):
        d = 2 + (self.r.random() < 0.3)
        init = ["n = " + self.int_([], d, False), "s = " + self.set_([], d, False), "f = [i \\in 1..3 |-> " + self.int_(["i"], 1, False) + "]",
                "r = [a |-> " + self.int_([], 1, False) + ", b |-> " + self.set_([], 1, False) + "]", "q = <<" + self.int_([], 1, False) + ">>", "c = 0"]
        if self.r.random() < 0.4:
            init[0] = "n \\in " + self.set_([], 1, False) + " \\cup {1}"

        def action(k):
            z = self.var()
            parts = ["c < 4", "c' = c + 1", f"n' = {self.int_([], d)}", f"s' = {self.set_([], d)}", f"f' = {self.fn_([], d)}", f"r' = {self.rec_([], d)}",
                     f"q' = {self.seq_([], d)}"]
            if self.r.random() < 0.4:   # nondeterminism: the successor's n ranges over a set
                parts[2] = f"\\E {z} \\in ({self.set_([], 1)} \\cup {{0}}) : n' = ({z} + {self.int_([z], 1)}) % 5"
            if self.r.random() < 0.3:
                parts.insert(0, self.bool_([], 1))                          # an enabling condition
            if self.r.random() < 0.25:
                parts[-1] = "UNCHANGED q"
            if self.r.random() < 0.2:
                parts[4 if len(parts) == 7 else 5] = "UNCHANGED f"
            return f"A{k} == " + "\n      ".join("/\\ " + p for p in parts)
        acts = [action(k) for k in range(self.r.randrange(1, 4))]
        inv = self.bool_([], 2) if self.r.random() < 0.5 else "TRUE"
        text = "---- MODULE Fz ----\nEXTENDS Naturals, Sequences, FiniteSets, TLC\nVARIABLES n, s, f, r, q, c\n"
        text += "Init == " + "\n        ".join("/\\ " + p for p in init) + "\n" + "\n".join(acts) + "\n"
        text += "Next == " + " \\/ ".join(f"A{k}" for k in range(len(acts))) + "\n"
        text += f"Inv == {inv}\nTyped == n \\in 0..4 /\\ c \\in 0..4 /\\ Len(q) <= 3 /\\ r.a \\in 0..4\n====\n"
        return text


def _digests(levels_text):
    return [hashlib.sha256("\n".join(sorted(lvl)).encode()).hexdigest()[:16] for lvl in levels_text]


@pytest.mark.parametrize("seed", range(400))
def test_random_module_product_evaluator_vs_oracle_evaluator(seed, tmp_path):
    import tlaplus as T
    text = Gen(seed).module()
    tla, cfg = tmp_path / "Fz.tla", tmp_path / "Fz.cfg"
    tla.write_text(text)
    cfg.write_text("INIT Init\nNEXT Next\nINVARIANT Inv Typed\n")
    try:
        c = T.Checker(tla, cfg_path=cfg, search=[])
        p = c.run_levels(check_deadlock=False, stop_on_violation=False, keep_states=True)
    except Exception as e:   # the generator made a module the oracle's evaluator rejects (an evaluation error): the product must fail too
        e_text = str(e)
        r = helpers.tlaeval_run(tla, cfg, deadlock=False)
        assert r["rc"] != 0 or r["verdict"] not in (0, 1), (e_text, r, text)
        pytest.skip(f"both refuse: {e_text[:80]}")
    dump = tmp_path / "dump.txt"
    r = helpers.tlaeval_run(tla, cfg, deadlock=False, dump=dump, order=ORDER)
    assert r["rc"] == 0, (r, text)
    want_v = {"ok": 0, "invariant": 1}[p["verdict"]]
    assert r["verdict"] == want_v, (r, p["verdict"], text)
    if p["verdict"] != "ok":
        return          # (an error ends the two searches at different points of the level: the verdict is the comparison)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (p["distinct"], p["generated"], p["depth"], p["levels"]), text
    by_level = helpers.read_dump(str(dump))
    got = _digests([by_level[k] for k in sorted(by_level)])
    want = _digests([[c.spec.state_text(s, ORDER) for s in lvl] for lvl in p["level_states"]])
    assert got == want, text
