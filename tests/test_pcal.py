"""PlusCal front-end (tla_rust_amd/csrc/pcal.cpp, pcal_compile.cpp, spec_vm.h) on the CPU:

* the translator (`mc --transpile` = the reference's `pcal2tla`, Makefile:3-4) is pinned by evaluating its
  OUTPUT with oracle/tla_eval.py (a TLC-like evaluator of TLA+ text) against the TLC run the reference
  publishes (README.md:267-321: 9097 / 6164 / 999, depth 7, the 6-state trace, the action positions);
* the compiled program (the bytecode every GPU lane interprets) is run by the host build of the same
  interpreter (tests/_shim) and compared with that evaluator: counters, per-level state SETS, verdicts;
* and with the hand lowerings of the two root specs (spec_pluscal.h).
The GPU leg of the same comparisons is tests/test_gpu_pcal.py."""
import json
import os
import sys
import tempfile
from pathlib import Path

import pytest

import helpers

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
from tla_eval import Checker  # noqa: E402

SPECS = ROOT / "specs"
# (module file, invariants, integer constants)
CASES = [
    (SPECS / "pcal_intro.tla", ["MoneyInvariant"], {}),                      # reference pcal_intro.tla:4-23 + .cfg:3
    (SPECS / "readme_variant" / "pcal_intro.tla", [], {}),                   # README.md:220-243 (labels A:, B:)
    (SPECS / "readme_variant" / "pcal_intro.tla", ["MoneyInvariant"], {}),
    (SPECS / "atomic_add.tla", [], {}),                                      # reference atomic_add.tla:4-23
    (SPECS / "pluscal" / "peterson.tla", ["MutualExclusion", "TurnInRange"], {}),
    (SPECS / "pluscal" / "cas_counter.tla", ["NeverTooMany", "SeenIsOld"], {"Workers": 2, "N": 2}),
    (SPECS / "pluscal" / "cas_counter.tla", ["NeverTooMany"], {"Workers": 3, "N": 1}),
    (SPECS / "pluscal" / "lost_update.tla", [], {}),
    (SPECS / "pluscal" / "euclid.tla", ["Positive"], {"M": 12}),
    (SPECS / "pluscal" / "ticket_lock.tla", ["Mutex", "Fifo"], {"P": 3, "Rounds": 1}),      # define block + macro
    (SPECS / "pluscal" / "ticket_lock.tla", ["Mutex", "Fifo"], {"P": 2, "Rounds": 2}),
    (SPECS / "pluscal" / "treiber_stack.tla", ["PoppedOnce", "TopIsNode", "Conservation"], {"N": 2}),   # CAS loops, pointers
    (SPECS / "pluscal" / "peterson_c.tla", ["MutualExclusion", "TurnInRange"], {}),                     # c-syntax
    (SPECS / "pluscal" / "csyntax_mix.tla", ["Inv"], {"N": 2}),   # c-syntax: define, macro, else-if, goto in if, either, with
    (SPECS / "pluscal" / "bounded_queue.tla", ["Bounded", "Fifo"], {"Items": 4, "MaxQ": 2, "Consumers": 1}),   # sequences
    (SPECS / "pluscal" / "bounded_queue.tla", ["Bounded", "Fifo"], {"Items": 3, "MaxQ": 2, "Consumers": 2}),   # ... assert fails
    (SPECS / "pluscal" / "wait_set.tla", ["Disjoint", "HolderNotWaiting", "Counted"], {"N": 3}),               # set variables
    (SPECS / "pluscal" / "swap.tla", [], {}),                                                                  # a := e || b := f
    (SPECS / "pluscal" / "scratch_locals.tla", ["AtMostN"], {"N": 2}),      # `variable tmp;`: defaultInitValue
    (SPECS / "pluscal" / "scratch_locals.tla", ["AtMostN"], {"N": 3}),
    # PROCEDURES (round 4; expanded into the calling processes, tla_rust_amd/csrc/pcal.h): call / return, parameters, procedure variables
    (SPECS / "pluscal" / "proc_demo.tla", [], {}),
    (SPECS / "pluscal" / "treiber_procs.tla", ["PopsDistinct"], {"N": 2}),      # the lock-free stack of the reference's roadmap, with procedures
    (SPECS / "pluscal" / "treiber_procs.tla", ["PopsDistinct"], {"N": 3}),
    (SPECS / "pluscal" / "proc_nested.tla", ["XBound", "Final"], {}),           # c-syntax; a procedure calling another; two kinds of processes
    # RECURSION (round 5; one copy of the body per process + a bounded call stack per procedure kept as plain variables, pcal.cpp call_recursive)
    (SPECS / "pluscal" / "recursive_sum.tla", ["Bounded"], {"N": 3}),           # four frames deep in two interleaving processes
    # (N = 4 needs a fifth frame: MC_EOVERFLOW — test_recursion_deeper_than_the_stack_fails_at_the_call_and_mutual_recursion)
    (SPECS / "pluscal" / "even_odd.tla", ["Answered"], {"N": 4}),               # mutual recursion: two stacks, return sites in each other's body
    # RECORDS (round 4; kept field by field, tla_rust_amd/csrc/pcal.h): r.f, r[i].f, r := [f |-> ..], r = s, r[i] := [..], records as process locals
    (SPECS / "pluscal" / "treiber_records.tla", ["PoppedOnce", "TopIsNode", "NextIsNode", "OldIsNode"], {"N": 2}),   # versioned head, nodes as records
    (SPECS / "pluscal" / "treiber_records.tla", ["PoppedOnce", "TopIsNode", "NextIsNode", "OldIsNode"], {"N": 3}),
    (SPECS / "pluscal" / "ring_buffer.tla", ["Fifo", "FullHasItem", "EmptyIsClean"], {"K": 3, "Items": 5, "Torn": False}),   # SPSC ring of record slots
    (SPECS / "pluscal" / "ring_buffer.tla", ["Fifo"], {"K": 2, "Items": 3, "Torn": True}),                                    # ... flag before item: assert fails
    # the Michael-Scott queue: the lock-free LIST of the reference's roadmap (README.md:26-42): two pointers moved by CAS, a lagging tail that is helped
    # forward, ghost tickets at the linearisation points; with a plain store instead of the linking CAS a node is lost and Fifo breaks (18-state trace)
    (SPECS / "pluscal" / "ms_queue.tla", ["PointersAreNodes", "TailLagsByOne", "NeverEmpty", "DequeuedOnce", "Fifo", "Conservation"], {"N": 2, "Racy": False}),
    (SPECS / "pluscal" / "ms_queue.tla", ["PointersAreNodes", "TailLagsByOne", "NeverEmpty", "DequeuedOnce", "Fifo", "Conservation"], {"N": 2, "Racy": True}),
    # NESTED records (round 5; one level per pass, pcal.cpp RecordFlattener): the Michael-Scott queue as published — every pointer a (ptr, count)
    # record inside the queue / node records, nodes freed and reused; Counted = FALSE compares the ptr halves only: ABA, Head swings to a freed node
    (SPECS / "pluscal" / "ms_queue_counted.tla", ["HeadLive", "TailLive", "PointersAreNodes", "TailAtMostOneBehind", "CountsGrow"], {"N": 2, "K": 3, "Counted": True}),
    (SPECS / "pluscal" / "ms_queue_counted.tla", ["HeadLive", "TailLive", "PointersAreNodes", "TailAtMostOneBehind", "CountsGrow"], {"N": 2, "K": 3, "Counted": False}),
    # CHANNELS (round 5, last part): ARRAYS of sequences (spec_vm.h VM_SEQSEL) and SEQUENCES of RECORDS (one sequence per field, pcal.cpp RecordFlattener):
    # two-phase commit over FIFO channels of [type, from] messages; Eager = TRUE commits on the first yes vote: Consistent breaks
    (SPECS / "pluscal" / "two_phase_channels.tla", ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator", "AtMostTwoWaiting"], {"RM": 2, "Eager": False}),
    (SPECS / "pluscal" / "two_phase_channels.tla", ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator", "AtMostTwoWaiting"], {"RM": 3, "Eager": False}),
    (SPECS / "pluscal" / "two_phase_channels.tla", ["Consistent", "InboxHoldsVotes", "FromTheCoordinator"], {"RM": 3, "Eager": True}),
    # ... q \o <<r>>, q[k] := r, an initial element, `||` over two sequences, an array of sequences of numbers beside the record channels
    (SPECS / "pluscal" / "mailboxes.tla", ["LogOk", "Pongs", "HeardTheLeft"], {"N": 2}),
    # SETS of RECORDS (a message soup: msgs := msgs \cup {[type |-> "prepared", rm |-> self]}, with m \in msgs, r \in msgs; spec_vm.h VM_RSADD): two-phase
    # commit after Lamport's TwoPhase; Hasty = TRUE commits on the first "prepared": Consistent breaks
    (SPECS / "pluscal" / "two_phase_soup.tla", ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"], {"RM": 2, "Hasty": False}),
    (SPECS / "pluscal" / "two_phase_soup.tla", ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"], {"RM": 3, "Hasty": False}),
    (SPECS / "pluscal" / "two_phase_soup.tla", ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages"], {"RM": 3, "Hasty": True}),
    # a procedure with a RECORD PARAMETER (call deposit(mine), call deposit([who |-> self, amount |-> 2])) and `with old = biggest` (a record bound field by field)
    (SPECS / "pluscal" / "record_args.tla", ["Sane"], {"N": 2}),
    (SPECS / "pluscal" / "record_args.tla", ["Sane"], {"N": 3}),
    # epoch-based reclamation (the lock-free epoch-based GC of the roadmap): pin / read / exchange / retire / advance / free; one epoch of grace instead
    # of two frees a node a pinned reader still holds
    (SPECS / "pluscal" / "epoch_gc.tla", ["HeadIsLive", "NoDanglingReader", "EpochInRange"], {"N": 2, "Grace": 2}),
    (SPECS / "pluscal" / "epoch_gc.tla", ["HeadIsLive", "NoDanglingReader", "EpochInRange"], {"N": 2, "Grace": 1}),
    # the roadmap's lock-free IO buffer: reserve / seal with one CAS on a header RECORD, release, the last writer out flushes; flushing a sealed buffer
    # while a writer still copies (Patient = FALSE) lets its bytes land after the flush (assert)
    (SPECS / "pluscal" / "io_buffer.tla", ["HeaderInRange", "SealedIsFull", "FlushedFull"], {"N": 2, "Cap": 1, "Patient": True}),
    (SPECS / "pluscal" / "io_buffer.tla", ["HeaderInRange", "SealedIsFull", "FlushedFull"], {"N": 3, "Cap": 2, "Patient": True}),
    (SPECS / "pluscal" / "io_buffer.tla", ["HeaderInRange", "SealedIsFull", "FlushedFull"], {"N": 3, "Cap": 2, "Patient": False}),
    # the roadmap's lock-free radix tree: a missing child is installed by CAS, the loser frees its node; with a plain store a subtree is unlinked
    (SPECS / "pluscal" / "radix_tree.tla", ["InsertedKeysAreFound", "NoLeak", "ChildrenAreNodes"], {"N": 2, "Plain": False}),
    (SPECS / "pluscal" / "radix_tree.tla", ["InsertedKeysAreFound", "ChildrenAreNodes"], {"N": 2, "Plain": True}),
    # the roadmap's lock-free pagecache: deltas linked onto a page's chain by CAS, a consolidation installed by CAS from the head it read; installed
    # blindly it loses the deltas linked meanwhile
    (SPECS / "pluscal" / "pagecache.tla", ["Conservation", "HeadIsAllocated"], {"N": 2, "Blind": False}),
    (SPECS / "pluscal" / "pagecache.tla", ["Conservation", "HeadIsAllocated"], {"N": 2, "Blind": True}),
    # the reference's own PlusCal example: FastMutex, examples/p-manual.pdf Figure 2 p.13 (translation walked through in App. B)
    (SPECS / "pluscal" / "fast_mutex.tla", ["MutualExclusion"], {"N": 2}),
    (SPECS / "pluscal" / "fast_mutex.tla", ["MutualExclusion"], {"N": 3}),
    # the manual's first example: Euclid's algorithm WITHOUT labels (p-manual sections 2.1-2.3): labels Lbl_1, Lbl_2 are added
    (SPECS / "pluscal" / "euclid_manual.tla", ["ResultIsGcd"], {"N": 4}),
    (SPECS / "pluscal" / "euclid_manual.tla", ["ResultIsGcd"], {"N": 30}),
    # p-manual section 2.4: `assert v = gcd(24, v_ini)` with the manual's CHOOSE definition of gcd before the translation
    (SPECS / "pluscal" / "euclid_assert.tla", [], {"N": 4}),
    (SPECS / "pluscal" / "euclid_assert.tla", [], {"N": 40}),
]


CHANNEL_STEMS = {"two_phase_channels", "mailboxes", "two_phase_soup", "record_args", "epoch_gc", "io_buffer", "radix_tree", "pagecache"}   # their GPU cases: tests/test_gpu_zz_channels.py


def strip_translation(text):
    """the module as its author wrote it: without the \\* BEGIN/END TRANSLATION block"""
    a = text.find("\\* BEGIN TRANSLATION")
    if a < 0:
        return text
    b = text.index("\\* END TRANSLATION", a)
    b = text.index("\n", b) + 1
    return text[:a] + text[b:]


def test_ms_queue_three_threads_compiled():
    """the Michael-Scott queue with three threads through the compiled program on the host: 91 727 states / 228 229 generated / depth 40,
    every invariant holds — the numbers oracle/tla_eval.py gives for the translation (61 s; the two-thread cases above run both)"""
    invs = ["PointersAreNodes", "TailLagsByOne", "NeverEmpty", "DequeuedOnce", "Fifo", "Conservation"]
    prog = helpers.ShimProgram((SPECS / "pluscal" / "ms_queue.tla").read_text(), invs, {"N": 3, "Racy": False})
    r = helpers.shim_run("pcal", prog.params)
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["queue_left"]) == (91727, 228229, 40, "ok", 0)
    if os.environ.get("TLAMC_SLOW"):
        o = Checker(prog.translated(), constants={"N": 3, "Racy": False}).run_levels(invariants=invs)
        assert (o["distinct"], o["generated"], o["depth"], o["verdict"]) == (91727, 228229, 40, "ok") and r["levels"] == o["levels"]


def test_translation_is_idempotent_and_matches_the_committed_specs():
    """specs/*.tla carry the translator's own output: transpiling the stripped source gives the file back"""
    for path in {c[0] for c in CASES} | {SPECS / "atomic_add_n.tla"}:
        text = path.read_text()
        if "BEGIN TRANSLATION" not in text:
            continue
        block = lambda t: t[t.index("\\* BEGIN TRANSLATION"):t.index("\\* END TRANSLATION")]  # noqa: E731
        assert block(helpers.pcal_translate(strip_translation(text))) == block(text), path
        assert helpers.pcal_translate(text) == text, path


def test_readme_golden_through_the_translator():
    """README.md:267-321: the TLC run of the README variant, reproduced by evaluating OUR translation"""
    src = strip_translation((SPECS / "readme_variant" / "pcal_intro.tla").read_text())
    tla = helpers.pcal_translate(src)
    r = Checker(tla).run()
    assert (r["generated"], r["distinct"], r["queue_left"], r["depth"]) == (9097, 6164, 999, 7)      # README.md:319-320
    assert r["verdict"] == "assert" and r["message"] == "Failure of assertion at line 16, column 4."  # README.md:269
    golden = json.loads((ROOT / "tests" / "golden" / "readme_pcal_intro_trace.json").read_text())
    assert len(r["trace"]) == 6
    got = [dict(l[3:].split(" = ", 1) for l in s.splitlines()) for s in r["trace"]]
    assert got == golden["states"]                                                                    # README.md:272-311
    # the action positions TLC printed for this layout (README.md:278,285,292,299): line of each definition
    lines = tla.splitlines()
    where = {name: next(i + 1 for i, l in enumerate(lines) if l.startswith(name + "(self) ==")) for name in ("Transfer", "A", "B", "C")}
    assert where == {"Transfer": 35, "A": 42, "B": 47, "C": 52}
    assert lines[39].rstrip() == " " * 34 + "money >>" and len(lines[39].rstrip()) == 42                # "line 40, col 42"
    assert len(lines[44].rstrip()) == 63 and len(lines[49].rstrip()) == 65                            # A: 45:63, B: 50:65
    assert lines[53].rstrip().endswith('"Failure of assertion at line 16, column 4.")') and len(lines[53].rstrip()) == 66


def test_committed_specs_through_the_translator():
    r = Checker(helpers.pcal_translate(Path("/root/reference/pcal_intro.tla").read_text() if Path("/root/reference").exists()
                                       else strip_translation((SPECS / "pcal_intro.tla").read_text()))).run(invariants=["MoneyInvariant"])
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (3800, 5850, 5, "ok")
    r = Checker(helpers.pcal_translate(strip_translation((SPECS / "atomic_add.tla").read_text()))).run()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (5, 7, 4, "ok")                # 2^N + 1, N 2^(N-1) + 3


def test_reference_files_translate_untouched():
    """the reference commits its two root specs UNtranslated; they must go through as they are"""
    ref = Path("/root/reference")
    if not ref.exists():
        pytest.skip("reference tree not present")
    for name in ("pcal_intro.tla", "atomic_add.tla"):
        out = helpers.pcal_translate((ref / name).read_text())
        assert "\\* BEGIN TRANSLATION" in out and "\\* END TRANSLATION" in out
        assert strip_translation(out) == (ref / name).read_text()


@pytest.mark.parametrize("path,invs,consts", CASES, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_compiled_program_vs_tla_evaluator(path, invs, consts):
    text = path.read_text()
    prog = helpers.ShimProgram(text, invs, consts)
    try:
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        r = helpers.shim_run("pcal", prog.params, dump=dump)
        o = Checker(prog.translated(), constants=consts).run_levels(invariants=invs)
        for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len"):
            assert r[k] == o[k], (k, r[k], o[k])
        assert r["levels"] == o["levels"]
        if r["verdict"] == "invariant":
            assert invs[r["violated_invariant"]] == o["violated"]
        states = helpers.read_dump(dump)
        os.unlink(dump)
        assert len(states) == len(o["states"])
        for lvl, want in enumerate(o["states"], 1):      # bit-exact: the SET of states of every BFS level
            assert states[lvl] == want, f"level {lvl}"
    finally:
        prog.close()


@pytest.mark.parametrize("bound,invs,verdict", [(3, ["NeverAhead"], "ok"), (6, ["NeverAhead"], "ok"), (4, ["NeverAhead", "Small"], "invariant")])
def test_constraint_bounds_an_infinite_algorithm(bound, invs, verdict):
    """cfg CONSTRAINT for compiled programs (FIFO/MCInnerFIFO.cfg:23-26, p-manual section 4.3 p.36): growing_counters has an infinite state
    space; under CONSTRAINT Small the states outside are generated and invariant-checked but neither stored nor expanded.  An
    INVARIANT that only fails outside the constraint IS reported (TLC checks a successor before it filters it)."""
    text = (SPECS / "pluscal" / "growing_counters.tla").read_text()
    consts = {"Bound": bound}
    prog = helpers.ShimProgram(text, invs, consts, constraints=["Small"])
    try:
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        r = helpers.shim_run("pcal", prog.params, dump=dump)
        o = Checker(prog.translated(), constants=consts).run_levels(invariants=invs, constraints=["Small"])
        for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
            assert r[k] == o[k], (k, r[k], o[k])
        assert r["verdict"] == verdict
        if verdict == "invariant":
            assert invs[r["violated_invariant"]] == o["violated"] == "Small" and r["trace_len"] == bound + 2
        else:
            assert r["generated"] > r["distinct"] > 2 * bound
        states = helpers.read_dump(dump)
        os.unlink(dump)
        for lvl, want in enumerate(o["states"], 1):
            assert states[lvl] == want, f"level {lvl}"
        assert all("produced = %d" % (bound + 1) not in t for lvl in states.values() for t in lvl) or verdict == "invariant"
    finally:
        prog.close()


def test_constraint_must_be_a_definition():
    text = (SPECS / "pluscal" / "growing_counters.tla").read_text()
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(text, [], {"Bound": 2}, constraints=["Tiny"])
    assert "CONSTRAINT Tiny" in str(e.value)


def test_compiled_program_vs_hand_lowering():
    """the two root specs have both a hand lowering (spec_pluscal.h) and a compiled program: same graph"""
    for path, hand, invs in [(SPECS / "pcal_intro.tla", ("pcal_intro", [0, 1, 20, 2]), ["MoneyInvariant"]),
                             (SPECS / "readme_variant" / "pcal_intro.tla", ("pcal_intro", [1, 0, 20, 2]), []),
                             (SPECS / "atomic_add.tla", ("atomic_add", [2]), [])]:
        prog = helpers.ShimProgram(path.read_text(), invs)
        a = helpers.shim_run("pcal", prog.params)
        b = helpers.shim_run(*hand)
        for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
            assert a[k] == b[k], (path.name, k)
        prog.close()


MODULE = "---- MODULE t ----\nEXTENDS Naturals\n(* --algorithm t\n%s\nend algorithm *)\n====\n"


@pytest.mark.parametrize("body,needle", [
    ("variables x = 0;\nprocedure p() begin L: skip; end procedure;\nbegin\nA: skip;", "run off the end of procedure p"),
    ("variables x = 0;\nmacro m(a) begin a := 1; end macro;\nbegin\nA: m(x + 1);", "must be instantiated with a variable"),
    ("variables x = 0;\nbegin\nA: x := 1; x := 2;", "second assignment to x"),
    ("variables x = 0;\nbegin\nA: x := 1 || x := 2;", "two assignments to x"),
    ("variables x = 0;\nbegin\nA: if x = 0 then B: x := 1; end if; x := 2;", "needs a label"),
    ("variables x = 0;\nbegin\nA: y := 1;", "undeclared variable y"),
    ("variables x = 0;\nbegin\nA: while x < 2 do x := x + 1; end while; B: call f();", "no such procedure"),
])
def test_refusals_are_explained(body, needle):
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(MODULE % body)
    assert needle in str(e.value)


def test_fast_mutex_translation_follows_the_manual_appendix_b():
    """examples/p-manual.pdf App. B pp.60-62 walks through the translation of FastMutex (Figure 2 p.13): the declarations, Init and
    the actions it prints (ncs, start, l1, l2, l4, l7, l8) come out of the translator conjunct by conjunct (the manual's rendering
    drops the `/\\ TRUE` of a skip and line breaks)."""
    text = (SPECS / "pluscal" / "fast_mutex.tla").read_text()
    tr = " ".join(helpers.pcal_translate(strip_translation(text)).split())
    for piece in [
        "\\* BEGIN TRANSLATION CONSTANT defaultInitValue VARIABLES x, y, b, pc, j vars == << x, y, b, pc, j >> ProcSet == (1..N)",
        "/\\ x = defaultInitValue /\\ y = 0 /\\ b = [i \\in 1..N |-> FALSE]",
        '/\\ j = [self \\in 1..N |-> defaultInitValue] /\\ pc = [self \\in ProcSet |-> "ncs"]',
        'ncs(self) == /\\ pc[self] = "ncs" /\\ TRUE /\\ pc\' = [pc EXCEPT ![self] = "start"] /\\ UNCHANGED << x, y, b, j >>',
        'start(self) == /\\ pc[self] = "start" /\\ b\' = [b EXCEPT ![self] = TRUE] /\\ pc\' = [pc EXCEPT ![self] = "l1"] /\\ UNCHANGED << x, y, j >>',
        'l1(self) == /\\ pc[self] = "l1" /\\ x\' = self /\\ pc\' = [pc EXCEPT ![self] = "l2"] /\\ UNCHANGED << y, b, j >>',
        'l2(self) == /\\ pc[self] = "l2" /\\ IF y # 0 THEN /\\ pc\' = [pc EXCEPT ![self] = "l3"] ELSE /\\ pc\' = [pc EXCEPT ![self] = "l5"] '
        '/\\ UNCHANGED << x, y, b, j >>',
        'l4(self) == /\\ pc[self] = "l4" /\\ y = 0 /\\ pc\' = [pc EXCEPT ![self] = "start"] /\\ UNCHANGED << x, y, b, j >>',
        'l7(self) == /\\ pc[self] = "l7" /\\ b\' = [b EXCEPT ![self] = FALSE] /\\ j\' = [j EXCEPT ![self] = 1] /\\ pc\' = [pc EXCEPT ![self] = "l8"] '
        '/\\ UNCHANGED << x, y >>',
        'l8(self) == /\\ pc[self] = "l8" /\\ IF j[self] <= N THEN /\\ ~b[j[self]] /\\ j\' = [j EXCEPT ![self] = j[self] + 1] '
        '/\\ pc\' = [pc EXCEPT ![self] = "l8"] ELSE /\\ pc\' = [pc EXCEPT ![self] = "l9"]',
    ]:
        assert piece in tr, piece


def test_euclid_of_the_manual_known_answer():
    """examples/p-manual.pdf p.10: model checking EuclidAlg with N = 4 prints <<24, 4, "have gcd", 4>>, <<24, 3, "have gcd", 3>>,
    <<24, 2, "have gcd", 2>>, <<24, 1, "have gcd", 1>>.  The algorithm has no labels: the front-end adds Lbl_1 (the while) and
    Lbl_2 (the second assignment to u), as the manual says its translator does (p.9)."""
    text = (SPECS / "pluscal" / "euclid_manual.tla").read_text()
    tr = helpers.pcal_translate(strip_translation(text))
    assert 'pc = "Lbl_1"' in tr and "Lbl_2 == /\\ pc = \"Lbl_2\"\n         /\\ u' = u - v" in tr and "Lbl_3" not in tr
    prog = helpers.ShimProgram(text, ["ResultIsGcd"], {"N": 4})
    fd, dump = tempfile.mkstemp()
    os.close(fd)
    r = helpers.shim_run("pcal", prog.params, dump=dump)
    states = helpers.read_dump(dump)
    os.unlink(dump)
    prog.close()
    assert r["verdict"] == "ok" and r["levels"][0] == 4            # four initial states: v \in 1..4
    done = sorted(t for lvl in states.values() for t in lvl if 'pc = "Done"' in t)
    assert done == ['/\\ u = 0 /\\ v = %d /\\ v_ini = %d /\\ pc = "Done"' % (k, k) for k in (1, 2, 3, 4)]   # gcd(24, k) = k


def test_manual_gcd_assertion_and_bounded_choose():
    """examples/p-manual.pdf section 2.4 pp.10-11: the assertion `v = gcd(24, v_ini)` holds with the manual's definition of gcd (a
    bounded CHOOSE, evaluated like TLC: first satisfying element in ascending order); a wrong definition trips it, and a CHOOSE
    that no element satisfies is an evaluation error, not a silent value."""
    text = (SPECS / "pluscal" / "euclid_assert.tla").read_text()
    assert text.index("gcd(x, y) ==") < text.index("\\* BEGIN TRANSLATION")          # "before the BEGIN TRANSLATION line" (p.10)
    prog = helpers.ShimProgram(text, [], {"N": 24})
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "ok"
    prog.close()
    prog = helpers.ShimProgram(text.replace("=> i >= j", "=> i <= j"), [], {"N": 4})     # the SMALLEST common divisor: wrong
    r = helpers.shim_run("pcal", prog.params)
    o = Checker(prog.translated(), constants={"N": 4}).run_levels()
    assert (r["verdict"], r["trace_len"]) == ("assert", o["trace_len"]) and o["verdict"] == "assert"
    prog.close()
    prog = helpers.ShimProgram(text.replace("=> i >= j", "=> i > j"), [], {"N": 4})      # nothing is greater than itself: no witness
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "spec-error"
    prog.close()
    # (CHOOSE over a set VARIABLE was refused until round 5's last part: the smallest member that satisfies the predicate, like TLC)
    r = _vm_equals_evaluator(MODULE % "variables s = {1, 2, 5}, x = 0;\nbegin\nA: x := CHOOSE i \\in s : i > 1;\nB: s := s \\ {x};\nC: x := CHOOSE i \\in s : i > 1;")
    assert r["verdict"] == "ok" and r["distinct"] == 4


def test_uninitialised_variables_translate_to_defaultInitValue():
    """`variable x;` (p-manual section 3.3): CONSTANT defaultInitValue + `x = defaultInitValue` in Init; the compiled program
    prints the model value bare like TLC, and READING a variable that still holds it is an evaluation error, not garbage."""
    text = (SPECS / "pluscal" / "scratch_locals.tla").read_text()
    tr = helpers.pcal_translate(strip_translation(text))
    assert "\\* BEGIN TRANSLATION\nCONSTANT defaultInitValue\nVARIABLES cell, winner, spare, pc, tmp, seen" in tr
    assert "/\\ winner = defaultInitValue" in tr and "/\\ tmp = [self \\in 1..N |-> defaultInitValue]" in tr
    prog = helpers.ShimProgram(text, ["AtMostN"], {"N": 2})
    fd, dump = tempfile.mkstemp()
    os.close(fd)
    r = helpers.shim_run("pcal", prog.params, dump=dump)
    states = helpers.read_dump(dump)
    os.unlink(dump)
    prog.close()
    assert r["verdict"] == "ok"
    assert states[1] == ['/\\ cell = 0 /\\ winner = defaultInitValue /\\ spare = defaultInitValue /\\ pc = <<"R", "R">> '
                         '/\\ tmp = <<defaultInitValue, defaultInitValue>> /\\ seen = <<defaultInitValue, defaultInitValue>>']
    assert all("spare = defaultInitValue" in t for lvl in states.values() for t in lvl)          # never assigned
    assert any("seen = <<FALSE, TRUE>>" in t or "seen = <<TRUE, FALSE>>" in t for lvl in states.values() for t in lvl)   # typed by its first assignment
    early = MODULE % "variables x, y = 0;\nbegin\nA: y := x + 1;"
    prog = helpers.ShimProgram(early)
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "spec-error"
    prog.close()
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(MODULE % "variables x;\nbegin\nA: if x = defaultInitValue then x := 1; end if;")
    assert "defaultInitValue can only be" in str(e.value)
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(MODULE % "variables x;\nbegin\nA: x := <<1>>;")
    assert "without an initial value" in str(e.value)


def test_sequence_longer_than_its_cells_is_an_error_not_a_truncation():
    text = (SPECS / "pluscal" / "bounded_queue.tla").read_text()
    prog = helpers.ShimProgram(text, ["Bounded"], {"Items": 9, "MaxQ": 9, "Consumers": 1})
    with pytest.raises(RuntimeError) as e:
        helpers.shim_run("pcal", prog.params)
    assert "-3" in str(e.value)      # MC_EOVERFLOW
    prog.close()


def test_c_syntax_translates_like_p_syntax():
    block = lambda t: t[t.index("\\* BEGIN TRANSLATION"):t.index("\\* END TRANSLATION")]  # noqa: E731
    a = helpers.pcal_translate(strip_translation((SPECS / "pluscal" / "peterson.tla").read_text()))
    b = helpers.pcal_translate(strip_translation((SPECS / "pluscal" / "peterson_c.tla").read_text()))
    assert block(a) == block(b)


def test_either_with_while_goto_translation_shape():
    """shape of the translation of the constructs the root specs do not use (p-manual App. B)"""
    tla = helpers.pcal_translate(strip_translation((SPECS / "pluscal" / "lost_update.tla").read_text()))
    assert "/\\ \\/ /\\ mode' = \"add\"" in tla and "\\E d \\in {2, 5}:" in tla
    assert 'pc = [self \\in ProcSet |-> CASE self \\in 1..2 -> "Pick"' in tla and '[] self = 3 -> "Final"]' in tla
    tla = helpers.pcal_translate(strip_translation((SPECS / "pluscal" / "cas_counter.tla").read_text()))
    assert "ELSE /\\ pc' = [pc EXCEPT ![self] = \"Read\"]" in tla          # goto inside if
    assert "/\\ IF done[self] < N" in tla                                     # while = IF on the loop label
    tla = helpers.pcal_translate(strip_translation((SPECS / "pluscal" / "euclid.tla").read_text()))
    assert "/\\ pc = \"Loop\"" in tla and "(pc = \"Done\" /\\ UNCHANGED vars)" in tla   # uniprocess


def test_wide_state_128_cells():
    """more than 64 scalar cells: atomic_add_n with 70 adders (72 cells) — the 128-cell instantiation of the interpreter"""
    text = (SPECS / "atomic_add_n.tla").read_text()
    prog = helpers.ShimProgram(text, [], {"N": 70})
    fd, dump = tempfile.mkstemp()
    os.close(fd)
    r = helpers.shim_run("pcal", prog.params, max_distinct=2000, dump=dump)
    o = Checker(prog.translated(), constants={"N": 70}).run_levels(max_distinct=2000)
    assert r["levels"] == o["levels"] == [1, 70, 2415]
    assert (r["distinct"], r["generated"]) == (o["distinct"], o["generated"])
    states = helpers.read_dump(dump)
    os.unlink(dump)
    assert [states[l + 1] for l in range(3)] == o["states"]
    prog.close()


def test_integer_overflow_and_division_like_tlc():
    """TLC's integers are 32-bit and an overflow is an error, not a wrap; `\\div` rounds towards minus infinity
    for a divisor of either sign; division by zero is an error"""
    prog = helpers.ShimProgram(MODULE % "variables x = 2147483000, y = 0;\nbegin\nA: x := x + 1000;\nB: y := 1;")
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "spec-error"                 # 2^31 - 648 + 1000 overflows
    prog.close()
    prog = helpers.ShimProgram(MODULE % "variables x = 65536, y = 0;\nbegin\nA: y := x * x;")
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "spec-error"
    prog.close()
    text = MODULE % "variables a = 7, b = 0 - 2, q = 0, r = 0;\nbegin\nA: q := a \\div b;\nB: r := (0 - 7) \\div 2;\nC: assert q = 0 - 4 /\\ r = 0 - 4;"
    prog = helpers.ShimProgram(text)
    r = helpers.shim_run("pcal", prog.params)
    o = Checker(prog.translated()).run_levels()
    assert r["verdict"] == o["verdict"] == "ok" and r["distinct"] == o["distinct"]
    prog.close()
    prog = helpers.ShimProgram(MODULE % "variables a = 7, b = 0, q = 0;\nbegin\nA: q := a \\div b;")
    assert helpers.shim_run("pcal", prog.params)["verdict"] == "spec-error"
    prog.close()


# ---------------------------------------------------------------------------------------------- procedures (round 4)
PROC_FIXTURES = ROOT / "tests" / "golden" / "pcal_procedures"


@pytest.mark.parametrize("spec,fixture,consts,invs", [("proc_demo", "ProcDemoStack", {}, []), ("treiber_procs", "TreiberStack", {"N": 2}, ["PopsDistinct"]),
                                                      # a RECORD parameter: pcal2tla's record-valued `req` (defaultInitValue, restored from the frame) against fields req_who, req_amount
                                                      ("record_args", "RecordArgsStack", {"N": 2}, ["Sane"])])
def test_procedure_expansion_equals_the_stack_translation(spec, fixture, consts, invs):
    """PlusCal procedures are EXPANDED into the calling processes (tla_rust_amd/csrc/pcal.h) instead of being translated with a `stack`
    variable as pcal2tla does (p-manual section 3.5).  For non-recursive procedures the two are the same state graph: the hand-written
    stack translation of each spec (tests/golden/pcal_procedures/*.tla: frames, Head / Tail, restore on return), evaluated by the general
    TLA+ evaluator, against the expansion — evaluated as text AND compiled — counters, depth, verdict, per-level counts."""
    import tlaplus as T
    c = T.Checker(PROC_FIXTURES / f"{fixture}.tla", cfg_path=PROC_FIXTURES / f"{fixture}.cfg", search=[])
    p = c.run_levels(keep_states=False)
    text = (SPECS / "pluscal" / f"{spec}.tla").read_text()
    o = Checker(helpers.pcal_translate(text), constants=consts).run_levels(invariants=invs)
    prog = helpers.ShimProgram(text, invs, consts)
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (p["distinct"], p["generated"], p["depth"], p["verdict"], p["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (p["distinct"], p["generated"], p["depth"], p["verdict"], p["levels"])
    assert p["distinct"] > 100


def test_recursive_procedure_equals_the_stack_translation():
    """round 5 (missing 4): a RECURSIVE procedure is compiled with one copy of its body per process and a bounded call
    stack kept as plain variables (depth counter, return-site codes, one slot per variable and level; $TLAMC_PCAL_STACK levels, default
    4).  Same state graph as pcal2tla's `stack` of frames: the hand-written stack translation of specs/pluscal/recursive_sum.tla
    (tests/golden/pcal_recursion/RecursiveSumStack.tla, N = 3: four frames deep), evaluated by the general TLA+ evaluator, against the
    product's translation — evaluated as text AND compiled — counters, depth, verdict, per-level counts."""
    import tlaplus as T
    fx = ROOT / "tests" / "golden" / "pcal_recursion"
    p = T.Checker(fx / "RecursiveSumStack.tla", cfg_path=fx / "RecursiveSumStack.cfg", search=[]).run_levels(keep_states=False)
    text = (SPECS / "pluscal" / "recursive_sum.tla").read_text()
    o = Checker(helpers.pcal_translate(text), constants={"N": 3}).run_levels(invariants=["Bounded"])
    prog = helpers.ShimProgram(text, ["Bounded"], {"N": 3})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (p["distinct"], p["generated"], p["depth"], p["verdict"], p["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (p["distinct"], p["generated"], p["depth"], p["verdict"], p["levels"])
    assert p["verdict"] == "ok" and p["distinct"] > 200


def test_recursion_deeper_than_the_stack_fails_at_the_call_and_mutual_recursion():
    """N = 4 needs a fifth frame: the run is not cut short silently — the check in front of the push fails at the depth where pcal2tla's
    unbounded stack would have had five frames.  ADVICE round 5: that is a CAPACITY limit of the translation, not an assertion of the
    algorithm — the compiled program reports MC_EOVERFLOW (like a sequence that outgrows its cells), and the Assert of the translated text
    says what it is and which variable to raise; with TLAMC_PCAL_STACK = 5 in the translator's environment it passes.  Mutual recursion
    (even / odd): evaluated translation == compiled program, and the answers are right."""
    import os
    text = (SPECS / "pluscal" / "recursive_sum.tla").read_text()
    tr = helpers.pcal_translate(text)
    assert "raise TLAMC_PCAL_STACK (a capacity limit, not an assertion of the algorithm)" in tr and "stack frames this translation reserves for procedure" in tr
    o = Checker(tr, constants={"N": 4}).run_levels(invariants=["Bounded"])
    assert o["verdict"] == "assert"     # (the evaluated TEXT stops at that Assert)
    prog = helpers.ShimProgram(text, ["Bounded"], {"N": 4})
    try:
        with pytest.raises(RuntimeError) as e:
            helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert "-3" in str(e.value)      # MC_EOVERFLOW
    os.environ["TLAMC_PCAL_STACK"] = "5"
    try:
        o5 = Checker(helpers.pcal_translate(text), constants={"N": 4}).run_levels(invariants=["Bounded"])
    finally:
        del os.environ["TLAMC_PCAL_STACK"]
    assert o5["verdict"] == "ok" and o5["distinct"] > o["distinct"]
    text = (SPECS / "pluscal" / "even_odd.tla").read_text()
    os.environ["TLAMC_PCAL_STACK"] = "3"   # (even and odd each hold at most three frames for N + 1 = 5)
    try:
        o = Checker(helpers.pcal_translate(text), constants={"N": 4}).run_levels(invariants=["Answered"])
        prog = helpers.ShimProgram(text, ["Answered"], {"N": 4})
        try:
            r = helpers.shim_run("pcal", prog.params)
        finally:
            prog.close()
    finally:
        del os.environ["TLAMC_PCAL_STACK"]
    assert o["verdict"] == "ok" and (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert o["distinct"] > 100


RECORD_FIXTURES = ROOT / "tests" / "golden" / "pcal_records"


@pytest.mark.parametrize("spec,fixture,cfg,consts,invs", [
    ("treiber_records", "TreiberRecords", "TreiberRecords", {"N": 2}, ["PoppedOnce", "TopIsNode", "NextIsNode", "OldIsNode"]),
    ("ring_buffer", "RingBuffer", "RingBuffer", {"K": 3, "Items": 5, "Torn": False}, ["Fifo", "FullHasItem", "EmptyIsClean"]),
    ("ring_buffer", "RingBuffer", "RingBufferTorn", {"K": 3, "Items": 5, "Torn": True}, ["Fifo"]),
    # nested records: Q.Head.ptr, mem[i].next.count — pcal2tla's EXCEPT !.Head, ![i].next.ptr against three passes of flattening
    ("ms_queue_counted", "MsQueueCounted", "MsQueueCounted", {"N": 2, "K": 3, "Counted": True}, ["HeadLive", "TailLive", "PointersAreNodes", "TailAtMostOneBehind", "CountsGrow"]),
    ("ms_queue_counted", "MsQueueCounted", "MsQueueUncounted", {"N": 2, "K": 3, "Counted": False}, ["HeadLive", "TailLive", "PointersAreNodes", "TailAtMostOneBehind", "CountsGrow"]),
    # sequences of records: pcal2tla's `chan' = [chan EXCEPT ![p] = Append(chan[p], [type |-> ..])]`, `msg' = Head(chan[0])` against one sequence per field
    ("two_phase_channels", "TwoPhaseChannels", "TwoPhaseChannels", {"RM": 3, "Eager": False}, ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator", "AtMostTwoWaiting"]),
    ("two_phase_channels", "TwoPhaseChannels", "TwoPhaseChannelsEager", {"RM": 3, "Eager": True}, ["Consistent", "InboxHoldsVotes", "FromTheCoordinator"]),
    ("mailboxes", "Mailboxes", "Mailboxes", {"N": 2}, ["LogOk", "Pongs", "HeardTheLeft"]),
    ("mailboxes", "Mailboxes", "Mailboxes3", {"N": 3}, ["LogOk", "Pongs", "HeardTheLeft"]),      # (deadlock: a node that has its pong leaves without answering)
    # round 6: the model of the driver line's `pcal` object, its golden no longer the product's alone
    ("pagecache", "PageCache", "PageCache", {"N": 2, "Blind": False}, ["Conservation", "HeadIsAllocated"]),
    ("pagecache", "PageCache", "PageCacheBlind", {"N": 2, "Blind": True}, ["Conservation", "HeadIsAllocated"]),   # (the blind store loses a delta: Conservation breaks)
    # ... and the other four specs whose large goldens were product-made (the same fixtures at the large sizes: tests/golden/pcal_oracle.json)
    ("epoch_gc", "EpochGc", "EpochGc", {"N": 2, "Grace": 2}, ["HeadIsLive", "NoDanglingReader", "EpochInRange"]),
    ("epoch_gc", "EpochGc", "EpochGcOneGrace", {"N": 2, "Grace": 1}, ["HeadIsLive", "NoDanglingReader", "EpochInRange"]),   # (one grace period: a pinned reader's node is freed)
    ("io_buffer", "IoBuffer", "IoBuffer", {"N": 2, "Cap": 2, "Patient": True}, ["HeaderInRange", "SealedIsFull", "FlushedFull"]),
    ("io_buffer", "IoBuffer", "IoBufferHasty", {"N": 3, "Cap": 2, "Patient": False}, ["HeaderInRange", "SealedIsFull", "FlushedFull"]),   # (flushed under a writer: the assert in Copy)
    ("radix_tree", "RadixTree", "RadixTree", {"N": 2, "Plain": False}, ["InsertedKeysAreFound", "NoLeak", "ChildrenAreNodes"]),
    ("radix_tree", "RadixTree", "RadixTreePlain", {"N": 3, "Plain": True}, ["InsertedKeysAreFound", "ChildrenAreNodes"]),   # (a plain store unlinks the first thread's subtree)
    ("two_phase_soup", "TwoPhaseSoup", "TwoPhaseSoup", {"RM": 3, "Hasty": False}, ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"]),
    ("two_phase_soup", "TwoPhaseSoup", "TwoPhaseSoupHasty", {"RM": 3, "Hasty": True}, ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages"]),
])
def test_records_field_by_field_equal_the_record_valued_translation(spec, fixture, cfg, consts, invs):
    """PlusCal record variables are kept FIELD BY FIELD (tla_rust_amd/csrc/pcal.h, RECORDS) instead of as one record-valued variable
    as pcal2tla keeps them.  The two are the same state graph: the hand-written record translation of each spec
    (tests/golden/pcal_records/*.tla: EXCEPT ![i].f, whole-record assignment and comparison), evaluated by the general TLA+
    evaluator, against the product's translation — evaluated as text, its invariants reading the records through the derived
    definitions `r == [f |-> r_f, ...]`, AND compiled — counters, depth, verdict, per-level counts."""
    import tlaplus as T
    c = T.Checker(RECORD_FIXTURES / f"{fixture}.tla", cfg_path=RECORD_FIXTURES / f"{cfg}.cfg", search=[])
    p = c.run_levels(keep_states=False)
    text = (SPECS / "pluscal" / f"{spec}.tla").read_text()
    o = Checker(helpers.pcal_translate(text), constants=consts).run_levels(invariants=invs)
    if cfg == "Mailboxes3":   # three mailboxes of three-field records + two more sequence variables: 4 cells per sequence instead of 8 fit the 128 cells of a state
        os.environ["TLAMC_PCAL_SEQ"] = "4"
    try:
        prog = helpers.ShimProgram(text, invs, consts)
    finally:
        os.environ.pop("TLAMC_PCAL_SEQ", None)
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert p["verdict"] == o["verdict"]
    if p["verdict"] == "ok":   # (an error stops the general evaluator at the state, the other two at the end of the level)
        assert (p["distinct"], p["generated"], p["depth"], p["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
        assert p["distinct"] > 50


REC_HEAD = "---- MODULE M ----\nEXTENDS Naturals\n(* --algorithm M\nvariables r = [a |-> 0, b |-> FALSE], q = [a |-> 1, b |-> TRUE], arr = [i \\in 1..2 |-> [a |-> 0, b |-> FALSE]], x = 0;\n"
REC_ERRORS = [
    ("begin L: x := r.c; end algorithm *)\n====\n", "record r has no field c"),
    ("begin L: r.c := 1; end algorithm *)\n====\n", "record r has no field c"),
    ("begin L: x.a := 1; end algorithm *)\n====\n", "x is not a record variable"),
    ("begin L: x := r; end algorithm *)\n====\n", "a record is assigned to x"),
    ("begin L: r := 3; end algorithm *)\n====\n", "must be a record constructor, a record variable or an element of a record array"),
    ("begin L: r := [a |-> 1]; end algorithm *)\n====\n", "does not have its fields"),
    ("begin L: x := Foo(r); end algorithm *)\n====\n", "used as a whole value"),
    ("begin L: arr := r; end algorithm *)\n====\n", "assignment to the whole record array arr"),
    ("begin L: r[1] := q; end algorithm *)\n====\n", "r is a record, not an array of records"),
    ("begin L: x := x.a; end algorithm *)\n====\n", "field access is supported on record variables"),
    ("begin L: if r = 3 then skip; end if; end algorithm *)\n====\n", "a record can only be compared with"),
    ("begin L: r.a := 1; r.b := TRUE; end algorithm *)\n====\n", "second assignment to r in one step"),
    ("begin L: r.a := 1 || r.a := 2; end algorithm *)\n====\n", "two assignments to r"),
]


@pytest.mark.parametrize("body,msg", REC_ERRORS, ids=[m[:24] for _, m in REC_ERRORS])
def test_record_errors_are_refused_with_a_message(body, msg):
    with pytest.raises(RuntimeError) as e:
        helpers.pcal_translate(REC_HEAD + body)
    assert msg in str(e.value), str(e.value)


NEST_HEAD = "---- MODULE M ----\nEXTENDS Naturals\n(* --algorithm M\nvariables r = [a |-> 0, b |-> FALSE], x = 0, n = [k |-> 0, inner |-> [p |-> 0, q |-> 0]];\n"
NEST_ERRORS = [
    # nested records (round 5)
    ("begin L: r.a.z := 1; end algorithm *)\n====\n", "r.a is not a record"),
    ("begin L: n.inner.c := 1; end algorithm *)\n====\n", "record n_inner has no field c"),
    ("begin L: x := n.inner.c; end algorithm *)\n====\n", "the record has no field c"),
    ("begin L: n.inner := 3; end algorithm *)\n====\n", "the value assigned to n.inner must be a record constructor"),
    ("begin L: n.inner := r; end algorithm *)\n====\n", "the value assigned to n.inner does not have its fields"),
    ("begin L: n.k := n.inner; end algorithm *)\n====\n", "a record is assigned to n.k"),
    ("begin L: x := n.inner; end algorithm *)\n====\n", "a record is assigned to x"),
    ("begin L: n.inner.p := 1 || n.inner := [p |-> 1, q |-> 2]; end algorithm *)\n====\n", "two assignments to n"),
    ("begin L: n.inner.p := 1; n.k := 2; end algorithm *)\n====\n", "second assignment to n in one step"),
    ("begin L: if n.inner = 3 then skip; end if; end algorithm *)\n====\n", "a record can only be compared with"),
    ("begin Next: x := 1; end algorithm *)\n====\n", "the label `Next` has the name of a definition of the translation"),
]


@pytest.mark.parametrize("body,msg", NEST_ERRORS, ids=[m[:28] for _, m in NEST_ERRORS])
def test_nested_record_errors_are_refused_with_a_message(body, msg):
    with pytest.raises(RuntimeError) as e:
        helpers.pcal_translate(NEST_HEAD + body)
    assert msg in str(e.value), str(e.value)


NESTED = r"""---- MODULE M ----
EXTENDS Naturals
(* --algorithm M
variables r = [a |-> 0, f |-> [g |-> 1, h |-> [k |-> 2, l |-> FALSE]]],
          s = [g |-> 5, h |-> [k |-> 7, l |-> TRUE]],
          mem = [i \in 1..2 |-> [val |-> 0, next |-> [ptr |-> 0, ver |-> 0]]],
          x = 0;
begin
 L1: r.f := s;
 L2: r.f.h.k := r.f.g + 1 || r.a := 3;
 L3: if r.f = s then x := 1; end if;
 L4: mem[1].next := [ptr |-> 2, ver |-> mem[1].next.ver + 1];
 L5: mem[2].next.ptr := mem[1].next.ptr || mem[2].val := 9;
 L6: if mem[1].next # mem[2].next then x := r.f.h.k; end if;
 L7: r := [a |-> 1, f |-> [g |-> 2, h |-> s.h]];
 L8: mem[2] := mem[1];
end algorithm *)
Inv == r.f.h.k <= 10 /\ mem[1].next.ver <= 1 /\ (pc = "Done" => mem[2] = mem[1] /\ r.f.h = s.h /\ x = 6)
====
"""


def test_nested_records_translate_level_by_level():
    """NESTED records are flattened one level per pass (pcal.cpp RecordFlattener): r.f.h.k is the variable r_f_h_k; assigning or comparing
    an inner record is the assignment / comparison of its leaves; the translation DEFINES every level (inner before outer) so that
    the text around the algorithm keeps saying r.f.h.k; both routes run the straight-line program to the same single behaviour"""
    tr = helpers.pcal_translate(NESTED)
    for line in ["VARIABLES r_a, r_f_g, r_f_h_k, r_f_h_l, s_g, s_h_k, s_h_l, mem_val, mem_next_ptr, mem_next_ver, x, pc",
                 "r_f_h == [k |-> r_f_h_k, l |-> r_f_h_l]\nr_f == [g |-> r_f_g, h |-> r_f_h]",
                 "mem_next == [i \\in 1..2 |-> [ptr |-> mem_next_ptr[i], ver |-> mem_next_ver[i]]]",
                 "mem == [i \\in 1..2 |-> [val |-> mem_val[i], next |-> mem_next[i]]]",
                 "/\\ r_f_g' = s_g\n      /\\ r_f_h_k' = s_h_k\n      /\\ r_f_h_l' = s_h_l",
                 "/\\ r_f_h_k' = r_f_g + 1\n      /\\ r_a' = 3",
                 "IF (r_f_g = s_g /\\ (r_f_h_k = s_h_k /\\ r_f_h_l = s_h_l))",
                 "/\\ mem_next_ptr' = [mem_next_ptr EXCEPT ![1] = 2]\n      /\\ mem_next_ver' = [mem_next_ver EXCEPT ![1] = mem_next_ver[1] + 1]",
                 "/\\ mem_next_ptr' = [mem_next_ptr EXCEPT ![2] = mem_next_ptr[1]]\n      /\\ mem_val' = [mem_val EXCEPT ![2] = 9]",
                 "/\\ r_a' = 1\n      /\\ r_f_g' = 2\n      /\\ r_f_h_k' = s_h_k\n      /\\ r_f_h_l' = s_h_l"]:
        assert line in tr, (line, tr)
    assert tr.index("r_f_h ==") < tr.index("r_f ==") < tr.index("\nr ==")
    o = Checker(tr).run_levels(invariants=["Inv"])
    prog = helpers.ShimProgram(NESTED, ["Inv"], {})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert o["verdict"] == "ok" and o["distinct"] == 9


def test_records_translate_like_their_fields():
    """what the field-by-field translation writes: whole-record assignment and `||` on two fields are simultaneous assignments, a
    comparison is a conjunction, an element of a record array is read and written through the fields' arrays, and the record itself
    is DEFINED after the variables so that the text around the algorithm can go on saying r.a"""
    tr = helpers.pcal_translate(REC_HEAD + "begin L1: r := q; L2: r.a := 5 || r.b := FALSE; L3: arr[1] := r; L4: if arr[2] # q then x := arr[1].a; end if; end algorithm *)\n====\n")
    for line in ["VARIABLES r_a, r_b, q_a, q_b, arr_a, arr_b, x, pc",
                 "r == [a |-> r_a, b |-> r_b]",
                 "arr == [i \\in 1..2 |-> [a |-> arr_a[i], b |-> arr_b[i]]]",
                 "/\\ r_a' = q_a\n      /\\ r_b' = q_b",
                 "/\\ r_a' = 5\n      /\\ r_b' = FALSE",
                 "/\\ arr_a' = [arr_a EXCEPT ![1] = r_a]\n      /\\ arr_b' = [arr_b EXCEPT ![1] = r_b]",
                 "IF (~(arr_a[2] = q_a /\\ arr_b[2] = q_b))",
                 "x' = arr_a[1]"]:
        assert line in tr, (line, tr)


PROC_WITH_RECORDS = r"""---- MODULE M ----
EXTENDS Naturals
CONSTANT N
(* --algorithm M
variables top = [ptr |-> 0, ver |-> 0], mem = [a \in 1..N |-> [val |-> 0, next |-> 0]], sum = 0;
procedure push(node)
variables old = [ptr |-> 0, ver |-> 0];
begin
  R: old := top;
  L: mem[node].next := old.ptr;
  C: if top = old then top := [ptr |-> node, ver |-> old.ver + 1]; else goto R; end if;
  X: return;
end procedure
process w \in 1..N
begin
  A: mem[self].val := self;
  B: call push(self);
  D: sum := sum + mem[top.ptr].val;
end process
end algorithm *)
Inv == top.ver <= N /\ (top.ptr # 0 => mem[top.ptr].val = top.ptr) /\ \A p \in 1..N : old[p].ver <= N
====
"""


@pytest.mark.parametrize("n,distinct", [(2, 91), (3, 1688)])
def test_a_procedure_with_a_record_variable(n, distinct):
    """procedures are expanded first, records flattened after: a procedure's own record variable becomes the calling processes'
    `old_ptr`, `old_ver`, is reset field by field on `return`, and the invariant reads it as `old[p].ver` through the derived definition"""
    tr = helpers.pcal_translate(PROC_WITH_RECORDS)
    assert "old == [self \\in 1..N |-> [ptr |-> old_ptr[self], ver |-> old_ver[self]]]" in tr
    o = Checker(tr, constants={"N": n}).run_levels(invariants=["Inv"])
    prog = helpers.ShimProgram(PROC_WITH_RECORDS, ["Inv"], {"N": n})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["verdict"], o["levels"])
    assert o["verdict"] == "ok" and o["distinct"] == distinct


PROC_HEAD = "---- MODULE M ----\nEXTENDS Naturals\n(* --algorithm M\nvariables x = 0;\n"
PROC_ERRORS = [
    ("procedure f(a) begin F1: x := a; return; end procedure; begin M1: call f(1); x := 2; end algorithm *)\n====\n", "after a `call` must have a label"),
    ("procedure f(a) begin F1: x := a; return; end procedure; begin M1: call f(1, 2); M2: skip; end algorithm *)\n====\n", "takes 1 arguments"),
    ("procedure f(a) begin x := a; return; end procedure; begin M1: call f(1); M2: skip; end algorithm *)\n====\n", "first statement of procedure f must have a label"),
    ("procedure f(a) begin F1: x := a; end procedure; begin M1: call f(1); M2: skip; end algorithm *)\n====\n", "run off the end of procedure f"),
    ("procedure f(a) begin F1: x := a; return; end procedure; procedure g() begin G1: call f(1); return; end procedure; begin M1: call g(); M2: skip; end algorithm *)\n====\n", "tail call"),
    ("begin M1: x := 1; return; end algorithm *)\n====\n", "`return` outside a procedure"),
    # (pcal2tla evaluates it once when the frame is pushed; the expansion would re-evaluate it at every return)
    ("procedure f(a) variables y = a + 1; begin F1: x := y; return; end procedure; begin M1: call f(1); M2: skip; end algorithm *)\n====\n", "mentions another variable of the procedure"),
    ("begin M1: call nope(1); M2: skip; end algorithm *)\n====\n", "no such procedure"),
    # (one name, one variable: pcal2tla would rename the second `t`; here both back-ends refuse, the translator included)
    ("process a = 1 variables t = 0; begin A1: t := 1; end process; process b = 2 variables t = 0; begin B1: t := 2; end process; end algorithm *)\n====\n", "variable `t` is declared twice"),
    ("process a = 1 variables x = 0; begin A1: x := 1; end process; end algorithm *)\n====\n", "variable `x` is declared twice"),
]


@pytest.mark.parametrize("body,msg", PROC_ERRORS, ids=[m[:18] for _, m in PROC_ERRORS])
def test_procedure_errors_are_refused_with_a_message(body, msg):
    with pytest.raises(RuntimeError) as e:
        helpers.pcal_translate(PROC_HEAD + body)
    assert msg in str(e.value), str(e.value)


def _vm_equals_evaluator(text, invs=(), consts=None):
    """the compiled program on the host VM against oracle/tla_eval.py on the translation: counters, verdict, the SET of states of every level"""
    consts = consts or {}
    prog = helpers.ShimProgram(text, list(invs), consts)
    try:
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        r = helpers.shim_run("pcal", prog.params, dump=dump)
        o = Checker(prog.translated(), constants=consts).run_levels(invariants=list(invs))
        for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
            assert r[k] == o[k], (k, r[k], o[k])
        states = helpers.read_dump(dump)
        os.unlink(dump)
        for lvl, want in enumerate(o["states"], 1):
            assert states[lvl] == want, f"level {lvl}"
        return r
    finally:
        prog.close()


def test_multiple_assignment_with_sequences_reads_the_values_before_the_statement():
    """`a := Tail(a) || x := Head(a) || b := Append(b, Head(a))`: every right-hand side (the scalar operands of the sequence expressions
    included, and the indices of an array of sequences) is evaluated before anything is stored (pcal_compile.cpp seq_operands_to_temps)"""
    text = """---- MODULE par ----
EXTENDS Naturals, Sequences
(* --algorithm par
variables a = <<1, 2, 3>>, b = <<10, 20>>, x = 0, arr = [i \\in 1..2 |-> <<5, 6>>], y = 0;
begin
  L1: a := Tail(a) || x := Head(a) || b := Append(b, Head(a));
  L2: a := <<Head(b), a[1]>> || b := b \\o <<Head(a), x>>;
  L3: arr[1] := Tail(arr[1]) || y := Head(arr[1]);
  L4: arr[y - 3] := Append(arr[y - 3], Len(arr[1]) + Len(arr[2]) * 10) || a[1] := a[2];
  L5: either arr[1] := <<>>; or arr[2] := <<arr[1][1], arr[2][1]>>; end either;
end algorithm *)
====
""".replace("\\\\", "\\")
    r = _vm_equals_evaluator(text)
    assert (r["distinct"], r["verdict"]) == (7, "ok")
    prog = helpers.ShimProgram(text)
    try:
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        helpers.shim_run("pcal", prog.params, dump=dump)
        states = helpers.read_dump(dump)
        os.unlink(dump)
    finally:
        prog.close()
    assert any("a = <<2, 3>>" in s and "x = 1" in s and "b = <<10, 20, 1>>" in s for s in states[2])          # after L1: all three read the OLD a
    assert any("a = <<10, 2>>" in s and "b = <<10, 20, 1, 2, 1>>" in s for s in states[3])                     # after L2
    assert any("arr = <<<<6>>, <<5, 6, 21>>>>" in s and "a = <<2, 2>>" in s for s in states[5])               # after L4: arr[y - 3] = arr[2]


def test_sequence_cells_come_from_the_environment():
    """$TLAMC_PCAL_SEQ cells per sequence (default 8): a longer sequence is MC_EOVERFLOW, never a truncated run"""
    text = """---- MODULE grow ----
EXTENDS Naturals, Sequences
(* --algorithm grow
variables box = [p \\in 1..2 |-> <<>>], n = 0;
begin
  L: while n < 6 do box[1 + (n % 2)] := Append(box[1 + (n % 2)], n); n := n + 1; end while;
end algorithm *)
====
""".replace("\\\\", "\\")
    assert _vm_equals_evaluator(text)["verdict"] == "ok"
    os.environ["TLAMC_PCAL_SEQ"] = "2"
    try:
        prog = helpers.ShimProgram(text)
    finally:
        del os.environ["TLAMC_PCAL_SEQ"]
    try:
        with pytest.raises(RuntimeError) as e:
            helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert "-3" in str(e.value)      # MC_EOVERFLOW


SEQ_HEAD = ("---- MODULE M ----\nEXTENDS Naturals, Sequences\n(* --algorithm M\nvariables q = <<>>, box = [p \\in 1..2 |-> <<>>], ints = [p \\in 1..2 |-> <<>>], "
            "r = [a |-> 0, b |-> FALSE], s = <<>>, x = 0;\n")
SEQ_ERRORS = [
    ("begin L: q := Append(q, [a |-> 1, b |-> TRUE]); x := q; end algorithm *)\n====\n", "the sequence of records q is used as a whole value"),
    ("begin L: q := Append(q, [a |-> 1, b |-> TRUE]); x := Head(q); end algorithm *)\n====\n", "a record is assigned to x"),
    ("begin L: q := Append(q, [a |-> 1]); M: q := Append(q, r); end algorithm *)\n====\n", "the record put into q does not have its fields"),
    ("begin L: q := Append(q, r); q.a := 1; end algorithm *)\n====\n", "q is a sequence of records: assign an element"),
    ("begin L: q := Append(q, r); M: if q = <<r>> then skip; end if; end algorithm *)\n====\n", "can only be compared with <<>>"),
    ("begin L: box[1] := Append(box[1], r); M: box := box; end algorithm *)\n====\n", "assignment to the whole array of sequences box"),
    ("begin L: box[1] := Append(box[1], r); M: box[2] := box[1]; end algorithm *)\n====\n", "can only start from the same element"),
    ("begin L: q := Append(q, [a |-> 1, b |-> [c |-> 1]]); end algorithm *)\n====\n", "can only have plain fields"),
    ("begin L: q := Append(q, r); M: q := Append(q, 3); end algorithm *)\n====\n", "must be a record constructor, a record variable or an element of a sequence / array of records"),
    ("begin L: q := Append(q, r); M: x := q[1].c; end algorithm *)\n====\n", "record q has no field c"),
    ("begin L: ints[1] := Append(ints[1], 3); M: ints[2] := ints[1]; end algorithm *)\n====\n", "can only start from the same element ints[i]"),
    ("begin L: ints[1] := Append(ints[2], 3); end algorithm *)\n====\n", "can only start from the same element ints[i]"),
    ("begin L: ints[1] := Append(ints[1], 3); M: ints := ints; end algorithm *)\n====\n", "assigning the whole array of sequences `ints` is not supported"),
    ("begin L: ints[1] := Append(ints[1], 3); M: x := ints[1]; end algorithm *)\n====\n", "the sequence `ints[..]` is used as a value here"),
    ("begin L: ints[1] := Append(ints[1], 3); M: s := ints[1]; end algorithm *)\n====\n", "copying a sequence into / out of the array `ints` is not supported"),
]


@pytest.mark.parametrize("body,msg", SEQ_ERRORS, ids=[m[:28] for _, m in SEQ_ERRORS])
def test_channel_errors_are_refused_with_a_message(body, msg):
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(SEQ_HEAD.replace("\\\\", "\\") + body)
    assert msg in str(e.value)


@pytest.mark.parametrize("case", ["two_phase_channels_rm4", "two_phase_channels_rm5"])
def test_two_phase_commit_larger_models_equal_the_record_valued_translation(case):
    """4 / 5 resource managers: the compiled program (chan one sequence per FIELD, host VM) against tests/golden/pcal_channels.json — the
    hand-written record-valued translation evaluated by the product's host evaluator tlaeval.cpp (tests/golden/make_pcal_channels_golden.py):
    another text, another engine.  RM = 5 (2 848 539 states, 40 s here) runs under $TLAMC_SLOW; its GPU leg always does (tests/test_gpu_zz_channels.py)"""
    g = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())[case]
    if g["RM"] > 4 and not os.environ.get("TLAMC_SLOW"):
        pytest.skip("40 s of host VM: $TLAMC_SLOW")
    invs = ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator", "AtMostTwoWaiting"]
    os.environ["TLAMC_PCAL_SEQ"] = str(g["seq_cells"])
    try:
        prog = helpers.ShimProgram((SPECS / "pluscal" / "two_phase_channels.tla").read_text(), invs, {"RM": g["RM"], "Eager": False})
    finally:
        del os.environ["TLAMC_PCAL_SEQ"]
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["queue_left"]) == (g["distinct"], g["generated"], g["depth"], "ok", 0)
    assert r["levels"] == g["levels"]


def test_with_over_a_set_of_records_binds_the_element_by_value():
    """`with m \\in msgs do msgs := (msgs \\ {m}) \\cup {...}; r := m; x := m.a ...`: m is the element as it was when the step chose it (its fields are
    copied: removing it moves the others), a chain of set operations reads the set as it was before the statement, `{... Cardinality(msgs) ...}`
    is evaluated before the set is cleared"""
    text = """---- MODULE byval ----
EXTENDS Naturals, FiniteSets
(* --algorithm byval
variables msgs = {[a |-> 2, b |-> TRUE], [a |-> 0, b |-> FALSE], [a |-> 1, b |-> TRUE]}, other = {[a |-> 1, b |-> TRUE]}, r = [a |-> 0, b |-> FALSE], x = 0, y = 0;
begin
  M: with m \\in msgs do
       msgs := (msgs \\ {m}) \\cup {[a |-> m.a + 3, b |-> ~m.b]};
       r := m;
       x := m.a + 10 * Cardinality(msgs);
     end with;
  N: if r \\in msgs \\/ [a |-> x % 10, b |-> r.b] \\notin other then y := 5; end if;
  O: msgs := {[a |-> Cardinality(msgs), b |-> \\E m \\in msgs : m.a > 3 /\\ m.b]};
end algorithm *)
====
""".replace("\\\\", "\\")
    r = _vm_equals_evaluator(text)
    assert (r["distinct"], r["generated"], r["verdict"]) == (10, 13, "ok")


def test_message_soup_equals_the_general_evaluators():
    """specs/pluscal/two_phase_soup.tla: its translation keeps msgs ONE set-valued variable, as pcal2tla does — the module file is the text
    oracle/tlaplus.py (the general evaluator that reads the reference's own specs) and the product's host evaluator tlaeval.cpp walk; the
    compiled program keeps the set as sorted cells (spec_vm.h VM_RSADD).  Counters, depth, per-level counts; RM = 6 / 7 (251 051 / 1 725 467
    states) against tests/golden/pcal_channels.json (tlaeval.cpp; 7 under $TLAMC_SLOW: 53 s of host VM)"""
    import tlaplus as T
    spec, invs = SPECS / "pluscal" / "two_phase_soup.tla", ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"]
    p = T.Checker(spec, cfg_path=SPECS / "pluscal" / "two_phase_soup.cfg", search=[]).run_levels(keep_states=False)
    e = helpers.tlaeval_run(spec, SPECS / "pluscal" / "two_phase_soup.cfg", search=[])
    prog = helpers.ShimProgram(spec.read_text(), invs, {"RM": 3, "Hasty": False})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (p["distinct"], p["generated"], p["depth"], p["verdict"], p["levels"]) == (r["distinct"], r["generated"], r["depth"], "ok", r["levels"]) and r["distinct"] == 827
    assert (e["distinct"], e["generated"], e["depth"], e["verdict"], e["levels"]) == (r["distinct"], r["generated"], r["depth"], 0, r["levels"])
    golden = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())
    for case in ("two_phase_soup_rm6", "two_phase_soup_rm7"):
        g = golden[case]
        if g["RM"] > 6 and not os.environ.get("TLAMC_SLOW"):
            continue
        os.environ["TLAMC_PCAL_SEQ"] = str(g["seq_cells"])
        try:
            prog = helpers.ShimProgram(spec.read_text(), invs, {"RM": g["RM"], "Hasty": False})
        finally:
            del os.environ["TLAMC_PCAL_SEQ"]
        try:
            r = helpers.shim_run("pcal", prog.params)
        finally:
            prog.close()
        assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])


def test_a_set_of_records_that_outgrows_its_cells_overflows():
    text = """---- MODULE grow ----
EXTENDS Naturals
(* --algorithm grow
variables msgs = {}, n = 0;
begin
  L: while n < 4 do msgs := msgs \\cup {[k |-> n, twice |-> 2 * n]}; n := n + 1; end while;
end algorithm *)
====
""".replace("\\\\", "\\")
    assert _vm_equals_evaluator(text)["verdict"] == "ok"
    os.environ["TLAMC_PCAL_SEQ"] = "3"
    try:
        prog = helpers.ShimProgram(text)
    finally:
        del os.environ["TLAMC_PCAL_SEQ"]
    try:
        with pytest.raises(RuntimeError) as e:
            helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert "-3" in str(e.value)      # MC_EOVERFLOW


RSET_HEAD = ("---- MODULE M ----\nEXTENDS Naturals, FiniteSets\n(* --algorithm M\nvariables msgs = {}, other = {[a |-> 1, b |-> TRUE]}, r = [a |-> 0, b |-> FALSE], "
             "s = {}, x = 0;\n")
RSET_ERRORS = [
    ("begin L: msgs := msgs \\cup {[a |-> 1, b |-> TRUE]}; x := msgs; end algorithm *)\n====\n", "the set of records `msgs` is used as a value here"),
    ("begin L: msgs := msgs \\cup {[a |-> 1]}; M: msgs := msgs \\cup {r}; end algorithm *)\n====\n", "does not have the fields of the elements of msgs"),
    ("begin L: msgs := msgs \\cup {r}; M: msgs := msgs \\cup {3}; end algorithm *)\n====\n", "an element of the set of records msgs must be a record constructor"),
    ("begin L: msgs := msgs \\cup {r}; M: msgs := msgs \\cup other; end algorithm *)\n====\n", "a set of records can be assigned {r, ...}"),
    ("begin L: msgs := msgs \\cup {r}; M: if msgs = other then skip; end if; end algorithm *)\n====\n", "comparing two sets of records is not supported"),
    ("begin L: msgs := msgs \\cup {r}; M: if msgs = {r} then skip; end if; end algorithm *)\n====\n", "can only be compared with {}"),
    ("begin L: msgs := msgs \\cup {r}; M: with m \\in msgs do x := m; end with; end algorithm *)\n====\n", "a record is assigned to x"),
    ("begin L: msgs := msgs \\cup {r}; M: with m \\in msgs do x := m.c; end with; end algorithm *)\n====\n", "the record has no field c"),
    ("begin L: msgs := msgs \\cup {[a |-> 1, b |-> [c |-> 1]]}; end algorithm *)\n====\n", "can only have plain fields"),
    ("begin L: msgs := msgs \\cup {r}; M: msgs.a := 1; end algorithm *)\n====\n", "msgs is a set of records: assign the set"),
    ("begin L: msgs := msgs \\cup {r}; M: msgs := other \\cup {r}; end algorithm *)\n====\n", "a set of records can be assigned {r, ...}"),
    ("begin L: msgs := msgs \\cup {r} || x := 1; end algorithm *)\n====\n", "`||` with a set variable is not supported"),
]


@pytest.mark.parametrize("body,msg", RSET_ERRORS, ids=[m[:28] for _, m in RSET_ERRORS])
def test_set_of_records_errors_are_refused_with_a_message(body, msg):
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(RSET_HEAD.replace("\\\\", "\\") + body.replace("\\\\", "\\"))
    assert msg in str(e.value)


def test_uniprocess_algorithm_with_procedure_variables_translates():
    """a uniprocess algorithm has no process identifier: its procedures' parameters / variables are plain variables of the translation
    (the translator dereferenced the missing identifier: a crash until round 5's last part)"""
    text = MODULE % "variables x = 0, r = [a |-> 0, b |-> FALSE];\nprocedure p(q) variables k = 1; begin P1: q.a := q.a + k; P2: x := q.a; return; end procedure;\nbegin\nL: call p(r);\nM: call p([a |-> 7, b |-> TRUE]);\nN: assert x = 8;"
    tr = helpers.pcal_translate(text)
    assert "/\\ q_a = defaultInitValue" in tr and "/\\ k = 1" in tr and "CONSTANT defaultInitValue" in tr
    r = _vm_equals_evaluator(text)
    assert (r["verdict"], r["distinct"]) == ("ok", 8)


@pytest.mark.parametrize("body,needle", [
    ("variables r = [a |-> 0, b |-> FALSE], x = 0;\nprocedure p(q) begin P1: x := q.a; P2: if x < 2 then call p(q); end if; P3: return; end procedure;\nbegin\nL: call p(r);\nM: skip;",
     "parameter q of the RECURSIVE procedure p is passed a record"),
    ("variables r = [a |-> 0, b |-> FALSE], x = 0;\nprocedure p(q = 3) begin P1: x := q; return; end procedure;\nbegin\nL: call p(r);\nM: skip;", "has a default value and is passed a record"),
    ("variables r = [a |-> 0, b |-> FALSE], x = 0;\nprocedure p(q) begin P1: x := q.c; return; end procedure;\nbegin\nL: call p(r);\nM: skip;", "record q has no field c"),
    ("variables r = [a |-> 0, b |-> FALSE], x = 0;\nprocedure p(q) begin P1: x := q.a; return; end procedure;\nbegin\nL: call p(r);\nM: call p(5);\nN: skip;",
     "the value assigned to record variable q must be a record"),
    ("variables r = [a |-> 0, b |-> FALSE], x = 0;\nbegin\nL: with v \\in {r} do x := v.a; end with;", "record variable r is used as a whole value"),
])
def test_record_argument_errors_are_refused_with_a_message(body, needle):
    with pytest.raises(RuntimeError) as e:
        helpers.ShimProgram(MODULE % body)
    assert needle in str(e.value)


def test_a_set_of_records_filled_through_a_record_parameter_is_typed_by_the_arguments():
    """`procedure send(m) ... msgs := msgs \\cup {m}`: the fields of the parameter start as defaultInitValue and say nothing about the types of
    the set's fields; they come from the first real values (`call send([t |-> "a", n |-> 1])`): a number prints as a number"""
    text = (MODULE % "variables msgs = {}, log = <<>>, x = 0;\nprocedure send(m) begin P1: msgs := msgs \\cup {m}; log := Append(log, m); return; end procedure;\nbegin\n"
            "L: call send([t |-> \"a\", n |-> 1]);\nM: call send([t |-> \"b\", n |-> 2]);\nN: with m \\in msgs do x := m.n; end with;").replace("EXTENDS Naturals", "EXTENDS Naturals, Sequences")
    r = _vm_equals_evaluator(text)     # (state TEXTS are compared: `[n |-> 1, t |-> "a"]`, not `[n |-> "M", ...]`)
    assert (r["verdict"], r["distinct"]) == ("ok", 7)


def test_paxos_in_pluscal_over_a_message_soup():
    """specs/pluscal/paxos_soup.tla: single-decree Paxos, one proposer process per ballot, three acceptors, the network a SET of RECORDS with
    five fields; the acceptors never stop (CHECK_DEADLOCK FALSE in the cfg).  The compiled program on the host VM against oracle/tla_eval.py
    on the translation (one value: 7 569 states, state sets included) and against the product's host evaluator tlaeval.cpp reading the
    module + cfg files (two values: 15 993 states / 58 405 generated / depth 25); a proposer that forgets what it was promised (Forgetful)
    lets two values be chosen: Agreement fails at depth 21 on every route"""
    spec = SPECS / "pluscal" / "paxos_soup.tla"
    invs = ["Agreement", "VotesAreProposed", "OneValuePerBallot", "PromisesAreHonest"]
    os.environ["TLAMC_PCAL_SEQ"] = "16"
    try:
        small = helpers.ShimProgram(spec.read_text(), invs, {"NA": 3, "NB": 2, "NV": 1, "Forgetful": False})
        full = helpers.ShimProgram(spec.read_text(), invs, {"NA": 3, "NB": 2, "NV": 2, "Forgetful": False})
        bad = helpers.ShimProgram(spec.read_text(), ["Agreement", "VotesAreProposed", "PromisesAreHonest"], {"NA": 3, "NB": 2, "NV": 2, "Forgetful": True})
    finally:
        del os.environ["TLAMC_PCAL_SEQ"]
    try:
        fd, dump = tempfile.mkstemp()
        os.close(fd)
        r = helpers.shim_run("pcal", small.params, check_deadlock=False, dump=dump)
        o = Checker(small.translated(), constants={"NA": 3, "NB": 2, "NV": 1, "Forgetful": False}).run_levels(invariants=invs, check_deadlock=False)
        for k in ("distinct", "generated", "depth", "verdict", "levels"):
            assert r[k] == o[k], (k, r[k], o[k])
        assert r["distinct"] == 7569
        states = helpers.read_dump(dump)
        os.unlink(dump)
        for lvl, want in enumerate(o["states"], 1):
            assert states[lvl] == want, f"level {lvl}"
        r = helpers.shim_run("pcal", full.params, check_deadlock=False)
        e = helpers.tlaeval_run(spec, SPECS / "pluscal" / "paxos_soup.cfg", search=[])
        assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (15993, 58405, 25, "ok")
        assert (e["distinct"], e["generated"], e["depth"], e["verdict"], e["levels"]) == (r["distinct"], r["generated"], r["depth"], 0, r["levels"])
        r = helpers.shim_run("pcal", bad.params, check_deadlock=False)
        e = helpers.tlaeval_run(spec, SPECS / "pluscal" / "paxos_soup_forgetful.cfg", search=[])
        assert (r["verdict"], r["trace_len"], r["distinct"]) == ("invariant", 21, 16533) and (e["verdict"], e["distinct"], e["levels"]) == (1, 16533, r["levels"])
    finally:
        small.close()
        full.close()
        bad.close()


def test_epoch_based_reclamation_three_threads():
    """specs/pluscal/epoch_gc.tla with three threads: 1 380 120 states / 3 557 242 generated / depth 46, no reader ever holds a freed node — the
    compiled program on the host VM against the product's host evaluator tlaeval.cpp on module + cfg (tests/golden/pcal_channels.json; the
    two-thread model is compared with oracle/tla_eval.py state by state above); with one epoch of grace NoDanglingReader fails at depth 14"""
    g = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["epoch_gc_n3"]
    invs = ["HeadIsLive", "NoDanglingReader", "EpochInRange"]
    text = (SPECS / "pluscal" / "epoch_gc.tla").read_text()
    prog = helpers.ShimProgram(text, invs, {"N": 3, "Grace": 2})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])
    assert r["distinct"] == 1380120
    prog = helpers.ShimProgram(text, invs, {"N": 3, "Grace": 1})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["verdict"], invs[r["violated_invariant"]], r["trace_len"]) == ("invariant", "NoDanglingReader", 14)


def test_io_buffer_four_writers():
    """specs/pluscal/io_buffer.tla with four writers and two slots: 539 320 states / 1 776 249 generated / depth 32 — the compiled program on the
    host VM against tlaeval.cpp on module + cfg (tests/golden/pcal_channels.json); the smaller models are compared with oracle/tla_eval.py
    state by state above"""
    g = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["io_buffer_n4"]
    prog = helpers.ShimProgram((SPECS / "pluscal" / "io_buffer.tla").read_text(), ["HeaderInRange", "SealedIsFull", "FlushedFull"], {"N": 4, "Cap": 2, "Patient": True})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"]) and r["distinct"] == 539320


def test_radix_tree_three_and_four_inserters():
    """specs/pluscal/radix_tree.tla: three inserters 50 361 states, four 3 411 041 states / 12 822 241 generated / depth 17 (the compiled program on
    the host VM against tlaeval.cpp, tests/golden/pcal_channels.json; four under $TLAMC_SLOW: 8 s of host VM after 100 s of evaluator when the
    golden was made); with a plain store instead of the CAS an inserted key is not found (a 14-state behaviour with three inserters)"""
    invs = ["InsertedKeysAreFound", "NoLeak", "ChildrenAreNodes"]
    text = (SPECS / "pluscal" / "radix_tree.tla").read_text()
    e = helpers.tlaeval_run(SPECS / "pluscal" / "radix_tree.tla", SPECS / "pluscal" / "radix_tree.cfg", search=[])
    prog = helpers.ShimProgram(text, invs, {"N": 3, "Plain": False})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (50361, 139105, 14, "ok")
    assert (e["distinct"], e["generated"], e["depth"], e["verdict"], e["levels"]) == (r["distinct"], r["generated"], r["depth"], 0, r["levels"])
    prog = helpers.ShimProgram(text, ["InsertedKeysAreFound", "ChildrenAreNodes"], {"N": 3, "Plain": True})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["verdict"], r["violated_invariant"], r["trace_len"]) == ("invariant", 0, 14)
    if os.environ.get("TLAMC_SLOW"):
        g = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["radix_tree_n4"]
        prog = helpers.ShimProgram(text, invs, {"N": 4, "Plain": False})
        try:
            r = helpers.shim_run("pcal", prog.params)
        finally:
            prog.close()
        assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])


def test_pagecache_three_threads():
    """specs/pluscal/pagecache.tla with three threads: the blind consolidation loses a delta after 14 states (342 457 states explored); the correct
    one has 20 254 597 states / 47 629 297 generated / depth 37 (tests/golden/pcal_channels.json: tlaeval.cpp on module + cfg) — 47 s of host
    VM, under $TLAMC_SLOW here, always on the GPU (tests/test_gpu_zz_channels.py)"""
    invs = ["Conservation", "HeadIsAllocated"]
    text = (SPECS / "pluscal" / "pagecache.tla").read_text()
    prog = helpers.ShimProgram(text, invs, {"N": 3, "Blind": True})
    try:
        r = helpers.shim_run("pcal", prog.params)
    finally:
        prog.close()
    assert (r["verdict"], invs[r["violated_invariant"]], r["trace_len"], r["distinct"]) == ("invariant", "Conservation", 14, 342457)
    if os.environ.get("TLAMC_SLOW"):
        g = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
        prog = helpers.ShimProgram(text, invs, {"N": 3, "Blind": False})
        try:
            r = helpers.shim_run("pcal", prog.params)
        finally:
            prog.close()
        assert (r["distinct"], r["generated"], r["depth"], r["verdict"], r["levels"]) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])


def test_let_in_expressions_and_definitions(tmp_path):
    """`LET a == e  f(x) == g ... IN body` in the algorithm's expressions and in the definitions around it: the front-end substitutes it
    where it is parsed (a definition sees the earlier ones; an unused or guarded one is never evaluated: Head(q) under `IF q = <<>>`); the
    evaluators read the LET as written — same graph, state by state, and a third opinion from tlaeval.cpp"""
    text = """---- MODULE lett ----
EXTENDS Naturals, Sequences
CONSTANTS N
(* --algorithm lett
variables q = <<>>, x = 0, hi = 0;
process P \\in 1..N
begin
  A: x := LET d == x + self
              twice(n) == n + n
              t == twice(d)
          IN  IF t > 6 THEN d ELSE t;
  B: q := Append(q, LET m == x % 3 IN m + 1);
  C: hi := LET best == IF q = <<>> THEN 0 ELSE Head(q) IN IF best > hi THEN best ELSE hi;
end process
end algorithm *)
Bounded == LET top == 3
               sm(k) == IF k <= Len(q) THEN q[k] ELSE 0
           IN  hi <= top /\\ sm(1) <= top /\\ sm(2) <= top /\\ (\\A k \\in 1..Len(q) : LET v == q[k] IN v >= 1)
====
""".replace("\\\\", "\\")
    r = _vm_equals_evaluator(text, ["Bounded"], {"N": 2})
    assert (r["distinct"], r["generated"], r["verdict"]) == (37, 55, "ok")
    tr = helpers.pcal_translate(text)
    assert "x' = (IF ((x + self) + (x + self)) > 6 THEN (x + self) ELSE ((x + self) + (x + self)))" in tr
    (tmp_path / "lett.tla").write_text(tr)
    (tmp_path / "lett.cfg").write_text("SPECIFICATION Spec\nCONSTANT N = 2\nINVARIANT Bounded\n")
    e = helpers.tlaeval_run(tmp_path / "lett.tla", tmp_path / "lett.cfg", search=[])
    assert (e["distinct"], e["generated"], e["verdict"], e["levels"]) == (r["distinct"], r["generated"], 0, r["levels"])


def test_more_of_the_tla_expression_language(tmp_path):
    """CASE ... [] OTHER, DOMAIN of a function variable / of a sequence, {x \\in S : P}, {e : x \\in S}, CHOOSE over a set, <=> / \\equiv and
    `\\in Nat` (the TypeOK conjunct), in the algorithm and in the definitions around it: the compiled program against oracle/tla_eval.py state
    by state, and against the product's host evaluator tlaeval.cpp on the translated module"""
    text = SYN2
    invs = ["TypeOK", "Quorum", "Images", "Sel", "Kind"]
    r = _vm_equals_evaluator(text, invs, {"N": 2})
    assert (r["distinct"], r["generated"], r["verdict"]) == (151, 279, "ok")
    (tmp_path / "syn2.tla").write_text(helpers.pcal_translate(text))
    (tmp_path / "syn2.cfg").write_text("SPECIFICATION Spec\nCONSTANT N = 2\nINVARIANT " + " ".join(invs) + "\n")
    e = helpers.tlaeval_run(tmp_path / "syn2.tla", tmp_path / "syn2.cfg", search=[])
    assert (e["distinct"], e["generated"], e["verdict"], e["levels"]) == (r["distinct"], r["generated"], 0, r["levels"])


SYN2 = r"""---- MODULE syn2 ----
EXTENDS Naturals, Sequences, FiniteSets
CONSTANTS N
(* --algorithm syn2
variables votes = {}, f = [i \in 1..3 |-> 0], q = <<>>, x = 0, small = 0, b = FALSE;
process P \in 1..N
begin
  A: votes := votes \cup {self};
     f[self] := self * 2;
  B: x := CASE Cardinality(votes) = 1 -> 10 [] Cardinality(votes) = 2 -> 20 [] OTHER -> 30;
     q := Append(q, self);
  C: small := CHOOSE v \in votes : \A w \in votes : v <= w;
     b := (x = 20) <=> (Cardinality(votes) = 2);
  D: with i \in DOMAIN q do
       x := x + q[i];
     end with;
  E: x := Cardinality({i \in DOMAIN f : f[i] > 0}) + Cardinality({f[i] \div 2 : i \in votes});
end process
end algorithm *)
TypeOK == x \in Nat /\ small \in Nat /\ (b \equiv b) /\ \A i \in DOMAIN f : f[i] \in Nat
Quorum == 2 * Cardinality({p \in 1..N : pc[p] # "A"}) >= Cardinality(votes)
Images == {f[i] : i \in votes} \subseteq {0, 2, 4, 6}
Sel == (votes # {}) => (CHOOSE v \in votes : TRUE) \in votes
Kind == CASE x < 100 -> TRUE [] OTHER -> FALSE
====
"""


def test_operators_over_whole_variables_in_state_predicates():
    """`Last(z) == z[Len(z)]`, `Sorted(z)`, `Size(t) == Cardinality(t)`, `Sum3(g)`: an operator whose argument is a whole sequence / set / set of
    records / function variable — or an element box[p] of an array of sequences, p the caller's — is SUBSTITUTED into the invariant instead of
    being evaluated to a number (state predicates only: inside an action the argument could name a value assigned earlier in the step)"""
    r = _vm_equals_evaluator(OPS, ["Inv1", "Inv2", "Inv3", "Inv4"], {"N": 2})
    assert (r["distinct"], r["verdict"]) == (13, "ok")
    r = _vm_equals_evaluator(OPS.replace("Last(q) \\in 1..N", "Last(q) \\in 2..N"), ["Inv1"], {"N": 2})
    assert r["verdict"] == "invariant"


OPS = r"""---- MODULE ops ----
EXTENDS Naturals, Sequences, FiniteSets
CONSTANTS N
(* --algorithm ops
variables q = <<>>, box = [i \in 1..2 |-> <<>>], s = {}, f = [i \in 1..3 |-> 0], msgs = {};
process P \in 1..N
begin
  A: q := Append(q, self);
     box[self] := Append(box[self], self * 3);
  B: s := s \cup {self};
     f[self] := self;
     msgs := msgs \cup {[k |-> self, v |-> self + 1]};
end process
end algorithm *)
Last(z) == z[Len(z)]
Sorted(z) == \A i \in 1..Len(z) : \A j \in 1..Len(z) : i < j => z[i] # z[j]
Size(t) == Cardinality(t)
Sum3(g) == g[1] + g[2] + g[3]
Inv1 == (q # <<>>) => Last(q) \in 1..N
Inv2 == Sorted(q) /\ Sorted(box[1]) /\ Sorted(box[2])
Inv3 == Size(s) <= N /\ Size(msgs) <= N /\ Sum3(f) <= 6
Inv4 == \A p \in 1..Len(q) % 3 : (box[p] # <<>>) => Last(box[p]) = p * 3
====
"""


def test_let_substitution_does_not_capture_a_binder_of_the_same_name():
    """`LET f(a) == \\E y \\in {1,2} : y + 1 = a IN \\E y \\in {2,3} : y = x /\\ f(y)` — the argument `y` of the call
    is the OUTER quantifier's variable; substituted textually into f's body it used to land under f's own `\\E y` and mean the inner one
    (ok = FALSE for x = 2, 3, and a translation that re-binds y, which SANY rejects).  The front-end now renames a binder whose variable is
    free in a body substituted below it.  Two opinions that never see the substitution: (1) the invariant states the value the LET must
    have; (2) the same algorithm with the operator in a `define` block (a real call, parameters bound by name) has the same state graph."""
    def module(let):
        expr = ("LET f(a) == \\E y \\in {1, 2} : y + 1 = a IN \\E y \\in {2, 3} : y = x /\\ f(y)" if let
                else "\\E y \\in {2, 3} : y = x /\\ F(y)")
        define = "" if let else "define F(a) == \\E y \\in {1, 2} : y + 1 = a end define;\n"
        return ("---- MODULE cap ----\nEXTENDS Naturals\n(* --algorithm cap\nvariables x = 0, ok = FALSE, done = FALSE;\n" + define +
                "begin\n  A: with v \\in 1..4 do x := v; end with;\n  B: ok := " + expr + ";\n     done := TRUE;\nend algorithm *)\n"
                "Inv == done => (ok <=> x \\in {2, 3})\n====\n")
    a = _vm_equals_evaluator(module(True), ["Inv"])
    b = _vm_equals_evaluator(module(False), ["Inv"])
    assert a["verdict"] == b["verdict"] == "ok"
    assert (a["distinct"], a["generated"], a["levels"]) == (b["distinct"], b["generated"], b["levels"]) and a["distinct"] == 9
    tr = helpers.pcal_translate(module(True))
    tr = tr[tr.index("BEGIN TRANSLATION"):]
    assert "\\E y_1 \\in {1, 2}" in tr and "\\E y \\in {1, 2}" not in tr, tr   # the inner binder is renamed, the outer `y` survives


def test_large_pluscal_goldens_are_pinned_to_the_oracle():
    """the larger PlusCal models' goldens (tests/golden/pcal_channels.json: made by the PRODUCT's host evaluator)
    equal, count for count and level by level, what oracle/tlaplus.py finds on the HAND-WRITTEN pcal2tla-style translations of the same
    algorithms (tests/golden/pcal_oracle.json, tests/golden/make_pcal_oracle_golden.py: 10^5 .. 3.4 x 10^6 states, up to ten minutes of
    evaluator each) — so every comparison of the compiled program (host VM, GPU interpreter, generated code) with pcal_channels.json is a
    comparison with the oracle: another text, another evaluator.  pagecache N = 3 (20 M states) is out of the evaluator's reach and stays
    product-made; its N = 2 is evaluated live above."""
    product = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())
    oracle = json.loads((ROOT / "tests" / "golden" / "pcal_oracle.json").read_text())
    assert set(oracle) == {"two_phase_channels_rm4", "two_phase_soup_rm6", "two_phase_soup_rm7", "epoch_gc_n3", "io_buffer_n4", "radix_tree_n4"}
    for case, o in oracle.items():
        p = product[case]
        assert o["source"].startswith("ORACLE-MADE: oracle/tlaplus.py on the HAND-WRITTEN")
        assert (p["distinct"], p["generated"], p["depth"], p["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"]), case
        assert o["distinct"] >= 90000
