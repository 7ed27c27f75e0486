"""GPU leg of tests/test_pcal.py: PlusCal modules compiled by mc_program_compile run on the HIP engine (the
bytecode interpreter of spec_vm.h inside the expand / materialise kernels) and are compared with
oracle/tla_eval.py evaluating the translation of the same module: counters, verdicts, per-level state SETS.
Then the drop-in: `mc X.tla` on PlusCal modules nobody hand-lowered, and `mc -generic` against the hand
lowering of the README variant (byte-identical report)."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
from tla_eval import Checker  # noqa: E402
from test_pcal import CASES, CHANNEL_STEMS, strip_translation  # noqa: E402

pytestmark = pytest.mark.gpu
MC = ROOT / "tla_rust_amd" / "_build" / "mc"


@pytest.fixture(scope="module")
def amd():
    import tla_rust_amd
    assert tla_rust_amd.device_count() >= 1, "no HIP device visible"
    return tla_rust_amd


def cfg_text(invs, consts):
    s = "SPECIFICATION Spec\n"
    if consts:
        s += "CONSTANTS " + " ".join(f"{k} = {str(v).upper() if isinstance(v, bool) else v}" for k, v in consts.items()) + "\n"
    if invs:
        s += "INVARIANTS " + " ".join(invs) + "\n"
    return s


# (the Michael-Scott queue was added after the round's last GPU minute: its GPU cases live in tests/test_gpu_zz_ms_queue.py, which sorts
#  behind every other GPU file — under the driver's `pytest -x` a surprise there cannot keep the rest of the suite from running)
# (... and so do the channel specs of the round's last part: tests/test_gpu_zz_channels.py)
GPU_CASES = [c for c in CASES if not c[0].stem.startswith("ms_queue") and c[0].stem not in CHANNEL_STEMS]


@pytest.mark.parametrize("path,invs,consts", GPU_CASES, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_compiled_program_on_gpu_vs_tla_evaluator(amd, path, invs, consts):
    check_compiled_program_on_gpu(amd, path, invs, consts)


def same_outcome(r, o, prog):
    """the engine's verdict against the evaluator's.  A BFS pass that holds SEVERAL errors (an Assert failing in one state of the level, an
    invariant broken by a successor of another) has no first one on a GPU: the engine reports the error with the smallest (arena index,
    slot), and the order of a level's states in the arena depends on which workgroup got there first — as TLC's report does with several
    workers.  Counters, depth and per-level state sets do not depend on it (the whole level is expanded); the reported error must be ONE OF
    those the evaluator finds in that pass (oracle/tla_eval.py run_levels: `errors`)."""
    if o["verdict"] == "ok":
        assert r.verdict == "ok" and r.trace_len == o["trace_len"]
        return
    got = (r.verdict, prog.invariant(r.violated_invariant) if r.verdict == "invariant" else None, r.trace_len)
    assert got in [tuple(e) for e in o["errors"]], (got, o["errors"])


def check_compiled_program_on_gpu(amd, path, invs, consts):
    prog = amd.Program(path.read_text(), cfg_text(invs, consts))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    r = eng.run()
    o = Checker(prog.translated(), constants=consts).run_levels(invariants=invs)
    for k in ("distinct", "generated", "queue_left", "depth", "levels"):
        assert getattr(r, k) == o[k], (k, getattr(r, k), o[k])
    same_outcome(r, o, prog)
    first = 0
    for lvl, n in enumerate(r.levels):
        got = sorted(t.replace("\n", " ") for t in eng.state_texts(first, n))
        assert got == o["states"][lvl], f"level {lvl + 1}"
        first += n
    if r.verdict != "ok":
        tr = eng.trace()
        assert len(tr) == r.trace_len and tr[0][0] == "Initial predicate"
    eng.close()
    prog.close()


@pytest.mark.parametrize("bound,invs,verdict", [(6, ["NeverAhead"], "ok"), (40, [], "ok"), (4, ["NeverAhead", "Small"], "invariant")])
def test_constraint_bounds_an_infinite_algorithm_on_gpu(amd, bound, invs, verdict):
    """cfg CONSTRAINT for compiled programs: growing_counters has an infinite state space; states outside the constraint are
    generated and checked, not stored (FIFO/MCInnerFIFO.cfg:23-26).  Same graph as the TLA+ evaluator, level by level."""
    text = (ROOT / "specs" / "pluscal" / "growing_counters.tla").read_text()
    prog = amd.Program(text, cfg_text(invs, {"Bound": bound}) + "CONSTRAINT Small\n")
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    r = eng.run()
    o = Checker(prog.translated(), constants={"Bound": bound}).run_levels(invariants=invs, constraints=["Small"])
    for k in ("distinct", "generated", "queue_left", "depth", "verdict", "trace_len", "levels"):
        assert getattr(r, k) == o[k], (k, getattr(r, k), o[k])
    assert r.verdict == verdict
    first = 0
    for lvl, n in enumerate(r.levels):
        assert sorted(t.replace("\n", " ") for t in eng.state_texts(first, n)) == o["states"][lvl], f"level {lvl + 1}"
        first += n
    if verdict == "invariant":
        tr = eng.trace()
        assert prog.invariant(r.violated_invariant) == "Small" and len(tr) == bound + 2 and f"produced = {bound + 1}" in tr[-1][1]
    eng.close()
    prog.close()


def test_mc_reads_constraint_from_the_cfg_beside_the_module():
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    p = subprocess.run([str(mc), str(ROOT / "specs" / "pluscal" / "growing_counters.tla")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    text = (ROOT / "specs" / "pluscal" / "growing_counters.tla").read_text()
    o = Checker(text, constants={"Bound": 6}).run(invariants=["NeverAhead"], constraints=["Small"])
    assert "Model checking completed. No error has been found." in p.stdout
    assert f"{o['generated']} states generated, {o['distinct']} distinct states found, 0 states left on queue." in p.stdout


def test_chunking_and_small_tables_do_not_change_the_graph(amd):
    path, invs, consts = CASES[-1]
    prog = amd.Program(path.read_text(), cfg_text(invs, consts))
    ref = None
    for chunk in (256, 1 << 10, 1 << 14):
        eng = amd.Engine("pcal", prog.params, table_capacity=1 << 14, arena_capacity=1 << 12, chunk_states=chunk)
        r = eng.run()
        ref = ref or r
        assert (r.distinct, r.generated, r.levels) == (ref.distinct, ref.generated, ref.levels)
        eng.close()
    prog.close()


def run_mc(*args):
    p = subprocess.run([str(MC), *map(str, args)], capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


def test_mc_on_untranslated_reference_shaped_module(tmp_path):
    """`mc pcal_intro.tla` on the module AS THE REFERENCE COMMITS IT (no translation in the file)"""
    src = strip_translation((ROOT / "specs" / "pcal_intro.tla").read_text())
    assert "BEGIN TRANSLATION" not in src
    (tmp_path / "pcal_intro.tla").write_text(src)
    shutil.copy(ROOT / "specs" / "pcal_intro.cfg", tmp_path / "pcal_intro.cfg")
    rc, out, err = run_mc(tmp_path / "pcal_intro.tla", "-generic")
    assert rc == 0, err
    assert "5850 states generated, 3800 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 5." in out


def test_mc_generic_report_equals_hand_lowering_report():
    f = ROOT / "specs" / "readme_variant" / "pcal_intro.tla"
    rc1, out1, _ = run_mc(f)
    rc2, out2, err = run_mc(f, "-generic")
    assert rc1 == rc2 == 12, err
    # same verdict, same counters, same action positions, same nested-expression positions (README.md:267-321);
    # the counterexample itself may be another shortest one (and differ from run to run): compare everything but
    # the states and which action led to each
    keep = lambda s: [l for l in s.splitlines() if not l.startswith("/\\") and not l.startswith("State ")]  # noqa: E731
    assert keep(out1) == keep(out2)
    acts = lambda s: {l.split(": ", 1)[1] for l in s.splitlines() if l.startswith("State ")}  # noqa: E731
    assert acts(out2) <= {"<Initial predicate>", "<Action line 35, col 19 to line 40, col 42 of module pcal_intro>",
                          "<Action line 42, col 12 to line 45, col 63 of module pcal_intro>",
                          "<Action line 47, col 12 to line 50, col 65 of module pcal_intro>"}             # README.md:278-306
    assert out1.count("\nState ") == out2.count("\nState ") == 6
    assert '"Failure of assertion at line 16, column 4."' in out2
    assert "0. Line 52, column 15 to line 52, column 28 in pcal_intro" in out2     # README.md:315
    assert "1. Line 53, column 15 to line 54, column 66 in pcal_intro" in out2     # README.md:316


def test_mc_new_pluscal_specs():
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "peterson.tla")
    assert rc == 0, err
    assert "105 states generated, 58 distinct states found, 0 states left on queue." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "cas_counter.tla")
    assert rc == 0 and "273 states generated, 159 distinct states found" in out, err
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "lost_update.tla")
    assert rc == 12, err
    lines = out.splitlines()
    assert lines[1] == "The first argument of Assert evaluated to FALSE; the second argument was:"
    assert lines[2] == '"Failure of assertion at line 40, column 5."'
    assert out.count("\nState ") == 7 and "<Action line" in out and "of module lost_update>" in out
    assert '/\\ pc = <<"Done", "Done", "Final">>' in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "euclid.tla")
    assert rc == 0 and "1768 states generated, 1624 distinct states found" in out, err


def test_mc_on_a_module_with_procedures():
    """`mc treiber_procs.tla` = tlc on a PlusCal module that uses PROCEDURES (round 4): the counts of pcal2tla's stack translation
    (tests/golden/pcal_procedures/TreiberStack.tla, evaluated in tests/test_pcal.py) on the GPU"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "treiber_procs.tla")
    assert rc == 0, err
    assert "574 states generated, 330 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 22." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "proc_nested.tla")
    assert rc == 0 and "12735 distinct states found" in out, err


def test_mc_on_a_module_with_records():
    """`mc treiber_records.tla` / `mc ring_buffer.tla` = tlc on PlusCal modules whose variables are RECORDS (round 4; kept field by
    field, tla_rust_amd/csrc/pcal.h): the counts of the record-valued translations pcal2tla would write
    (tests/golden/pcal_records/*.tla, evaluated in tests/test_pcal.py) on the GPU; the torn ring buffer's assertion is found"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "treiber_records.tla")
    assert rc == 0, err
    assert "19363 states generated, 9052 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 24." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ring_buffer.tla")
    assert rc == 0 and "88 states generated, 55 distinct states found" in out, err
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ring_buffer.tla", "-config", ROOT / "specs" / "pluscal" / "ring_buffer_torn.cfg")
    assert rc == 12, err
    assert '"Failure of assertion at line 36, column 9."' in out and "/\\ buf_full = " in out


def test_bigger_program_throughput_smoke(amd):
    """cas_counter with 3 workers x 3 increments: a graph large enough to run many chunks"""
    path = ROOT / "specs" / "pluscal" / "cas_counter.tla"
    prog = amd.Program(path.read_text(), cfg_text(["NeverTooMany", "SeenIsOld"], {"Workers": 3, "N": 3}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 22, arena_capacity=1 << 20, chunk_states=1 << 14)
    r = eng.run()
    assert r.verdict == "ok" and r.distinct > 10000
    eng2 = amd.Engine("pcal", prog.params, table_capacity=1 << 22, arena_capacity=1 << 20, chunk_states=1 << 10)
    r2 = eng2.run()
    assert (r.distinct, r.generated, r.depth) == (r2.distinct, r2.generated, r2.depth)
    eng.close()
    eng2.close()
    prog.close()


def test_reference_makefile_flow(tmp_path):
    """the reference's Makefile:3-7 on its root directory: `pcal2tla *tla` then `tlc *tla`, with mc in both roles.
    The root holds pcal_intro.tla + pcal_intro.cfg and atomic_add.tla (no cfg), both untranslated."""
    for name in ("pcal_intro.tla", "atomic_add.tla"):
        (tmp_path / name).write_text(strip_translation((ROOT / "specs" / name).read_text()))
    shutil.copy(ROOT / "specs" / "pcal_intro.cfg", tmp_path / "pcal_intro.cfg")
    # test: works on the untranslated files ...
    for name, want in (("pcal_intro.tla", "5850 states generated, 3800 distinct states found"), ("atomic_add.tla", "7 states generated, 5 distinct states found")):
        rc, out, err = run_mc(tmp_path / name)
        assert rc == 0 and want in out, (name, out, err)
    # ... transpile: inserts the translation in place, keeps X.old ...
    p = subprocess.run([str(MC), "--transpile", str(tmp_path / "pcal_intro.tla"), str(tmp_path / "atomic_add.tla")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert (tmp_path / "pcal_intro.old").exists() and "\\* BEGIN TRANSLATION" in (tmp_path / "pcal_intro.tla").read_text()
    block = lambda t: t[t.index("\\* BEGIN TRANSLATION"):t.index("\\* END TRANSLATION")]  # noqa: E731
    assert block((tmp_path / "pcal_intro.tla").read_text()) == block((ROOT / "specs" / "pcal_intro.tla").read_text())
    # ... and test again on the translated files: same answers
    for name, want in (("pcal_intro.tla", "5850 states generated, 3800 distinct states found"), ("atomic_add.tla", "7 states generated, 5 distinct states found")):
        rc, out, err = run_mc(tmp_path / name)
        assert rc == 0 and want in out, (name, out, err)


def test_sequence_overflow_is_reported_on_gpu(amd):
    text = (ROOT / "specs" / "pluscal" / "bounded_queue.tla").read_text()
    prog = amd.Program(text, "CONSTANTS Items = 9 MaxQ = 9 Consumers = 1\n")
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 16, arena_capacity=1 << 14)
    with pytest.raises(amd.McError) as e:
        eng.run()
    assert e.value.code == -3 and "sequence" in str(e.value)      # MC_EOVERFLOW
    eng.close()
    prog.close()


def test_mc_bounded_queue_race():
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "bounded_queue.tla", "-config", ROOT / "specs" / "pluscal" / "bounded_queue_race.cfg")
    assert rc == 12, err
    assert '"Failure of assertion at line 37, column 9."' in out and out.count("\nState ") == 10
    assert "/\\ queue = <<" in out


def test_mc_dump_writes_every_state(tmp_path):
    """`mc X.tla -dump FILE` = TLC's -dump: every distinct state, the same SET the TLA+ evaluator reaches"""
    f = ROOT / "specs" / "pluscal" / "peterson.tla"
    out_file = tmp_path / "states.dump"
    rc, out, err = run_mc(f, "-dump", out_file)
    assert rc == 0, err
    blocks = [b for b in out_file.read_text().split("\n\n") if b.strip()]
    assert len(blocks) == 58 and blocks[0].startswith("State 1:\n/\\ flag = ")
    got = sorted(" ".join(b.splitlines()[1:]) for b in blocks)
    o = Checker(f.read_text()).run_levels(invariants=["MutualExclusion", "TurnInRange"])
    assert got == sorted(s for lvl in o["states"] for s in lvl)


def test_random_algorithms_on_gpu(amd):
    """the seeded random algorithms of tests/test_pcal_random.py through the HIP engine"""
    from test_pcal_random import Gen
    checked = 0
    for seed in range(1000, 1040):
        text = Gen(seed).module(f"rnd{seed}")
        try:
            prog = amd.Program(text, "INVARIANT Small\n")
        except amd.McError:
            continue   # the generator broke a PlusCal rule; refusals are covered on the CPU
        eng = amd.Engine("pcal", prog.params, table_capacity=1 << 18, arena_capacity=1 << 16, chunk_states=1 << 10, deadlock=False)
        r = eng.run()
        o = Checker(prog.translated()).run_levels(invariants=["Small"], check_deadlock=False)
        for k in ("distinct", "generated", "queue_left", "depth", "levels"):
            assert getattr(r, k) == o[k], (seed, k, getattr(r, k), o[k])
        same_outcome(r, o, prog)
        first = 0
        for lvl, n in enumerate(r.levels):
            assert sorted(t.replace("\n", " ") for t in eng.state_texts(first, n)) == o["states"][lvl], (seed, lvl)
            first += n
        eng.close()
        prog.close()
        checked += 1
    assert checked >= 25


def test_wide_state_on_gpu(amd):
    """72 scalar cells (the 128-cell interpreter instantiation): 70 adders, level-budgeted, against the closed form"""
    from math import comb
    prog = amd.Program((ROOT / "specs" / "atomic_add_n.tla").read_text(), "CONSTANT N = 70\n")
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 22, arena_capacity=1_100_000, chunk_states=1 << 16, max_distinct=100_000, trace=False)
    r = eng.run()
    assert r.levels == [comb(70, k) for k in range(5)] and r.verdict == "budget"
    assert r.generated == 1 + sum(comb(70, k) * (70 - k) for k in range(4))
    eng.close()
    prog.close()


@pytest.mark.parametrize("seed", range(60))
def test_random_algorithm_on_gpu_vs_tla_evaluator(amd, seed, tmp_path):
    """the seeded random algorithms of tests/test_pcal_fuzz.py (records, set / sequence variables, macros, procedures, every kind of
    statement) compiled and run by the HIP engine, against the oracle's evaluator on the translation: counters, verdict, trace
    length, per-level counts and the set of states of every level"""
    import helpers
    from test_pcal_fuzz import Gen, MAX_STATES
    text, invs = Gen(seed).program()
    try:
        helpers.pcal_translate(text)
    except RuntimeError as e:
        pytest.skip(f"refused: {e}")
    prog = helpers.ShimProgram(text, invs, {})
    try:
        n = helpers.shim_run("pcal", prog.params)["distinct"]
    finally:
        prog.close()
    if n > MAX_STATES:
        pytest.skip(f"{n} states: too many for the Python evaluator in a unit test")
    path = tmp_path / "Fz.tla"
    path.write_text(text)
    test_compiled_program_on_gpu_vs_tla_evaluator(amd, path, invs, {})
