"""GPU tests of specs/pluscal/ms_queue.tla (the Michael-Scott lock-free queue).  The module was added after the round's last GPU
minute; everything it needs was checked on the host (the product's own compilation of it runs on the host build of the interpreter with
the evaluator's counts, tests/test_pcal.py).  The file name sorts behind every other GPU file on purpose: under `pytest -x` a surprise
here cannot keep the rest of the suite from running."""
from pathlib import Path

import pytest

from test_gpu_pcal import ROOT, amd, cfg_text, check_compiled_program_on_gpu, run_mc  # noqa: F401  (amd: the fixture)
from test_pcal import CASES

pytestmark = pytest.mark.gpu
MSQ = [c for c in CASES if c[0].stem.startswith("ms_queue")]   # ms_queue.tla and (round 5, nested records) ms_queue_counted.tla


@pytest.mark.parametrize("path,invs,consts", MSQ, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_ms_queue_compiled_program_on_gpu_vs_tla_evaluator(amd, path, invs, consts):  # noqa: F811
    check_compiled_program_on_gpu(amd, path, invs, consts)


def test_mc_on_the_michael_scott_queue():
    """`mc ms_queue.tla` = tlc on the lock-free linked-list queue (the "lists" of the reference's roadmap, README.md:26-42): three threads,
    the counts the TLA+ evaluator gives for the translation (tests/test_pcal.py); with the linking CAS replaced by a plain store
    (ms_queue_racy.cfg) the Fifo invariant breaks (an 18-state behaviour: tests/test_pcal.py compares its length on the host)"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ms_queue.tla")
    assert rc == 0, err
    assert "228229 states generated, 91727 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 40." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ms_queue.tla", "-config", ROOT / "specs" / "pluscal" / "ms_queue_racy.cfg")
    assert rc == 12, err
    assert "Error: Invariant Fifo is violated." in out and "State 1: <Initial predicate>" in out


def test_mc_on_the_counted_pointer_queue():
    """`mc ms_queue_counted.tla` = tlc on the Michael-Scott queue as published: every pointer a (ptr, count) record nested in the queue /
    node records (kept leaf by leaf: tla_rust_amd/csrc/pcal.cpp RecordFlattener), nodes freed and reused.  The counts are those of the
    hand-written record-valued translation (tests/golden/pcal_records/MsQueueCounted.tla, evaluated in tests/test_pcal.py); comparing
    the ptr halves only (ms_queue_uncounted.cfg) lets a delayed compare-and-swap put Head on a freed node"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ms_queue_counted.tla")
    assert rc == 0, err
    assert "43013 states generated, 22670 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 69." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "ms_queue_counted.tla", "-config", ROOT / "specs" / "pluscal" / "ms_queue_uncounted.cfg")
    assert rc == 12, err
    # (the failing pass holds two errors — HeadLive broken by a successor of one state, an assert failing in another: a parallel search may
    #  report either, test_gpu_pcal.same_outcome)
    assert ("Error: Invariant HeadLive is violated." in out or "The first argument of Assert evaluated to FALSE" in out) and "/\\ Q_Head_ptr = " in out


def test_counted_pointer_queue_three_threads_on_gpu(amd):  # noqa: F811
    """three threads, three nodes: 35 263 910 states / 99 861 367 generated / depth 105, every invariant holds.  The expected numbers are
    those of the SAME compiled program on the host build of the interpreter (tests/_shim, 240 s; the evaluators cannot walk 35 M
    states): a device-against-host check of the engine on a compiled program of 45 cells, not an independent pin — the independent
    pins are the two-thread cases above"""
    path = ROOT / "specs" / "pluscal" / "ms_queue_counted.tla"
    invs = ["HeadLive", "TailLive", "PointersAreNodes", "TailAtMostOneBehind", "CountsGrow"]
    prog = amd.Program(path.read_text(), cfg_text(invs, {"N": 3, "K": 3, "Counted": True}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 28, arena_capacity=40 << 20, chunk_states=1 << 20, deadlock=True)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (35263910, 99861367, 105, "ok", 0)
    eng.close()
    prog.close()
