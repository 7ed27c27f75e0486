"""GPU tests of specs/pluscal/ms_queue.tla (the Michael-Scott lock-free queue).  The module was added after the round's last GPU
minute; everything it needs was checked on the host (the product's own compilation of it runs on the host build of the interpreter with
the evaluator's counts, tests/test_pcal.py).  The file name sorts behind every other GPU file on purpose: under `pytest -x` a surprise
here cannot keep the rest of the suite from running."""
from pathlib import Path

import pytest

from test_gpu_pcal import ROOT, amd, check_compiled_program_on_gpu, run_mc  # noqa: F401  (amd: the fixture)
from test_pcal import CASES

pytestmark = pytest.mark.gpu
MSQ = [c for c in CASES if c[0].stem == "ms_queue"]


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
