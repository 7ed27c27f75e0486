"""The TLA+ text evaluator (oracle/tlaplus.py — the thing that pins the C oracle to the reference's raft / SI texts) against the
only real TLC logs in the reference tree: examples/SpecifyingSystems/AdvancedExamples/testout1 and testout2, TLC 1.57 on
MCInnerSerial.tla + MCInnerSerial.cfg (records, SUBSET of relation sets, CHOOSE, `<-` overrides, a CONSTRAINT, a tuple-subscripted
fairness formula).

  testout2:3 / testout1:3   "Finished computing initial states: 4 distinct states generated."
  testout1:4                "Progress(4): 772 states generated, 160 distinct states found, 79 states left on queue."
                            — TLC's first progress report: level 4 has just been completed (80 states on the queue, one of them
                            dequeued), 772 successors generated so far.
  testout2:265-266          6181 states generated, 195 distinct states found, diameter 5 — TLC needed 22 hours for the whole run
                            (testout2:267); the fifth level alone is hours of evaluation and is not run here.

Runs only where /root/reference exists (the build container)."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
D = Path("/root/reference/examples/SpecifyingSystems/AdvancedExamples")

pytestmark = pytest.mark.skipif(not D.exists(), reason="/root/reference is only present in the build container")


def _checker():
    import tlaplus as T
    return T.Checker(D / "MCInnerSerial.tla", cfg_text=(D / "MCInnerSerial.cfg").read_text(),
                     search=[D, D.parent / "Standard", D.parent / "CachingMemory", D.parent / "TLC"])


def test_initial_states_of_the_reference_tlc_log():
    r = _checker().run_levels(max_levels=1)
    assert (r["distinct"], r["generated"], r["levels"]) == (4, 4, [4])                   # testout2:3


def test_first_progress_line_of_the_reference_tlc_log():
    r = _checker().run_levels(max_levels=4)
    assert r["levels"] == [4, 16, 60, 80]
    assert (r["generated"], r["distinct"]) == (772, 160)                                 # testout1:4
    assert r["queue_left"] == 80                                                         # TLC had dequeued one of them: 79


def test_complete_run_of_the_reference_tlc_log():
    """testout2:260-266, the end of the one complete TLC run the reference holds a log of: "Model checking completed. No error
    has been found." / "6181 states generated, 195 distinct states found, 0 states left on queue." / "The state graph has
    diameter 5."  The evaluator's complete run (about an hour on 8 cores: tests/golden/make_tlc_log_golden.py expands every
    level's frontier with forked workers) is committed as tests/golden/tlc_log_mcinnerserial.json; the fixture must say what
    TLC's log says, and the four levels this suite can afford to re-evaluate must be its first four."""
    import json
    g = json.loads((ROOT / "tests" / "golden" / "tlc_log_mcinnerserial.json").read_text())
    assert (g["generated"], g["distinct"], g["depth"], g["verdict"]) == (6181, 195, 5, "ok")       # testout2:265-266, :260
    assert g["levels"][:4] == [4, 16, 60, 80] and sum(g["levels"]) == 195
    assert any(p["levels"] == 4 and (p["generated"], p["distinct"]) == (772, 160) for p in g["progress"])   # testout1:4 on the way
    r = _checker().run_levels(max_levels=4)
    assert r["levels"] == g["levels"][:4]
