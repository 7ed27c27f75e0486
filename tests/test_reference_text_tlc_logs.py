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
