"""The C oracle (oracle/spec_ssi.c) pinned to the REFERENCE'S OWN TEXT: oracle/tlaplus.py evaluates
/root/reference/examples/serializableSnapshotIsolation.tla:219-996 (and textbookSnapshotIsolation.tla) under specs/MCssi.tla /
MCtextbookSI.tla the way TLC does — recursive operators, CHOOSE, SelectSeq with LAMBDA, records, sets of records, the wait-for
graph walk, all eight "should never be violated" invariants of :59-79 evaluated on every state — and the hand restatement must
give the same state graph: per-level SETS of states as canonical TLA+ text, counters, depth, verdict.

/root/reference exists only in the build container: there the evaluator runs on the reference file itself and must reproduce
the committed fixture (tests/golden/ssi_reference_text.json, made by tests/golden/make_reference_text_golden.py); everywhere
the fixture is compared with the C oracle.
"""
import hashlib
import json
import sys
from pathlib import Path

import pytest

import helpers

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests" / "golden"))
REF = Path("/root/reference/examples")
GOLD = json.loads((ROOT / "tests" / "golden" / "ssi_reference_text.json").read_text())

from make_reference_text_golden import SSI_MODELS, run_ssi_text  # noqa: E402


def level_digests(by_level):
    return [hashlib.sha256("\n".join(sorted(by_level[k])).encode()).hexdigest()[:16] for k in sorted(by_level)]


DEEP = {"ssi_2x3": [2, 3, 127, 0]}   # tests/golden/make_deep_text_pin.py: the product's C++ evaluator on the text (1 h 51 min) == the C oracle


@pytest.mark.parametrize("name", sorted(DEEP))
def test_c_oracle_equals_the_deep_reference_text_fixture(name, tmp_path):
    """serializableSnapshotIsolation.tla at 2 txns x 3 keys — 7 910 565 states, 17 levels — evaluated from the
    reference's TEXT by tlaeval.cpp and equal, level by level as state SETS, to the C oracle (an independent pair: the evaluator is no
    port of the oracle).  The suite re-runs the oracle against the fixture's counters (20 s); with TLAMC_SLOW=1 it also dumps the 7.9 M
    states (5 GB of text) and compares the per-level digests; re-evaluating the text is `python tests/golden/make_deep_text_pin.py ssi_2x3`."""
    import os
    g = GOLD[name]
    assert "make_deep_text_pin" in g["source"] and g["distinct"] == 7910565
    slow = os.environ.get("TLAMC_SLOW") == "1"
    dump = tmp_path / "dump.txt"
    o = helpers.oracle_run("ssi", DEEP[name], dump=str(dump) if slow else None)
    assert (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]) == (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    if slow:
        sys.path.insert(0, str(ROOT / "tests" / "golden"))
        from make_deep_text_pin import digests
        assert digests(dump) == g["level_digests"]


@pytest.mark.parametrize("name", sorted(n for n in GOLD if n not in DEEP))
def test_c_oracle_equals_reference_text_fixture(name, tmp_path):
    g = GOLD[name]
    dump = tmp_path / "dump.txt"
    o = helpers.oracle_run("ssi", SSI_MODELS[name]["params"], dump=str(dump))
    assert (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]) == \
           (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    by_level = helpers.read_dump(str(dump))
    if len(SSI_MODELS[name]["params"]) > 4 and SSI_MODELS[name]["params"][4]:
        # textbookSnapshotIsolation.tla has three variables; the oracle prints Cahill's three (never touched there) as well
        by_level = {k: [t.split(" /\\ inConflict")[0] for t in v] for k, v in by_level.items()}
    assert level_digests(by_level) == g["level_digests"]


def test_known_counts_of_the_survey():
    """SURVEY.md section 6 (an independent Python BFS of the surveyor): 569 and 29 629 distinct states"""
    assert GOLD["ssi_2x1"]["distinct"] == 569
    if "ssi_2x2" in GOLD:
        assert (GOLD["ssi_2x2"]["distinct"], GOLD["ssi_2x2"]["generated"], GOLD["ssi_2x2"]["depth"]) == (29629, 50121, 13)


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
def test_fixture_is_what_the_reference_text_gives():
    r = run_ssi_text("ssi_2x1")
    assert {k: r[k] for k in GOLD["ssi_2x1"]} == GOLD["ssi_2x1"]


SWEEP = [(2, 3, 0, 7), (3, 2, 0, 7), (4, 1, 0, 7), (3, 1, 1, 9), (2, 2, 1, 11), (3, 2, 1, 7), (4, 2, 0, 6), (3, 3, 0, 6)]   # (each under ~10 s)


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
@pytest.mark.parametrize("nt,nk,textbook,depth", SWEEP)
def test_c_oracle_equals_the_reference_text_on_other_sizes(nt, nk, textbook, depth, tmp_path):
    """serializableSnapshotIsolation.tla / textbookSnapshotIsolation.tla, evaluated from the reference's TEXT by the product's C++
    evaluator (tlaeval.cpp) at sizes between and beyond the fixtures' (up to 4 transactions x 2 keys and 3 x 3, both variants),
    against the C oracle's hand restatement: counters, per-level counts and the per-level SETS of states over the first `depth`
    levels, every invariant checked on every state"""
    from make_reference_text_golden import SSI_INVARIANTS, SSI_ORDER, TEXTBOOK_ORDER, ssi_cfg
    cfg = tmp_path / "m.cfg"
    cfg.write_text(ssi_cfg(nt, nk, [i for i in SSI_INVARIANTS if not (textbook and i in ("CahillOK", "BernsteinOK"))]))
    ed, od = tmp_path / "e.txt", tmp_path / "o.txt"
    e = helpers.tlaeval_run(ROOT / "specs" / ("MCtextbookSI.tla" if textbook else "MCssi.tla"), cfg, search=[str(REF)], dump=ed,
                            order=TEXTBOOK_ORDER if textbook else SSI_ORDER, max_levels=depth)
    assert e["rc"] == 0, e
    o = helpers.oracle_run("ssi", [nt, nk, 127, 0, textbook], dump=str(od), max_levels=depth)
    assert (o["distinct"], o["generated"], o["depth"], o["levels"]) == (e["distinct"], e["generated"], e["depth"], e["levels"])
    mine = helpers.read_dump(str(od))
    if textbook:   # three variables in the text; the oracle prints Cahill's three (never touched there) as well
        mine = {k: [t.split(" /\\ inConflict")[0] for t in v] for k, v in mine.items()}
    assert level_digests(mine) == level_digests(helpers.read_dump(str(ed)))
    assert o["distinct"] > 1000
