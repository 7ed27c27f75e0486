"""The C oracle (oracle/spec_raft.c) pinned to the REFERENCE'S OWN TEXT: oracle/tlaplus.py evaluates
/root/reference/examples/raft.tla:110-507 under specs/MCraft.tla the way TLC does, and the hand restatement must give the
same state graph — per-level SETS of states as canonical TLA+ text, counters, depth — on the 2-server anchors of
BASELINE.md section 2 (6 128 and 13 634 distinct) and on THREE-server models (the bench model's cfg with 4 / 5 / 6 message keys:
48 274 / 178 654 / 641 869 states — one election, then the first AppendEntries round trip), including the negative control of SURVEY.md App. B item 0: evaluating
raft.tla:392-393 as an unconditional assignment ("naive") yields 15 794.

/root/reference exists only in the build container: there the test runs the evaluator on the reference file itself and
checks the committed fixture (tests/golden/raft_reference_text.json, made by tests/golden/make_reference_text_golden.py) is
what it produces; on the GPU box (no /root/reference) the fixture alone is compared with the oracle.
"""
import hashlib
import json
import sys
from pathlib import Path

import pytest

import helpers

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
REF = Path("/root/reference/examples")
GOLD = json.loads((ROOT / "tests" / "golden" / "raft_reference_text.json").read_text())

sys.path.insert(0, str(ROOT / "tests" / "golden"))
from make_reference_text_golden import RAFT_MODELS, oracle_params, raft_cfg, run_raft_text  # noqa: E402


def level_digests(by_level):
    return [hashlib.sha256("\n".join(sorted(by_level[k])).encode()).hexdigest()[:16] for k in sorted(by_level)]


@pytest.mark.parametrize("name", sorted(RAFT_MODELS))
def test_c_oracle_equals_reference_text_fixture(name, tmp_path):
    """C oracle vs the fixture produced from the reference's text: counters, per-level counts, per-level state-set digests"""
    g = GOLD[name]
    dump = tmp_path / "dump.txt"
    o = helpers.oracle_run("raft", oracle_params(name), dump=str(dump))
    if RAFT_MODELS[name]["clash"] == "ignore":
        # the negative control: the oracle (TLC semantics) must NOT reproduce the naive count
        assert o["distinct"] != g["distinct"] and g["distinct"] == 15794
        return
    assert (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]) == \
           (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    assert level_digests(helpers.read_dump(str(dump))) == g["level_digests"]


def test_c_oracle_and_lowering_equal_the_deep_reference_text_fixture(tmp_path):
    """raft.tla at 3 servers / 7 message keys — 2 303 950 states, 27 levels — evaluated from the reference's TEXT by
    the product's C++ evaluator (tlaeval.cpp, 9.5 min) and equal, level by level as state SETS, to the C oracle: an independent pair (the
    C++ evaluator is a port of oracle/tlaplus.py, not of the oracle).  The suite re-runs the oracle AND the host build of the device
    lowering against the fixture's counters; with TLAMC_SLOW=1 also against the per-level digests (a 4 GB dump each); re-evaluating the
    text is `python tests/golden/make_deep_text_pin.py raft_3s_keys7`."""
    import os
    g = GOLD["raft_3s_keys7"]
    assert "make_deep_text_pin" in g["source"] and g["distinct"] == 2303950
    slow = os.environ.get("TLAMC_SLOW") == "1"
    dump = tmp_path / "dump.txt"
    o = helpers.oracle_run("raft", [3, 4, 2, 3, 1, 3, 0, 7], dump=str(dump) if slow else None)
    assert (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]) == (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    if slow:
        from make_deep_text_pin import digests
        assert digests(dump) == g["level_digests"]
    s = helpers.shim_run("raft", [3, 4, 2, 3, 1, 3, 0, 0, 0, 7], dump=str(dump) if slow else None)
    assert (s["distinct"], s["generated"], s["depth"], s["levels"], s["verdict"], s["fp_mismatch"]) == (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"], 0)
    if slow:
        assert digests(dump) == g["level_digests"]


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
@pytest.mark.parametrize("name", ["raft_2s_mcr1"])   # the others: python tests/golden/make_reference_text_golden.py raft <name> (1 - 15 min each)
def test_fixture_is_what_the_reference_text_gives(name):
    r = run_raft_text(name)
    g = GOLD[name]
    assert {k: r[k] for k in g} == g


@pytest.mark.parametrize("name", ["raft_2s_mcr2_keys8", "raft_3s_keys4", "raft_3s_keys5", "raft_3s_keys6"])
def test_lowering_equals_reference_text_fixture(name, tmp_path):
    """the DEVICE lowering (tla_rust_amd/csrc/spec_raft.h, host build) against the fixture made from the reference's text — three
    servers included, where a quorum is a real majority: counters, per-level counts, per-level state-set digests"""
    from make_reference_text_golden import device_params
    g = GOLD[name]
    dump = tmp_path / "dump.txt"
    s = helpers.shim_run("raft", device_params(name), dump=str(dump))
    assert (s["distinct"], s["generated"], s["depth"], s["levels"], s["verdict"]) == \
           (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    assert s["fp_mismatch"] == 0
    assert level_digests(helpers.read_dump(str(dump))) == g["level_digests"]


def test_models_use_the_committed_wrapper():
    assert "EXTENDS raft" in (ROOT / "specs" / "MCraft.tla").read_text()
    assert "MaxTerm = 2" in raft_cfg(2, 1, 2, 9, 1)


def _sweep_config(seed):
    import random
    r = random.Random(500 + seed)
    n = r.choice([2, 2, 2, 3])
    mcr, mt, mll, mm = r.randrange(1, 4), r.choice([2, 3]), r.choice([2, 3, 9]), r.choice([1, 1, 2])
    mk = r.choice([4, 5, 6, 8]) if n == 2 else r.choice([3, 4])
    inv = r.choice([1, 3])
    return [n, mcr, mt, mll, mm, inv], mk


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
@pytest.mark.parametrize("seed", range(16))
def test_c_oracle_equals_the_reference_text_on_random_configurations(seed, tmp_path):
    """the reference's raft.tla, evaluated from its TEXT by the product's C++ evaluator (tlaeval.cpp) under seeded random constants
    (server count, MaxClientRequests, MaxTerm, MaxLogLen, MaxMsgs, MaxMsgKeys, invariants), against the C oracle's hand restatement:
    counters, per-level counts and the per-level SETS of states (canonical TLA+ text) over the first 11 levels (9 with three servers) — the fixed models
    above pin the configurations the goldens use, this walks the parameter space between them.  (To 13 levels — up to 330 s a case —
    the first ten seeds were equal too when the test was written.)"""
    params, mk = _sweep_config(seed)
    cfg = tmp_path / "m.cfg"
    cfg.write_text(raft_cfg(*params[:5], params[5], mk))
    from make_reference_text_golden import RAFT_ORDER
    ed, od = tmp_path / "e.txt", tmp_path / "o.txt"
    depth = 9 if params[0] == 3 else 11          # (what keeps a case under ~20 s of evaluation)
    e = helpers.tlaeval_run(ROOT / "specs" / "MCraft.tla", cfg, search=[str(REF)], dump=ed, order=RAFT_ORDER, max_levels=depth)
    assert e["rc"] == 0, e
    o = helpers.oracle_run("raft", params + [0, mk], dump=str(od), max_levels=depth)
    assert (o["distinct"], o["generated"], o["depth"], o["levels"]) == (e["distinct"], e["generated"], e["depth"], e["levels"]), (params, mk)
    assert level_digests(helpers.read_dump(str(od))) == level_digests(helpers.read_dump(str(ed))), (params, mk)
    assert o["distinct"] > 300
