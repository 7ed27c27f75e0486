"""The C oracle (oracle/spec_paxos.c) and the device lowering (tla_rust_amd/csrc/spec_paxos.h, host build) pinned to the
REFERENCE'S OWN TEXT: oracle/tlaplus.py evaluates /root/reference/examples/Paxos/{Voting,Paxos,Consensus}.tla under the
reference's model modules MCVoting.tla + .cfg and MCPaxos.tla + .cfg exactly as committed (INSTANCE with implicit
substitution, Thm!: and Def!k selectors, `Ballot <-[Voting] MCBallot`, PROPERTY C!Spec / V!Spec as a per-transition
refinement check, SYMMETRY) and under specs/paxos/MCPaxos3.tla, which carries the three-acceptor sizes MCPaxos.tla:7-9 names
in its comments.  Counters, depth, per-level counts; per-level state SETS as canonical TLA+ text where no SYMMETRY is
involved (under SYMMETRY the stored representative is whichever state of an orbit is met first: counts are compared).

/root/reference exists only in the build container: there the evaluator is run on the reference files themselves and must
reproduce the committed fixture (tests/golden/paxos_reference_text.json, made by tests/golden/make_reference_text_golden.py);
everywhere the fixture is compared with the oracle and with the lowering.  Deadlock checking is off: Voting over a finite
Ballot set ends in states without successors.
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
REF = Path("/root/reference/examples/Paxos")
GOLD = json.loads((ROOT / "tests" / "golden" / "paxos_reference_text.json").read_text())

from make_reference_text_golden import (PAXOS_MODELS, PAXOS_NEGATIVE, VOTING_ALTERNATIVES, run_paxos_negative, run_paxos_text,  # noqa: E402
                                        run_voting_alternative)


def level_digests(by_level):
    return [hashlib.sha256("\n".join(sorted(by_level[k])).encode()).hexdigest()[:16] for k in sorted(by_level)]


def test_reference_numbers():
    """what the evaluator found for the reference's two model files as committed"""
    assert (GOLD["voting_mc"]["distinct"], GOLD["voting_mc"]["generated"], GOLD["voting_mc"]["depth"]) == (77, 406, 11)
    assert (GOLD["paxos_mc"]["distinct"], GOLD["paxos_mc"]["generated"], GOLD["paxos_mc"]["depth"]) == (25, 82, 9)
    assert all(GOLD[m]["verdict"] == "ok" for m in PAXOS_MODELS)   # Inv / Inv1..Inv4 and the PROPERTY hold


@pytest.mark.parametrize("name", sorted(PAXOS_MODELS))
def test_c_oracle_equals_reference_text_fixture(name, tmp_path):
    g = GOLD[name]
    dump = tmp_path / "dump.txt"
    o = helpers.oracle_run("paxos", PAXOS_MODELS[name]["params"], check_deadlock=False, dump=str(dump))
    assert (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]) == \
           (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    if "level_digests" in g:
        assert level_digests(helpers.read_dump(str(dump))) == g["level_digests"]


@pytest.mark.parametrize("name", sorted(PAXOS_MODELS))
def test_lowering_equals_reference_text_fixture(name, tmp_path):
    g = GOLD[name]
    dump = tmp_path / "dump.txt"
    s = helpers.shim_run("paxos", PAXOS_MODELS[name]["params"], check_deadlock=False, dump=str(dump))
    assert (s["distinct"], s["generated"], s["depth"], s["levels"], s["verdict"]) == \
           (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    assert s["fp_mismatch"] == 0
    if "level_digests" in g:
        assert level_digests(helpers.read_dump(str(dump))) == g["level_digests"]


@pytest.mark.parametrize("name", sorted(PAXOS_NEGATIVE))
def test_negative_controls(name):
    """bad quorums break C!Spec (and nothing else); a Phase2a without its quorum conjunct breaks Inv!3 one step after Init:
    evaluator, oracle and lowering name the same violated formula and the same shortest counterexample length"""
    g = GOLD[name]
    params = PAXOS_NEGATIVE[name]["params"]
    o = helpers.oracle_run("paxos", params, check_deadlock=False)
    s = helpers.shim_run("paxos", params, check_deadlock=False)
    assert g["verdict"] in ("invariant", "property")
    assert (o["verdict"], o["violated_invariant"], len(o["trace"])) == ("invariant", g["index"], g["trace_len"])
    assert (s["verdict"], s["violated_invariant"], s["trace_len"]) == ("invariant", g["index"], g["trace_len"])
    assert g["name"] == {"voting_badquorum": "ConsensusSpecBar", "paxos_bad_phase2a": "Inv3"}[name]


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
@pytest.mark.parametrize("name", [n for n in sorted(PAXOS_MODELS) if not PAXOS_MODELS[n].get("slow")])
def test_fixture_is_what_the_reference_text_gives(name):
    r = run_paxos_text(name)
    assert r == GOLD[name]


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
@pytest.mark.parametrize("name", sorted(PAXOS_NEGATIVE))
def test_negative_fixture_is_what_the_text_gives(name):
    assert run_paxos_negative(name) == GOLD[name]


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
def test_theorems_of_mcvoting_hold():
    """MCVoting.tla:32,41-46: `ASSUME QuorumNonEmpty!:` and MCInv (the statements of five THEOREMs of Voting.tla as one
    invariant) — evaluated on every reachable state of the unreduced model"""
    import tlaplus as T
    c = T.Checker(REF / "MCVoting.tla", search=[REF], symmetry=False)
    assert all(c.spec.cv(a, {})({}, None, None) is True for m in c.spec.modules for a in m.assumes)
    mcinv = c.spec.compile_value("MCInv")
    r = c.run_levels(check_deadlock=False)
    assert r["distinct"] == 599
    assert all(mcinv({}, st, None) is True for lvl in r["level_states"] for st in lvl)


@pytest.mark.skipif(not REF.exists(), reason="/root/reference is only present in the build container")
def test_mcconsensus_inductive_invariant_model():
    """examples/Paxos/MCConsensus.tla + .cfg as committed: `SPECIFICATION ISpec` with ISpec == IInv /\\ [][Next]_chosen checks that
    Inv is inductive — the initial predicate is the invariant itself (every subset of Value = {"a", "b", "c"} with at most one
    element: 4 initial states; its second conjunct READS what the first one assigned), from {} each value can be chosen"""
    import tlaplus as T
    c = T.Checker(REF / "MCConsensus.tla", search=[REF])
    r = c.run_levels(check_deadlock=False)
    assert (r["distinct"], r["generated"], r["depth"], r["verdict"]) == (4, 7, 1, "ok")
    assert sorted(c.spec.state_text(s) for s in r["level_states"][0]) == ['/\\ chosen = {"a"}', '/\\ chosen = {"b"}', '/\\ chosen = {"c"}', '/\\ chosen = {}']


def test_alternative_configurations_of_mcvoting():
    """MCVoting.cfg:7-8 names two more configurations in its comments (MCVoting.tla:36-55 explains them): the statements of five
    THEOREMs of Voting.tla hold on EVERY type-correct state (110 592 of them, no transitions: [][FALSE]_vars), and Inv is inductive
    (2 771 type-correct states satisfy it; all 11 745 steps from them end in one of the 2 771)"""
    t, i = GOLD["voting_theorems_on_all_type_correct_states"], GOLD["voting_inv_is_inductive"]
    assert (t["distinct"], t["generated"], t["depth"], t["verdict"]) == (110592, 110592, 1, "ok")      # (2^4)^3 vote sets x 3^3 maxBal
    assert (i["distinct"], i["generated"], i["depth"], i["verdict"]) == (2771, 2771 + 11745, 1, "ok")
    if REF.exists():
        assert run_voting_alternative("voting_inv_is_inductive") == i


def test_census_of_all_type_correct_voting_states():
    """the same two configurations restated: the C oracle (oracle_voting_census) and the device lowering (host build,
    shim_voting_census) enumerate all 110 592 type-correct states of the MCVoting model — 2 771 satisfy Inv, Next generates 11 745
    successors from those (TLC's witness multiplicities), none of them violates Inv: the numbers of the evaluator's MCSpecI run.
    This exercises the invariant code on states no reachable-state test ever sees."""
    import ctypes as C
    g = GOLD["voting_inv_is_inductive"]
    params = [1, 3, 2, 2, 1, 0, 0]
    out = (C.c_uint64 * 4)()
    lib = helpers.oracle_lib()
    lib.oracle_voting_census.argtypes = [C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_uint64)]
    assert lib.oracle_voting_census((C.c_int64 * len(params))(*params), len(params), out) == 0
    assert list(out) == [GOLD["voting_theorems_on_all_type_correct_states"]["distinct"], g["distinct"], g["generated"] - g["distinct"], 0]
    assert list(out) == [110592, 2771, 11745, 0]
    sh = helpers.shim_lib()
    d = helpers.spec_desc("paxos", params)
    out2 = (C.c_uint64 * 4)()
    sh.shim_voting_census.argtypes = [C.POINTER(helpers.McSpecDesc), C.POINTER(C.c_uint64)]
    assert sh.shim_voting_census(C.byref(d), out2) == 0
    assert list(out2) == list(out)
