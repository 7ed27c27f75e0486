"""GPU parity of the Paxos family (examples/Paxos/Voting.tla, Paxos.tla): the HIP engine through
the C ABI against the CPU oracle (oracle/spec_paxos.c, itself pinned to the reference's text by
tests/test_reference_text_paxos.py) and against the fixture made from the reference's text
(tests/golden/paxos_reference_text.json).  Every comparison is exact."""
import json
import os
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests" / "golden" / "paxos_reference_text.json").read_text())

# params: {kind (0 Paxos, 1 Voting), nAcceptor, nValue, nBallot, invariant mask, symmetry, property, [nQuorum, masks...]}
NO_SYMMETRY = [[1, 3, 2, 2, 1, 0, 1], [0, 1, 1, 2, 15, 0, 1], [0, 3, 2, 2, 15, 0, 1], [1, 2, 3, 3, 1, 0, 1], [0, 2, 3, 2, 15, 0, 1]]
# (the engine addresses at most 255 action slots per state: the witness enumeration of VoteFor / Phase2a bounds the sizes)
SYMMETRY = [[1, 3, 2, 2, 1, 3, 1], [0, 3, 2, 2, 15, 3, 1], [0, 3, 2, 3, 15, 3, 1], [1, 4, 2, 2, 1, 3, 1], [0, 3, 3, 2, 15, 3, 1],
            [0, 3, 2, 2, 15, 1, 1], [0, 3, 2, 2, 15, 2, 1], [1, 3, 3, 3, 1, 3, 1], [0, 4, 2, 2, 15, 3, 1]]
FIXTURE = {"voting_mc": [1, 3, 2, 2, 1, 3, 1], "paxos_mc": [0, 1, 1, 2, 15, 3, 1], "paxos_3x2": [0, 3, 2, 2, 15, 3, 1],
           "paxos_3x2_b3": [0, 3, 2, 3, 15, 3, 1], "voting_3x2_b3": [1, 3, 2, 3, 1, 3, 1]}


@pytest.fixture(scope="module")
def amd():
    import tla_rust_amd
    assert tla_rust_amd.device_count() >= 1, "no HIP device visible"
    return tla_rust_amd


def _levels_text(eng, res):
    out, first = {}, 0
    for lvl, n in enumerate(res.levels, start=1):
        out[lvl] = sorted(t.replace("\n", " ") for t in eng.state_texts(first, n))
        first += n
    return out


@pytest.mark.parametrize("params", NO_SYMMETRY)
def test_same_states_per_level(amd, oracle, tmp_path, params):
    od = str(tmp_path / "o.txt")
    o = oracle.oracle_run("paxos", params, check_deadlock=False, dump=od)
    eng = amd.Engine("paxos", params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12, deadlock=False)
    r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels", "queue_left"):
        assert o[k] == r[k], k
    assert oracle.read_dump(od) == _levels_text(eng, r)
    eng.close()


@pytest.mark.parametrize("params", SYMMETRY)
def test_symmetry_orbit_counts(amd, oracle, params):
    """the engine keeps the least image (sorted acceptor blocks x value shuffles), the oracle searches all na!·nv! images by
    brute force and keeps the first state met: the orbit counts, generated counts and depth agree level by level"""
    o = oracle.oracle_run("paxos", params, check_deadlock=False)
    eng = amd.Engine("paxos", params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12, deadlock=False)
    r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels", "queue_left"):
        assert o[k] == r[k], k
    eng.close()


@pytest.mark.parametrize("name", sorted(FIXTURE))
def test_reference_text_fixture_on_gpu(amd, name):
    g = GOLD[name]
    eng = amd.Engine("paxos", FIXTURE[name], table_capacity=1 << 20, arena_capacity=1 << 18, deadlock=False)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.levels, r.verdict) == (g["distinct"], g["generated"], g["depth"], g["levels"], g["verdict"])
    eng.close()


def test_representatives_are_states_of_the_unreduced_graph(amd):
    """every representative stored under SYMMETRY is a reachable state of the model without SYMMETRY, on the same level"""
    full = amd.Engine("paxos", [0, 3, 2, 2, 15, 0, 1], table_capacity=1 << 20, arena_capacity=1 << 18, deadlock=False)
    rf = full.run()
    red = amd.Engine("paxos", [0, 3, 2, 2, 15, 3, 1], table_capacity=1 << 20, arena_capacity=1 << 18, deadlock=False)
    rr = red.run()
    tf, tr = _levels_text(full, rf), _levels_text(red, rr)
    assert rr.distinct == 443 and rf.distinct == 3921 and rr.depth == rf.depth
    for lvl in tr:
        assert set(tr[lvl]) <= set(tf[lvl])
    full.close()
    red.close()


def test_too_many_witness_slots_is_refused(amd):
    """Voting with 4 acceptors, 3 values, 3 ballots enumerates 444 (action, witness) slots per state: refused, not truncated"""
    with pytest.raises(amd.McError):
        amd.Engine("paxos", [1, 4, 3, 3, 1, 3, 1], table_capacity=1 << 16, arena_capacity=1 << 14)


def test_deadlock_is_reported_for_voting(amd, oracle):
    """Voting over the finite MCBallot = 0..1 runs out of ballots: TLC's default deadlock check reports it"""
    o = oracle.oracle_run("paxos", [1, 3, 2, 2, 1, 3, 1])
    eng = amd.Engine("paxos", [1, 3, 2, 2, 1, 3, 1], table_capacity=1 << 16, arena_capacity=1 << 14)
    r = eng.run()
    assert r.verdict == o["verdict"] == "deadlock" and r.trace_len == len(o["trace"])
    eng.close()


@pytest.mark.parametrize("name,params,index,length", [("voting_badquorum", [1, 3, 2, 2, 1, 0, 1, 3, 1, 2, 4], 1, 4),
                                                      ("paxos_bad_phase2a", [0, 3, 2, 2, 15, 0, 3], 2, 2)])
def test_negative_controls_on_gpu(amd, name, params, index, length):
    """specs/paxos/MCVotingBadQuorum.tla breaks C!Spec (reported as index 1, after the invariant), MCPaxosBad.tla breaks Inv!3"""
    g = GOLD[name]
    assert (g["index"], g["trace_len"]) == (index, length)
    eng = amd.Engine("paxos", params, table_capacity=1 << 16, arena_capacity=1 << 14, deadlock=False)
    r = eng.run()
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", index, length)
    tr = eng.trace()
    assert len(tr) == length and tr[0][0] == "Initial predicate"
    if name == "voting_badquorum":
        assert [a for a, _ in tr[1:]].count("VoteFor") >= 2     # two values are chosen by two one-acceptor quorums
    else:
        assert tr[1][0] == "Phase2a" and '"2a"' in tr[1][1]
    eng.close()


def test_front_end_on_the_model_files(amd, monkeypatch):
    """`mc specs/paxos/MCPaxos3.tla` and MCVoting3.tla: sizes read from the module text (MCAcceptor == {...}), PROPERTY and
    SYMMETRY from the cfg.  The GPU box has no reference tree: the text of Paxos.tla / Voting.tla cannot be verified there,
    the run carries the -unverified warning (tests/test_frontend.py checks the verification in the build container)."""
    monkeypatch.setenv("TLAMC_UNVERIFIED", "1")
    res, rep = amd.check_files(str(ROOT / "specs" / "paxos" / "MCPaxos3.tla"))
    assert (res.verdict, res.distinct, res.generated, res.depth) == ("ok", 443, 2697, 17)
    assert "Model checking completed. No error has been found." in rep
    assert "2697 states generated, 443 distinct states found, 0 states left on queue." in rep
    res, rep = amd.check_files(str(ROOT / "specs" / "paxos" / "MCVoting3.tla"))
    assert res.verdict == "deadlock" and "Error: Deadlock reached." in rep     # TLC's default; -deadlock turns the check off
