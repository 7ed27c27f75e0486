"""The CPU oracle against every golden vector the reference holds for this path
and against the survey-derived regression anchors.  CPU only."""
import json
from pathlib import Path

import pytest

GOLDEN = Path(__file__).parent / "golden"


def test_readme_tlc_run_reproduced_exactly(oracle):
    """README.md:267-321 — the only real TLC output for an in-scope spec: README variant of
    pcal_intro (labels A:/B:), no invariant, TLC stops at the first Assert failure."""
    r = oracle.oracle_run("pcal_intro", [1, 0, 20, 2], stop=2)
    assert r["verdict"] == "assert"
    assert (r["generated"], r["distinct"], r["queue_left"]) == (9097, 6164, 999)   # README.md:319
    assert r["depth"] == 7                                                           # README.md:320
    golden = json.loads((GOLDEN / "readme_pcal_intro_trace.json").read_text())
    assert len(r["trace"]) == 6
    for (act, text), g in zip(r["trace"], golden["states"]):
        got = dict(line[3:].split(" = ", 1) for line in text.split("\n"))
        assert got == g                     # same six states, variable by variable (README.md:272-311)
    assert "alice_account = -1" in r["trace"][-1][1]


def test_committed_pcal_intro_passes(oracle):
    """README.md:349-352: with the labels removed 'Re-running tlc should produce no errors';
    pcal_intro.cfg:2-3 checks MoneyInvariant."""
    r = oracle.oracle_run("pcal_intro", [0, 1, 20, 2])
    assert r["verdict"] == "ok"
    assert (r["distinct"], r["generated"], r["depth"]) == (3800, 5850, 5)
    assert r["levels"][0] == 400            # money \in [1..2 -> 1..20]


def test_readme_variant_violates_money_invariant(oracle):
    r = oracle.oracle_run("pcal_intro", [1, 1, 20, 2])
    assert r["verdict"] == "invariant" and len(r["trace"]) == 3   # Init, Transfer, A: alice dropped, bob not yet credited


@pytest.mark.parametrize("n", [1, 2, 3, 4, 10, 16])
def test_atomic_add_closed_form(oracle, n):
    r = oracle.oracle_run("atomic_add", [n])
    assert r["verdict"] == "ok"
    assert r["distinct"] == 2 ** n + 1
    assert r["generated"] == n * 2 ** (n - 1) + 3
    assert r["depth"] == n + 2


RAFT_ANCHORS = [  # BASELINE.md §2 (survey-derived, independent implementation)
    ([2, 1, 2, 9, 1, 1], 6128, 51949, 22),
    ([2, 2, 2, 9, 1, 1], 13634, 104515, 34),
    ([2, 2, 2, 9, 2, 1], 270972, 2277995, 46),
    ([2, 3, 2, 9, 1, 3], 88490, 575389, 45),
]


@pytest.mark.parametrize("params,d,g,depth", RAFT_ANCHORS)
def test_raft_anchors(oracle, params, d, g, depth):
    r = oracle.oracle_run("raft", params)
    assert r["verdict"] == "ok"
    assert (r["distinct"], r["generated"], r["depth"]) == (d, g, depth)


def test_raft_naive_commit_lowering_is_detected(oracle):
    """lowering raft.tla:392-402 as an unconditional assignment
    changes the reachable set (13 634 -> 15 794)."""
    r = oracle.oracle_run("raft", [2, 2, 2, 9, 1, 1, 1])
    assert (r["distinct"], r["generated"], r["depth"]) == (15794, 118339, 36)


def test_raft_golden_levels(oracle):
    g = json.loads((GOLDEN / "raft_levels.json").read_text())
    for case in g["cases"]:
        if case["distinct"] > 400000:
            continue                        # the big prefixes are checked on the GPU only
        r = oracle.oracle_run("raft", case["params"], max_distinct=case.get("max_distinct", 0))
        assert r["levels"] == case["levels"], case["name"]
        assert r["generated"] == case["generated"]


# ------------------------------------------------------------------ serializableSnapshotIsolation.tla
SSI_ANCHORS = [  # BASELINE.md §2 / SURVEY.md §6 (survey-derived, independent implementation)
    ([2, 1], 569, 945, 9), ([2, 2], 29629, 50121, 13), ([3, 1], 90430, 152554, 13),
]


@pytest.mark.parametrize("params,d,g,depth", SSI_ANCHORS)
def test_ssi_anchors_all_invariants_hold(oracle, params, d, g, depth):
    """serializableSnapshotIsolation.tla:61-79 'Should NEVER be violated' — all seven checked on every state."""
    r = oracle.oracle_run("ssi", params + [127, 0])
    assert r["verdict"] == "ok"
    assert (r["distinct"], r["generated"], r["depth"]) == (d, g, depth)


def test_ssi_4x3_prefix_levels(oracle):
    """per-level distinct counts of the 4 txns x 3 keys model."""
    r = oracle.oracle_run("ssi", [4, 3, 127, 0], max_levels=7)
    assert r["levels"] == [1, 4, 32, 264, 2532, 24576, 236844]


def test_ssi_in_spec_unit_tests(oracle):
    """serializableSnapshotIsolation.tla:1068-1077 (9 cycle-finder cases), :1184-1205 (10 well-formedness cases) and
    textbookSnapshotIsolation.tla:1231-1263 (Fekete's read-only anomaly: ReadOnlyAnomaly(h) holds, the transaction is T_3)."""
    assert oracle.oracle_lib().oracle_ssi_unit_tests() == 0


@pytest.mark.parametrize("find,trace_len", [(1, 3), (2, 6), (3, 7), (4, 12), (5, 12), (6, 9), (7, 7)])
def test_ssi_expected_violations_are_reachable(oracle, find, trace_len):
    """:81-96 'EXPECTED to be violated': every abort reason and two simultaneous lock waiters are reachable (3 txns x 2 keys)."""
    r = oracle.oracle_run("ssi", [3, 2, 127, find])
    assert r["verdict"] == "invariant" and r["violated_invariant"] == 7 and len(r["trace"]) == trace_len


def test_textbook_si_is_not_serializable_and_both_formulations_agree(oracle):
    """examples/textbookSnapshotIsolation.tla (SSI minus Cahill's variables): snapshot isolation admits write skew.
    3 txns x 2 keys is the smallest model (a key must be committed before it can be read, :365-378); the reference
    asks that Cahill's and Bernstein's formulations be equivalent (:84-89): same shortest counterexample length."""
    a = oracle.oracle_run("ssi", [3, 2, 32, 0, 1])
    b = oracle.oracle_run("ssi", [3, 2, 64, 0, 1])
    assert (a["verdict"], a["violated_invariant"], len(a["trace"])) == ("invariant", 5, 13)
    assert (b["verdict"], b["violated_invariant"], len(b["trace"])) == ("invariant", 6, 13)
    assert a["distinct"] == b["distinct"] == 16559944
    ok = oracle.oracle_run("ssi", [2, 2, 127, 0, 1])       # too small for write skew: everything holds
    assert ok["verdict"] == "ok" and ok["distinct"] == 29629


@pytest.mark.parametrize("params", [[2, 2, 127, 0, 0], [3, 1, 127, 0, 0], [2, 3, 127, 0, 0], [3, 1, 31, 0, 1]])
@pytest.mark.parametrize("sym", [1, 2, 3])
def test_symmetry_orbit_counts_do_not_depend_on_the_representative(oracle, params, sym):
    """TLC stores and expands the orbit member it met FIRST, the oracle (and the device lowering) the
    canonical one; Commit's AbortOpSeq (serializableSnapshotIsolation.tla:465-474) CHOOSEs an order, so the two could in
    principle count different numbers of orbits.  They do not: with sym bit 2 the oracle runs TLC's scheme (seen-set keyed
    by the canonical form, the generated state stored) — same orbits, same generated count, same depth, level by level.
    (3 x 2 under both symmetry sets: 6 734 049 orbits / 11 514 563 generated / 19 levels either way)"""
    a = oracle.oracle_run("ssi", params + [sym])
    b = oracle.oracle_run("ssi", params + [sym | 4])
    assert (a["distinct"], a["generated"], a["depth"], a["levels"], a["verdict"]) == \
           (b["distinct"], b["generated"], b["depth"], b["levels"], b["verdict"])
    assert a["distinct"] <= oracle.oracle_run("ssi", params)["distinct"]   # (Key symmetry cannot reduce a one-key model)
