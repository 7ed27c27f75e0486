"""GPU tests of the PlusCal CHANNELS (arrays of sequences, sequences of records: specs/pluscal/two_phase_channels.tla, mailboxes.tla;
spec_vm.h VM_SEQSEL / VM_SEQLEN, pcal.cpp RecordFlattener) and of the message SOUP (a set of records: two_phase_soup.tla; VM_RSADD / VM_RSDEL / VM_RSHAS).  Added late in round 5; like tests/test_gpu_zz_ms_queue.py the file sorts
behind every other GPU file so that under `pytest -x` a surprise here cannot keep the rest of the suite from running."""
import json
import os
from pathlib import Path

import pytest

from test_gpu_pcal import ROOT, amd, cfg_text, check_compiled_program_on_gpu, run_mc  # noqa: F401  (amd: the fixture)
from test_pcal import CASES, CHANNEL_STEMS

pytestmark = pytest.mark.gpu
CH = [c for c in CASES if c[0].stem in CHANNEL_STEMS]
GOLDEN = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())
# (round 6: six of these entries also exist ORACLE-made — oracle/tlaplus.py on hand-written translations, tests/golden/pcal_oracle.json — and
#  are the ones compared with where they exist; tests/test_pcal.py asserts that the two files agree)
for _case, _o in json.loads((ROOT / "tests" / "golden" / "pcal_oracle.json").read_text()).items():
    GOLDEN[_case] = {**GOLDEN[_case], **_o}   # (counts, levels and `source` from the oracle; RM / seq_cells from the product-made entry)
INVS = ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator", "AtMostTwoWaiting"]


@pytest.mark.parametrize("path,invs,consts", CH, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_channels_compiled_program_on_gpu_vs_tla_evaluator(amd, path, invs, consts):  # noqa: F811
    """counters, verdict, trace length and the SET of states of every BFS level equal those of oracle/tla_eval.py on the translation"""
    check_compiled_program_on_gpu(amd, path, invs, consts)


def test_three_mailboxes_with_four_cells_per_sequence(amd, monkeypatch):  # noqa: F811
    """$TLAMC_PCAL_SEQ = 4: three mailboxes of three-field records + two more sequence variables fit the 128 cells of a state (deadlock:
    a node that has its pong leaves without answering its left neighbour's ping)"""
    monkeypatch.setenv("TLAMC_PCAL_SEQ", "4")
    check_compiled_program_on_gpu(amd, ROOT / "specs" / "pluscal" / "mailboxes.tla", ["LogOk", "Pongs", "HeardTheLeft"], {"N": 3})


@pytest.mark.parametrize("case", ["two_phase_channels_rm4", "two_phase_channels_rm5"])
def test_two_phase_commit_larger_models_equal_the_record_valued_translation(amd, monkeypatch, case):  # noqa: F811
    """4 / 5 resource managers (98 945 / 2 848 539 states): per-level counts of the HAND-WRITTEN pcal2tla-style translation
    (tests/golden/pcal_records/TwoPhaseChannels.tla: chan one function to sequences of records) evaluated by the product's host
    evaluator tlaeval.cpp — another text, another engine — tests/golden/pcal_channels.json"""
    g = GOLDEN[case]
    monkeypatch.setenv("TLAMC_PCAL_SEQ", str(g["seq_cells"]))
    prog = amd.Program((ROOT / "specs" / "pluscal" / "two_phase_channels.tla").read_text(), cfg_text(INVS, {"RM": g["RM"], "Eager": False}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 24, arena_capacity=1 << 22, chunk_states=1 << 16)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (g["distinct"], g["generated"], g["depth"], "ok", 0)
    assert list(r.levels) == g["levels"]
    eng.close()
    prog.close()


def test_mc_on_two_phase_commit_over_channels():
    """`mc two_phase_channels.tla` = tlc on the message-passing protocol; the eager coordinator (two_phase_channels_eager.cfg) commits on the
    first yes vote.  The pass that finds it holds TWO errors — a manager that voted no receives the commit (the assert in Act) while a
    successor of another state breaks Consistent — and which one a parallel search reports is not defined (test_gpu_pcal.same_outcome);
    either way: exit code 12, 1 051 states, a trace from the initial state"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "two_phase_channels.tla")
    assert rc == 0, err
    assert "11905 states generated, 4523 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 28." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "two_phase_channels.tla", "-config", ROOT / "specs" / "pluscal" / "two_phase_channels_eager.cfg")
    assert rc == 12, err
    assert ("Error: Invariant Consistent is violated." in out or "The first argument of Assert evaluated to FALSE" in out) and "State 1: <Initial predicate>" in out
    assert "1051 distinct states found" in out
    assert 'chan_type = (0 :> <<' in out and '"commit"' in out     # the channels print as TLC prints a function on 0..RM of sequences


SOUP_INVS = ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"]


@pytest.mark.parametrize("case", ["two_phase_soup_rm6", "two_phase_soup_rm7"])
def test_message_soup_larger_models_equal_the_host_evaluator(amd, monkeypatch, case):  # noqa: F811
    """two-phase commit with a message soup, 6 / 7 resource managers (251 051 / 1 725 467 states): per-level counts of the module's text — msgs
    one set-valued variable, as pcal2tla keeps it — under the product's host evaluator tlaeval.cpp (tests/golden/pcal_channels.json); the
    compiled program keeps the set as sorted cells"""
    g = GOLDEN[case]
    monkeypatch.setenv("TLAMC_PCAL_SEQ", str(g["seq_cells"]))
    prog = amd.Program((ROOT / "specs" / "pluscal" / "two_phase_soup.tla").read_text(), cfg_text(SOUP_INVS, {"RM": g["RM"], "Hasty": False}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 24, arena_capacity=1 << 22, chunk_states=1 << 16)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (g["distinct"], g["generated"], g["depth"], "ok", 0)
    assert list(r.levels) == g["levels"]
    eng.close()
    prog.close()


def test_mc_on_the_message_soup():
    """`mc two_phase_soup.tla` = tlc on Lamport-style two-phase commit; a hasty transaction manager (two_phase_soup_hasty.cfg) breaks Consistent"""
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "two_phase_soup.tla")
    assert rc == 0, err
    assert "2300 states generated, 827 distinct states found, 0 states left on queue." in out
    assert "The depth of the complete state graph search is 12." in out
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "two_phase_soup.tla", "-config", ROOT / "specs" / "pluscal" / "two_phase_soup_hasty.cfg")
    assert rc == 12, err
    assert "Error: Invariant Consistent is violated." in out and "State 1: <Initial predicate>" in out
    assert '/\\ msgs = {[rm |-> 0, type |-> "commit"], [rm |-> ' in out      # a set of records, fields in name order


def test_random_algorithms_with_channels_on_gpu(amd, tmp_path):  # noqa: F811
    """40 of the seeded random algorithms of tests/test_pcal_fuzz.py ChanGen (an array of sequences, a sequence of records and a set of records
    inside random steps) through the HIP engine against the evaluator on the translation"""
    import helpers
    from test_pcal_fuzz import ChanGen, MAX_STATES
    checked = 0
    for seed in range(40):
        text, invs = ChanGen(seed).program()
        try:
            helpers.pcal_translate(text)
            prog = helpers.ShimProgram(text, invs, {})
        except RuntimeError:
            continue
        try:
            n = helpers.shim_run("pcal", prog.params)["distinct"]
        finally:
            prog.close()
        if n > MAX_STATES:
            continue
        path = tmp_path / f"Fz{seed}.tla"
        path.write_text(text.replace("MODULE Fz ", f"MODULE Fz{seed} "))
        check_compiled_program_on_gpu(amd, path, invs, {})
        checked += 1
    assert checked >= 30


def test_paxos_in_pluscal_on_gpu(amd, monkeypatch):  # noqa: F811
    """specs/pluscal/paxos_soup.tla (single-decree Paxos over a set of five-field records, 16 cells): 15 993 states / 58 405 generated / depth 25 —
    what oracle/tla_eval.py (7 569-state variant, state sets) and tlaeval.cpp give on the host (tests/test_pcal.py); the forgetful proposer
    breaks Agreement at depth 21; `mc` reads CHECK_DEADLOCK FALSE from the cfg (the acceptors never stop)"""
    monkeypatch.setenv("TLAMC_PCAL_SEQ", "16")
    spec = ROOT / "specs" / "pluscal" / "paxos_soup.tla"
    invs = ["Agreement", "VotesAreProposed", "OneValuePerBallot", "PromisesAreHonest"]
    prog = amd.Program(spec.read_text(), cfg_text(invs, {"NA": 3, "NB": 2, "NV": 2, "Forgetful": False}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12, deadlock=False)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (15993, 58405, 25, "ok", 0)
    eng.close()
    prog.close()
    prog = amd.Program(spec.read_text(), cfg_text(["Agreement", "VotesAreProposed", "PromisesAreHonest"], {"NA": 3, "NB": 2, "NV": 2, "Forgetful": True}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12, deadlock=False)
    r = eng.run()
    assert (r.verdict, r.trace_len, r.distinct) == ("invariant", 21, 16533) and prog.invariant(r.violated_invariant) == "Agreement"
    assert len(eng.trace()) == 21
    eng.close()
    prog.close()
    rc, out, err = run_mc(spec)
    assert rc == 0, err
    assert "58405 states generated, 15993 distinct states found, 0 states left on queue." in out
    rc, out, err = run_mc(spec, "-config", ROOT / "specs" / "pluscal" / "paxos_soup_forgetful.cfg")
    assert rc == 12 and "Error: Invariant Agreement is violated." in out, err


def test_epoch_based_reclamation_on_gpu(amd):  # noqa: F811
    """specs/pluscal/epoch_gc.tla, three threads: 1 380 120 states = tlaeval.cpp on module + cfg (tests/golden/pcal_channels.json); `mc` with one
    epoch of grace: NoDanglingReader is violated (a 14-state behaviour)"""
    g = GOLDEN["epoch_gc_n3"]
    invs = ["HeadIsLive", "NoDanglingReader", "EpochInRange"]
    prog = amd.Program((ROOT / "specs" / "pluscal" / "epoch_gc.tla").read_text(), cfg_text(invs, {"N": 3, "Grace": 2}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 24, arena_capacity=1 << 21, chunk_states=1 << 16)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, list(r.levels)) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])
    eng.close()
    prog.close()
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "epoch_gc.tla", "-config", ROOT / "specs" / "pluscal" / "epoch_gc_one_grace.cfg")
    assert rc == 12 and "Error: Invariant NoDanglingReader is violated." in out and "State 14:" in out, err


def test_io_buffer_on_gpu(amd):  # noqa: F811
    """specs/pluscal/io_buffer.tla, four writers, two slots: 539 320 states = tlaeval.cpp on module + cfg; `mc` on the hasty flusher: the assert
    in Copy fails"""
    g = GOLDEN["io_buffer_n4"]
    invs = ["HeaderInRange", "SealedIsFull", "FlushedFull"]
    prog = amd.Program((ROOT / "specs" / "pluscal" / "io_buffer.tla").read_text(), cfg_text(invs, {"N": 4, "Cap": 2, "Patient": True}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 23, arena_capacity=1 << 20, chunk_states=1 << 15)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, list(r.levels)) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])
    eng.close()
    prog.close()
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "io_buffer.tla", "-config", ROOT / "specs" / "pluscal" / "io_buffer_hasty.cfg")
    assert rc == 12 and "The first argument of Assert evaluated to FALSE" in out, err


def test_radix_tree_on_gpu(amd):  # noqa: F811
    """specs/pluscal/radix_tree.tla, four inserters: 3 411 041 states = tlaeval.cpp (tests/golden/pcal_channels.json); `mc` on the plain-store
    variant: an inserted key is not found"""
    g = GOLDEN["radix_tree_n4"]
    invs = ["InsertedKeysAreFound", "NoLeak", "ChildrenAreNodes"]
    prog = amd.Program((ROOT / "specs" / "pluscal" / "radix_tree.tla").read_text(), cfg_text(invs, {"N": 4, "Plain": False}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 25, arena_capacity=1 << 22, chunk_states=1 << 17)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, list(r.levels)) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])
    eng.close()
    prog.close()
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "radix_tree.tla", "-config", ROOT / "specs" / "pluscal" / "radix_tree_plain.cfg")
    assert rc == 12 and "Error: Invariant InsertedKeysAreFound is violated." in out, err


def test_pagecache_on_gpu(amd):  # noqa: F811
    """specs/pluscal/pagecache.tla, three threads: 20 254 597 states / 47 629 297 generated / depth 37 (tests/golden/pcal_channels.json); `mc` on
    the blind consolidation: Conservation is violated after 14 states"""
    g = GOLDEN["pagecache_n3"]
    invs = ["Conservation", "HeadIsAllocated"]
    prog = amd.Program((ROOT / "specs" / "pluscal" / "pagecache.tla").read_text(), cfg_text(invs, {"N": 3, "Blind": False}))
    eng = amd.Engine("pcal", prog.params, table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 20, trace=False)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.verdict, list(r.levels)) == (g["distinct"], g["generated"], g["depth"], "ok", g["levels"])
    eng.close()
    prog.close()
    rc, out, err = run_mc(ROOT / "specs" / "pluscal" / "pagecache.tla", "-config", ROOT / "specs" / "pluscal" / "pagecache_blind.cfg")
    assert rc == 12 and "Error: Invariant Conservation is violated." in out and "State 14:" in out, err
