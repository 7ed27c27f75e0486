"""The N>1 path on CPU: world_size-2 (3, 4, 8) gloo runs of the library's level loop (tla_rust_amd/csrc/shard_loop.h, the one
`mc -gpus P` and `bench.py --gpus N` run over RCCL) with the host build of the lowerings standing in for the HIP step kernels
and torch.distributed's gloo collectives handed to the loop as its transport.  Counts must equal the oracle's (= the 1-GPU
engine's), whatever the number of ranks or the chunk size."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def run_dist(mode, world, spec, params, tmp_path, opts=None, timeout=600):
    out = tmp_path / "out.json"
    # --standalone: the launcher binds its rendezvous store to a port the kernel picks (no "find a free port, close it, hope"): several
    # of these run side by side when the CPU suite is spread over the cores (tests/conftest.py)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={world}",
           str(ROOT / "tests" / "dist_worker.py"), mode, spec, json.dumps(params), str(out), json.dumps(opts or {})]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(out.read_text())


CASES = [("atomic_add", [9], {}), ("pcal_intro", [0, 1, 20, 2], {"chunk": 300}), ("raft", [2, 2, 2, 9, 1, 1], {"chunk": 700}),
         ("raft", [3, 2, 2, 9, 1, 1], {"max_distinct": 60000, "chunk": 5000}),
         ("ssi", [2, 2, 127, 0], {"chunk": 900}),
         ("ssi", [3, 2, 127, 0, 0, 3], {"max_distinct": 50000, "chunk": 3000}),   # cfg SYMMETRY: orbit representatives are sharded like states
         # examples/Paxos/Paxos.tla, 3 acceptors x 2 values (Inv1-4 per stored state, V!Spec per transition), with and without SYMMETRY
         ("paxos", [0, 3, 2, 2, 15, 0, 1], {"chunk": 500}), ("paxos", [0, 3, 2, 2, 15, 3, 1], {"chunk": 100})]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("spec,params,opts", CASES)
def test_sharded_counts_equal_oracle(oracle, shim, tmp_path, world, spec, params, opts):
    o = oracle.oracle_run(spec, params, max_distinct=opts.get("max_distinct", 0))
    r = run_dist("shim", world, spec, params, tmp_path, opts)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
           (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
    assert sum(r["shares"]) == o["distinct"] and len(r["shares"]) == world
    if o["distinct"] > 1000:        # fingerprint ownership balances the states across ranks
        assert min(r["shares"]) > 0.6 * o["distinct"] / world


def test_sharded_verdict_propagates(oracle, shim, tmp_path):
    r = run_dist("shim", 2, "pcal_intro", [1, 0, 20, 2], tmp_path, {"chunk": 500})
    assert r["verdict"] == "assert"


@pytest.mark.parametrize("replicate_until", [0, 50])
def test_sharded_budget_stop_checks_the_last_level(oracle, shim, tmp_path, replicate_until):
    """the SI models check invariants on expansion: a sharded run cut by max_levels at the depth of a violation reports it
    (mc_shard_check_frontier on every rank's unexpanded frontier), one level earlier it reports the budget"""
    params = [2, 2, 127, 3]
    L = len(oracle.oracle_run("ssi", params)["trace"])
    r = run_dist("shim", 2, "ssi", params, tmp_path, {"chunk": 400, "max_levels": L, "replicate_until": replicate_until})
    assert r["verdict"] == "invariant" and r["depth"] == L
    r = run_dist("shim", 2, "ssi", params, tmp_path, {"chunk": 400, "max_levels": L - 1, "replicate_until": replicate_until})
    assert r["verdict"] == "budget" and r["depth"] == L - 1


@pytest.mark.parametrize("world", [2, 3])
def test_stay_mode_counts_equal_oracle(oracle, shim, tmp_path, world):
    """States stay on the generating rank once the frontier is large (threshold lowered to 50 states per rank
    here, loose rebalance ratio): only fingerprints and answers are exchanged — fixed-capacity buckets with in-band counts,
    equal-split all-to-alls, no size exchange; counts must not change."""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=60000)
    r = run_dist("shim", world, "raft", params, tmp_path, {"max_distinct": 60000, "chunk": 2000, "stay_threshold": 50, "rebalance_ratio": 1.6})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert r["phases"].get("stay_levels", 0) >= 5 and r["phases"].get("move_levels", 0) >= 3
    assert sum(r["shares"]) == o["distinct"]


@pytest.mark.parametrize("world", [2, 3])
def test_compiled_pluscal_program_sharded(shim, tmp_path, world):
    """the compiled-program path (spec_vm.h) through the same sharded exchange: counts of the 1-rank run"""
    import helpers
    path = ROOT / "specs" / "pluscal" / "cas_counter.tla"
    spec = {"path": str(path), "invariants": ["NeverTooMany", "SeenIsOld"], "constants": {"Workers": 3, "N": 2}}
    prog = helpers.ShimProgram(path.read_text(), spec["invariants"], spec["constants"])
    one = helpers.shim_run("pcal", prog.params)
    prog.close()
    r = run_dist("shim", world, "pcal_file", spec, tmp_path, {"chunk": 200, "stay_threshold": 40, "rebalance_ratio": 2.0})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
           (one["distinct"], one["generated"], one["depth"], one["levels"], one["verdict"])
    assert sum(r["shares"]) == one["distinct"] and min(r["shares"]) > 0


def test_compiled_program_with_a_set_of_records_sharded(shim, tmp_path):
    """the message soup (a set of records as sorted cells, `with m \\in msgs` by value; spec_vm.h VM_RSADD) and the channels (an array of
    sequences of records; VM_SEQSEL) through the sharded exchange at world size 2: counts of the 1-rank run, which equal the evaluators'
    (tests/test_pcal.py)"""
    import helpers
    for name, invs, consts in (("two_phase_soup", ["Consistent", "OneDecision", "PreparedWereSent", "KnownMessages", "SoupIsSmall"], {"RM": 4, "Hasty": False}),
                               ("two_phase_channels", ["Consistent", "CommitNeedsAllVotes", "InboxHoldsVotes", "FromTheCoordinator"], {"RM": 3, "Eager": False})):
        path = ROOT / "specs" / "pluscal" / f"{name}.tla"
        spec = {"path": str(path), "invariants": invs, "constants": consts}
        prog = helpers.ShimProgram(path.read_text(), invs, consts)
        one = helpers.shim_run("pcal", prog.params)
        prog.close()
        r = run_dist("shim", 2, "pcal_file", spec, tmp_path, {"chunk": 300, "stay_threshold": 40, "rebalance_ratio": 2.0})
        assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
               (one["distinct"], one["generated"], one["depth"], one["levels"], one["verdict"]), name
        assert sum(r["shares"]) == one["distinct"] and min(r["shares"]) > 0 and one["distinct"] > 4000


@pytest.mark.parametrize("world", [4, 8])
def test_eight_way_sharding_counts_equal_oracle(oracle, shim, tmp_path, world):
    """the widths the driver's scaling run uses (N = 4, 8): 8 owners, 8 x 8 count exchange, both exchange modes"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=40000)
    r = run_dist("shim", world, "raft", params, tmp_path, {"max_distinct": 40000, "chunk": 600, "stay_threshold": 30, "rebalance_ratio": 2.5})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert len(r["shares"]) == world and sum(r["shares"]) == o["distinct"] and min(r["shares"]) > 0
    assert r["phases"].get("stay_levels", 0) >= 1 and r["phases"].get("move_levels", 0) >= 3


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("until", [5, 200])
def test_replicated_prefix_then_sharded(oracle, shim, tmp_path, world, until):
    """every rank runs the small first levels itself, then takes its slice of the first large level"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=40000)
    r = run_dist("shim", world, "raft", params, tmp_path, {"max_distinct": 40000, "chunk": 700, "stay_threshold": 60, "rebalance_ratio": 2.0,
                                                           "replicate_until": until})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert sum(r["shares"]) == o["distinct"] and min(r["shares"]) > 0


def test_replicated_prefix_covers_a_whole_small_graph_and_a_violation(oracle, shim, tmp_path):
    o = oracle.oracle_run("atomic_add", [6])
    r = run_dist("shim", 2, "atomic_add", [6], tmp_path, {"replicate_until": 1 << 20})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], "ok")
    r = run_dist("shim", 2, "pcal_intro", [1, 0, 20, 2], tmp_path, {"replicate_until": 1 << 20})
    o = oracle.oracle_run("pcal_intro", [1, 0, 20, 2])
    assert r["verdict"] == "assert" and (r["distinct"], r["generated"], r["levels"]) == (o["distinct"], o["generated"], o["levels"])


def test_a_drifted_level_is_rebalanced(oracle, shim, tmp_path):
    """rebalance_ratio 1.0 (+ nothing ever balanced exactly): every large level is a MOVE level, the new states travel to their
    owners and the shares stay even; ratio 100: the same graph with stay levels only after the first large level — same counts"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=40000)
    for ratio, want_stay in ((1.0, False), (100.0, True)):
        r = run_dist("shim", 3, "raft", params, tmp_path, {"max_distinct": 40000, "chunk": 900, "stay_threshold": 40, "rebalance_ratio": ratio,
                                                           "replicate_until": 30})
        assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
        assert (r["phases"]["stay_levels"] > 0) == want_stay and sum(r["shares"]) == o["distinct"]
        if not want_stay:
            assert max(r["shares"]) < 1.15 * o["distinct"] / 3


@pytest.mark.parametrize("world,replicate_until", [(2, 0), (3, 0), (3, 40)])
def test_counterexample_walked_back_across_ranks_on_cpu(oracle, shim, tmp_path, world, replicate_until):
    """README.md:267-321 on several ranks: the behaviour that ends in the failing Assert is rebuilt by the loop's collective walk
    (mc_shard_trace_transport: parents of states that moved travelled with them) — the oracle's shortest length"""
    params = [1, 0, 20, 2]
    o = oracle.oracle_run("pcal_intro", params)
    r = run_dist("shim", world, "pcal_intro", params, tmp_path, {"chunk": 512, "trace": True, "replicate_until": replicate_until})
    assert r["verdict"] == "assert" == o["verdict"]
    tr = r["trace"]
    assert tr is not None and len(tr) == len(o["trace"]) == 6 and tr[0][0] == "Initial predicate"
    assert "alice_account = -" in tr[-1][1] and len({t for _, t in tr}) == 6


def test_invariant_counterexample_across_ranks_on_cpu(oracle, shim, tmp_path):
    """an INVARIANT violated by a successor that is stored nowhere (rebuilt from its parent) and one found when the state is
    expanded (SI models), three ranks, move and stay levels: the oracle's shortest length, a behaviour without repeats"""
    for spec, params, last in (("pcal_intro", [1, 1, 20, 2], "account_total"), ("ssi", [2, 2, 127, 3], "history")):
        o = oracle.oracle_run(spec, params)
        assert o["verdict"] == "invariant"
        r = run_dist("shim", 3, spec, params, tmp_path, {"chunk": 256, "trace": True, "stay_threshold": 40, "rebalance_ratio": 2.5})
        tr = r["trace"]
        assert r["verdict"] == "invariant" and tr is not None and len(tr) == len(o["trace"]) and len({t for _, t in tr}) == len(tr)
        assert tr[0][0] == "Initial predicate" and all(a and a != "?" for a, _ in tr) and last in tr[-1][1]


INIT_VIOL = """---- MODULE init_viol ----
EXTENDS Naturals
(* --algorithm init_viol
variables v \\in 1..8, w = 0;
begin
  A: w := v;
end algorithm *)
Small == v < 6
====
"""


@pytest.mark.parametrize("world", [2, 3])
def test_invariant_violated_by_an_initial_state_sharded(shim, tmp_path, world):
    """with Init owner-filtered over the ranks, the violation's index is the ORDINAL of the initial state, not an
    arena index of the rank that found it: the counterexample is that one state, rebuilt from the ordinal (v = 6, the first of
    1..8 that is not < 6), not whatever state sits at arena[ordinal] on that rank."""
    path = tmp_path / "init_viol.tla"
    path.write_text(INIT_VIOL)
    spec = {"path": str(path), "invariants": ["Small"], "constants": {}}
    r = run_dist("shim", world, "pcal_file", spec, tmp_path, {"chunk": 100, "trace": True})
    assert r["verdict"] == "invariant"
    assert len(r["trace"]) == 1 and r["trace"][0][0] == "Initial predicate"
    assert "v = 6" in r["trace"][0][1] or "v = 7" in r["trace"][0][1] or "v = 8" in r["trace"][0][1]
    assert "w = 0" in r["trace"][0][1]


# ---------------------------------------------------------------------------------------------- one checkpoint file per rank
@pytest.mark.parametrize("world,opts", [(2, {"max_levels": 9, "chunk": 700}),                                            # stopped in a move level
                                        (3, {"max_levels": 14, "chunk": 900, "stay_threshold": 50, "rebalance_ratio": 1.6}),   # ... in a stay level
                                        (2, {"max_levels": 3, "chunk": 700, "replicate_until": 100000})])                 # ... inside the replicated prefix
def test_checkpoint_per_rank_then_continue_in_fresh_engines(oracle, shim, tmp_path, world, opts):
    """TLC's checkpoint / -recover for the sharded engine (testout1:10): a run stopped on its budget writes ONE FILE PER RANK
    (arena, parents, the rank's seen-set slice, counters, the job's level table); fresh engines restore their files and the level
    loop continues with the unexpanded frontier — the completed run must equal the uninterrupted one (= the oracle's)"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params)
    cut = oracle.oracle_run("raft", params, max_levels=opts["max_levels"])
    r = run_dist("shim", world, "raft", params, tmp_path, dict(opts, checkpoint=str(tmp_path / "ck")))
    assert (r["first"]["verdict"], r["first"]["distinct"], r["first"]["levels"]) == ("budget", cut["distinct"], cut["levels"])
    assert all((tmp_path / f"ck.rank{k}").stat().st_size > 0 for k in range(world))
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
    assert sum(r["shares"]) == o["distinct"]


@pytest.mark.parametrize("world", [2, 3])
def test_checkpoint_of_a_finished_run_recovers_as_finished(oracle, shim, tmp_path, world):
    """mc_shard_checkpoint accepts a run that FINISHED (`mc -gpus P -checkpoint` writes it and prints TLC's
    "Checkpointing ... completed"); recovering it must report the finished result again — no frontier is left, the level loop does
    not run — instead of refusing it as "frontiers do not add up"."""
    params = [2, 1, 2, 9, 1, 1]
    o = oracle.oracle_run("raft", params)
    r = run_dist("shim", world, "raft", params, tmp_path, {"chunk": 700, "checkpoint": str(tmp_path / "ck")})
    assert (r["first"]["verdict"], r["first"]["distinct"], r["first"]["levels"]) == ("ok", o["distinct"], o["levels"])
    assert (r["distinct"], r["depth"], r["levels"], r["verdict"], r["queue_left"]) == (o["distinct"], o["depth"], o["levels"], "ok", 0)
    assert sum(r["shares"]) == o["distinct"]


def test_counterexample_after_a_restore_walks_into_the_checkpointed_part(oracle, shim, tmp_path):
    """the parent pointers travel with the checkpoint: a violation found after the restore is traced back to Init across ranks"""
    params = [1, 0, 20, 2]   # pcal_intro, the README's failing Assert (depth 7)
    o = oracle.oracle_run("pcal_intro", params)
    r = run_dist("shim", 2, "pcal_intro", params, tmp_path, {"chunk": 300, "max_levels": 4, "checkpoint": str(tmp_path / "ck"), "trace": True})
    assert r["first"]["verdict"] == "budget" and r["verdict"] == "assert"
    assert len(r["trace"]) == len(o["trace"]) and r["trace"][0][0] == "Initial predicate"


def test_restore_refuses_another_ranks_file(oracle, shim, tmp_path):
    r = run_dist("shim", 2, "raft", [2, 2, 2, 9, 2, 1], tmp_path, {"max_levels": 6, "chunk": 700, "checkpoint": str(tmp_path / "ck"), "restore_wrong_rank": True})
    assert all("another rank" in e for e in r["restore_errors"]), r["restore_errors"]


def test_ranks_that_did_not_restore_the_same_run_stop_together(oracle, shim, tmp_path):
    """rank 0 restored, rank 1 starts from Init: the level tables travel with the first all-gather and every rank refuses"""
    r = run_dist("shim", 2, "raft", [2, 2, 2, 9, 2, 1], tmp_path, {"max_levels": 6, "chunk": 700, "checkpoint": str(tmp_path / "ck"), "restore_only_rank0": True})
    assert len(r["run_errors"]) == 2 and all("same run" in e for e in r["run_errors"])


def _random_sharding(seed):
    import random
    from test_lowering_sweep import raft_config, ssi_config
    r = random.Random(3000 + seed)
    spec = r.choice(["raft", "raft", "ssi"])
    params = raft_config(r.randrange(40)) if spec == "raft" else ssi_config(r.randrange(24))
    world = r.choice([2, 3, 4, 5])
    opts = {"max_distinct": 30000, "chunk": r.choice([150, 400, 1000, 4000]), "stay_threshold": r.choice([20, 60, 400, 1 << 15]),
            "rebalance_ratio": r.choice([1.2, 1.6, 2.5]), "replicate_until": r.choice([0, 0, 30, 300])}
    # (drawn last: the seeds keep the models and settings they had before the stay levels had three forms)
    opts.update(r.choice([{}, {"exchange": "exact"}, {"exchange": "packed"}, {"exchange": "measured", "cap_safety_pct": 110}, {"exchange": "measured", "cap_safety_pct": 250}]))
    return spec, params, world, opts


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TLAMC_SWEEP", "10"))))
def test_random_model_world_and_exchange_settings(oracle, shim, tmp_path, seed):
    """seeded random combinations of a model (the raft / SI configurations of tests/test_lowering_sweep.py), a world size (2-5), the
    chunk size, the frontier size from which states stay on their rank, the rebalancing ratio and the length of the replicated prefix:
    whatever mix of replicated, move and stay levels that gives, the counters and every per-level count are the oracle's"""
    spec, params, world, opts = _random_sharding(seed)
    oparams = oracle.raft_oracle_params(params) if spec == "raft" else params
    o = oracle.oracle_run(spec, oparams, max_distinct=opts["max_distinct"])
    r = run_dist("shim", world, spec, params, tmp_path, opts)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
           (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]), (spec, params, world, opts)
    assert sum(r["shares"]) == o["distinct"] and len(r["shares"]) == world


@pytest.mark.parametrize("seed", [2, 9, 29, 34, 37] + list(range(42, 42 + int(__import__("os").environ.get("TLAMC_SWEEP", "0")))))
def test_random_violations_walked_back_across_ranks(oracle, shim, tmp_path, seed):
    """seeded: a snapshot-isolation model with one of its seven expected violations (or the README's failing Assert), 2 - 4 ranks, the
    three forms of the stay levels, stay thresholds from 10 states on: the verdict, and a behaviour of the oracle's (shortest) length that
    starts in an initial state and never repeats a state (42 combinations ran clean when this was written; five are kept here)"""
    import random
    r = random.Random(7000 + seed)
    if r.random() < 0.7:
        spec, params = "ssi", [r.choice([2, 2, 3]), r.choice([1, 2]), 127, r.randrange(1, 8)]
    else:
        spec, params = "pcal_intro", [1, r.choice([0, 1]), 20, 2]
    world = r.choice([2, 3, 4])
    opts = {"chunk": r.choice([100, 400, 2000]), "stay_threshold": r.choice([10, 40, 200]), "rebalance_ratio": r.choice([1.3, 2.5]), "replicate_until": r.choice([0, 50]),
            "exchange": r.choice(["exact", "exact", "measured", "packed"]), "trace": True}
    o = oracle.oracle_run(spec, params)
    g = run_dist("shim", world, spec, params, tmp_path, opts)
    assert g["verdict"] == o["verdict"] and o["verdict"] in ("invariant", "assert"), (spec, params, world, opts)
    tr = g["trace"]
    assert tr is not None and len(tr) == len(o["trace"]) and tr[0][0] == "Initial predicate" and len({t for _, t in tr}) == len(tr), (spec, params, world, opts)


def _random_large_rounds(seed):
    import random
    from test_lowering_sweep import raft_config, ssi_config
    r = random.Random(9000 + seed)
    spec = r.choice(["raft", "raft", "ssi"])
    params = raft_config(r.randrange(40)) if spec == "raft" else ssi_config(r.randrange(24))
    world = r.choice([2, 3, 4, 5])
    opts = {"max_distinct": 150000, "chunk": r.choice([2000, 5000, 20000]), "stay_threshold": r.choice([1024, 1500, 4000]), "rebalance_ratio": r.choice([1.3, 2.0]),
            "replicate_until": r.choice([0, 300]), "exchange": r.choice(["measured", "measured", "packed", "exact"]), "cap_safety_pct": r.choice([60, 100, 140, 200]),
            "packed_fanout": r.choice([4, 16, 16])}
    return spec, params, world, opts


# (seeds 4, 5, 14, 16, 26 of the first 36 restart: a measured bucket too small at 60 / 100 / 140 %, an allowance of 4 successors per state)
@pytest.mark.parametrize("seed", [4, 5, 14, 16, 23, 26] + list(range(36, 36 + int(__import__("os").environ.get("TLAMC_SWEEP", "0")))))
def test_random_models_with_rounds_large_enough_to_measure(oracle, shim, tmp_path, seed):
    """like the sweep above with rounds of 2 000 - 20 000 states (the fill of a bucket is only measured on rounds of >= 1 024), 150 000
    states, the three forms of the stay levels, safety margins from 60 % (the measured buckets overflow: restart) to 200 %, fan-out
    allowances of 4 and 16: whatever restarts that takes, the oracle's counters and per-level counts"""
    spec, params, world, opts = _random_large_rounds(seed)
    oparams = oracle.raft_oracle_params(params) if spec == "raft" else params
    o = oracle.oracle_run(spec, oparams, max_distinct=opts["max_distinct"])
    r = run_dist("shim", world, spec, params, tmp_path, opts, timeout=1200)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
           (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]), (spec, params, world, opts)
    assert sum(r["shares"]) == o["distinct"] and len(r["shares"]) == world
    if seed in (4, 5, 14, 16, 26):
        assert r["stats"]["restarts"] >= 1, r["stats"]


def test_a_full_exchange_bucket_restarts_the_search(oracle, shim, tmp_path):
    """more in-model successors per state than `packed_fanout` allows for (found by the sweep above: seed 38 of 70) is MC_EROUTE on
    every rank — nothing truncated, nobody left in a collective — and mc_shard_run* starts over with twice the allowance: the run
    ends with the oracle's counters and says how often it restarted; with an allowance of 1 it needs several doublings"""
    params = [3, 4, 3, 3, 2, 3, 5, 0, 0, 5]
    opts = {"max_distinct": 30000, "chunk": 1000, "stay_threshold": 60, "rebalance_ratio": 1.6, "replicate_until": 30}
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(params), max_distinct=30000)
    r = run_dist("shim", 2, "raft", params, tmp_path, dict(opts, packed_fanout=8, exchange="packed"))   # (these states have up to ~11 in-model successors)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert r["stats"]["restarts"] >= 1
    r = run_dist("shim", 3, "raft", params, tmp_path, dict(opts, packed_fanout=1, move_fanout=1, exchange="packed"))
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert r["stats"]["restarts"] >= 3


K5 = [3, 4, 2, 3, 1, 1, 0, 0, 0, 5]   # examples/raft.tla, 3 servers, MaxMsgKeys = 5: 178 654 states, complete


@pytest.mark.parametrize("world", [2, 3])
def test_the_three_forms_of_a_stay_level(oracle, shim, tmp_path, world):
    """A stay level's exchange in its three forms — fixed-capacity buckets sized from packed_fanout (MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS), the same
    buckets sized from the previous level's measured fill (MC_SHARD_PACKED, cap_safety_pct), host-paced rounds with exact sizes (the default):
    the same counters and per-level counts (the oracle's), and the bytes that crossed between ranks for fingerprints and answers are
    9 per routed candidate in the exact form, less in the measured than in the fixed form"""
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(K5))
    base = {"chunk": 4096, "stay_threshold": 1024}
    got = {}
    for form, extra in (("fixed", {"exchange": "packed"}), ("measured", {"exchange": "measured", "cap_safety_pct": 140}), ("exact", {"exchange": "exact"})):
        r = run_dist("shim", world, "raft", K5, tmp_path, dict(base, **extra))
        assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]), form
        assert r["stats"]["restarts"] == 0 and r["stats"]["stay_levels"] >= 8, (form, r["stats"])
        got[form] = r["stats"]
    assert got["exact"]["fp_answer_bytes"] == 9 * got["exact"]["routed_candidates"] and got["exact"]["measured_levels"] == 0
    assert got["fixed"]["measured_levels"] == 0 and got["measured"]["measured_levels"] >= 6
    assert got["fixed"]["routed_candidates"] == got["measured"]["routed_candidates"]   # (rank 0's: the same candidates, whatever carried them)
    assert got["exact"]["fp_answer_bytes"] < got["measured"]["fp_answer_bytes"] < got["fixed"]["fp_answer_bytes"]


@pytest.mark.parametrize("exchange", ["exact", "measured"])
def test_stay_rounds_keep_the_ranks_balanced(oracle, shim, tmp_path, exchange):
    """a state that several ranks generate in the same round stays with the rank whose candidate reaches the owner's table first: the
    host-paced stay rounds probe the sources' buckets in segments, in rotating source order (the fixed-capacity rounds walk them
    interleaved inside their kernel; the host stand-in does the same), so that no rank wins those ties systematically (probed as they
    lie, rank 0 ended with 4 x rank 3's states on the 7-key model: 1 037 687 / 630 885 / 384 312 / 251 066)"""
    params = [3, 4, 2, 3, 1, 1, 0, 0, 0, 6]
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(params))
    r = run_dist("shim", 4, "raft", params, tmp_path, {"chunk": 65536, "stay_threshold": 2048, "rebalance_ratio": 3.0, "exchange": exchange, "cap_safety_pct": 250})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
    assert r["stats"]["stay_levels"] >= 8 and max(r["shares"]) < 1.12 * o["distinct"] / 4, (r["shares"], r["stats"])


def test_a_measured_bucket_that_is_too_small_restarts_with_fixed_capacities(oracle, shim, tmp_path):
    """buckets sized at HALF of what the previous level's fullest bucket held per state cannot hold the next level's: MC_EROUTE on every
    rank, and the search that is started over sizes its buckets from (twice) packed_fanout alone — the oracle's counters, one restart"""
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(K5))
    r = run_dist("shim", 2, "raft", K5, tmp_path, {"chunk": 65536, "stay_threshold": 1024, "exchange": "measured", "cap_safety_pct": 50})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
    assert r["stats"]["restarts"] == 1 and r["stats"]["measured_levels"] == 0, r["stats"]


SMALL_COMPLETE = [("raft", [2, 1, 2, 9, 1, 1]), ("raft", [2, 2, 2, 9, 1, 1]), ("raft", [2, 1, 2, 9, 2, 1, 6, 0, 0, 6]), ("ssi", [2, 2, 127, 0]), ("ssi", [3, 1, 127, 0, 1]),
                  ("pcal_intro", [0, 1, 20, 2]), ("atomic_add", [9]), ("paxos", [0, 3, 2, 2, 15, 0, 1]), ("paxos", [0, 3, 2, 2, 15, 3, 1])]


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TLAMC_SWEEP", "8"))))
def test_checkpoint_at_a_random_level_of_a_random_run(oracle, shim, tmp_path, seed):
    """seeded random: a complete model, a world size (2-4), exchange settings, and the BFS level the run is stopped at; one checkpoint
    file per rank, fresh engines, restore, continue — the completed run is the uninterrupted one (= the oracle's), wherever the cut
    falls (replicated prefix, move level, stay level) and whatever the ranks held at that moment"""
    import random
    r = random.Random(4000 + seed)
    spec, params = r.choice(SMALL_COMPLETE)
    world = r.choice([2, 3, 4])
    kw = {"check_deadlock": False} if spec == "paxos" else {}
    oparams = oracle.raft_oracle_params(params) if spec == "raft" else params
    o = oracle.oracle_run(spec, oparams, **kw)
    cut_level = r.randrange(2, max(3, o["depth"] - 1))
    cut = oracle.oracle_run(spec, oparams, max_levels=cut_level, **kw)
    opts = {"max_levels": cut_level, "chunk": r.choice([100, 300, 900]), "stay_threshold": r.choice([10, 40, 1 << 15]), "rebalance_ratio": r.choice([1.3, 2.0]),
            "replicate_until": r.choice([0, 0, 20, 500]), "checkpoint": str(tmp_path / "ck")}
    got = run_dist("shim", world, spec, params, tmp_path, opts)
    assert (got["first"]["verdict"], got["first"]["distinct"], got["first"]["levels"]) == ("budget", cut["distinct"], cut["levels"]), (spec, params, world, opts)
    assert (got["distinct"], got["generated"], got["depth"], got["levels"], got["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"]), (spec, params, world, opts)
    assert sum(got["shares"]) == o["distinct"]


def _run_expect(world, spec, params, tmp_path, opts, fail_at):
    """every rank's own return code of mc_shard_run_transport with $TLAMC_TEST_FAIL_AT set (include/tlamc.h, test-only hooks)"""
    import os
    out = tmp_path / "out.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={world}",
           str(ROOT / "tests" / "dist_worker.py"), "shim", spec, json.dumps(params), str(out), json.dumps(dict(opts, expect_error=True))]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, TLAMC_TEST_FAIL_AT=fail_at))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]   # (a hang in a collective would be the timeout above)
    return [json.loads((tmp_path / f"out.json.rank{r}").read_text()) for r in range(world)]


def test_ranks_failing_differently_in_one_level_agree_on_one_code(oracle, shim, tmp_path):
    """rank 0 fails level 4 with MC_EROUTE (-10: restart with twice the allowance), rank 1 fails the SAME level
    with MC_EARENA (-5).  Every rank must leave the loop with the same code — the one that is not a restart — instead of one rank
    restarting into collectives the other never enters (a hang).  Three ranks: the third did not fail at all and still agrees."""
    for world, fail_at in ((2, "0:4:-10,1:4:-5"), (3, "2:4:-10,1:4:-4")):
        res = _run_expect(world, "raft", [2, 2, 2, 9, 1, 1], tmp_path, {"chunk": 700, "stay_threshold": 50}, fail_at)
        want = -5 if world == 2 else -4
        assert [r["code"] for r in res] == [want] * world, res


def test_a_route_failure_on_one_rank_restarts_every_rank(oracle, shim, tmp_path):
    """... and when the only failure of the level is MC_EROUTE — on ONE rank — every rank starts over (restarts = 1 on each) and the
    search ends with the oracle's counters"""
    params = [2, 2, 2, 9, 1, 1]
    o = oracle.oracle_run("raft", params)
    res = _run_expect(2, "raft", params, tmp_path, {"chunk": 700, "stay_threshold": 50}, "1:4:-10")
    assert [r["code"] for r in res] == [0, 0] and [r["restarts"] for r in res] == [1, 1], res
    assert all((r["distinct"], r["generated"], r["levels"]) == (o["distinct"], o["generated"], o["levels"]) for r in res), res
