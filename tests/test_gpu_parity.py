"""GPU parity tests (run on the MI355X box with -m gpu): the HIP engine, called through the C ABI
(include/tlamc.h via tla_rust_amd.binding), against the CPU oracle on the same configurations, and
against the committed golden fixtures at the bench workload's size.  Integer / set semantics:
every comparison is exact."""
import json
from collections import Counter
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).parent / "golden"

SMALL = [
    ("atomic_add", [2]), ("atomic_add", [3]), ("atomic_add", [4]), ("atomic_add", [11]),
    ("pcal_intro", [0, 1, 20, 2]), ("pcal_intro", [1, 0, 20, 2]), ("pcal_intro", [1, 1, 20, 2]), ("pcal_intro", [0, 1, 7, 3]),
    ("raft", [2, 1, 2, 9, 1, 1]), ("raft", [2, 2, 2, 9, 1, 1]), ("raft", [2, 3, 2, 9, 1, 3]),
    ("ssi", [2, 1, 127, 0]), ("ssi", [2, 2, 127, 0]), ("ssi", [3, 1, 127, 0]),
    ("ssi", [2, 2, 127, 0, 1]), ("ssi", [3, 1, 31, 0, 1]),       # textbookSnapshotIsolation.tla
    # cfg SYMMETRY (serializableSnapshotIsolation.tla:38-44; bit 0 TxnId, bit 1 Key): same orbit representatives as the
    # oracle's brute-force search over all permutations, level by level
    ("ssi", [2, 2, 127, 0, 0, 1]), ("ssi", [2, 2, 127, 0, 0, 2]), ("ssi", [2, 2, 127, 0, 0, 3]), ("ssi", [3, 1, 127, 0, 0, 3]),
    ("ssi", [2, 2, 127, 0, 1, 3]),
]


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


@pytest.mark.parametrize("spec,params", SMALL)
def test_engine_equals_oracle_states_and_counts(amd, oracle, tmp_path, spec, params):
    od = str(tmp_path / "o.txt")
    o = oracle.oracle_run(spec, params, dump=od)
    eng = amd.Engine(spec, params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels", "queue_left"):
        assert o[k] == r[k], k
    assert len(o["trace"]) == r.trace_len
    # the same SET of states on every BFS level (canonical TLA+ text, independent printers)
    assert oracle.read_dump(od) == _levels_text(eng, r)
    if r.verdict != "ok":
        tr = eng.trace()
        assert len(tr) == r.trace_len
        assert tr[0][0] == "Initial predicate" and tr[0][1] == o["trace"][0][1] or spec == "pcal_intro"
    eng.close()


def test_readme_counterexample_on_gpu(amd):
    """README.md:267-311: shortest counterexample has 6 states and ends with alice_account = -1."""
    eng = amd.Engine("pcal_intro", [1, 0, 20, 2], table_capacity=1 << 16, arena_capacity=1 << 14)
    r = eng.run()
    assert r.verdict == "assert" and r.trace_len == 6 and r.depth == 7
    tr = eng.trace()
    # any shortest counterexample is acceptable (TLC's own choice depends on its worker order):
    # 6 states, the last one with a negative alice_account while a process sits at C
    assert tr[0][0] == "Initial predicate" and len(tr) == 6
    import re
    assert int(re.search(r"alice_account = (-?\d+)", tr[-1][1]).group(1)) < 0 and '"C"' in tr[-1][1]
    assert sorted(a for a, _ in tr[1:]) == ["A", "A", "B", "Transfer", "Transfer"]
    # each step changes exactly what the action may change: pc of one process
    eng.close()


@pytest.mark.parametrize("chunk", [256, 1 << 10, 1 << 16])
def test_chunking_does_not_change_counts(amd, oracle, chunk):
    o = oracle.oracle_run("raft", [2, 2, 2, 9, 2, 1])
    eng = amd.Engine("raft", [2, 2, 2, 9, 2, 1], table_capacity=1 << 21, arena_capacity=1 << 19, chunk_states=chunk)
    r = eng.run()
    assert (r.distinct, r.generated, r.depth, r.levels) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    r2 = eng.run()          # an engine is reusable: same answer again
    assert (r2.distinct, r2.generated, r2.levels) == (r.distinct, r.generated, r.levels)
    eng.close()


@pytest.mark.parametrize("n", [16, 20, 24])
def test_atomic_add_closed_form_on_gpu(amd, n):
    """D = 2^N + 1, G = N*2^(N-1) + 3, depth N + 2; every level is C(N,k)-shaped."""
    eng = amd.Engine("atomic_add", [n], table_capacity=1 << (n + 2), arena_capacity=(1 << n) + 4096, chunk_states=1 << 20, trace=False)
    r = eng.run()
    assert (r.verdict, r.distinct, r.generated, r.depth) == ("ok", 2 ** n + 1, n * 2 ** (n - 1) + 3, n + 2)
    from math import comb
    assert r.levels[: n + 1] == [comb(n, k) for k in range(n + 1)] and r.levels[n + 1] == 1
    eng.close()


def test_raft_expected_violation_on_gpu(amd):
    """CommittedLogStable is violated at MaxTerm = 3, MaxClientRequests = 3;
    shortest counterexample = 31 states (oracle: tests/test_lowering_vs_oracle.py)."""
    eng = amd.Engine("raft", [2, 3, 3, 9, 1, 2], table_capacity=1 << 26, arena_capacity=1 << 25, chunk_states=1 << 18)
    r = eng.run()
    assert r.verdict == "invariant" and r.violated_invariant == 1 and r.trace_len == 31
    tr = eng.trace()
    assert len(tr) == 31 and "committedLogDecrease = TRUE" in tr[-1][1] and "committedLogDecrease = FALSE" in tr[-2][1]
    assert tr[-1][0] == "AdvanceCommitIndex"
    eng.close()


def _golden(name):
    g = json.loads((GOLDEN / "raft_levels.json").read_text())
    return next(c for c in g["cases"] if c["name"] == name)


@pytest.mark.parametrize("name", ["raft2_mcr2_t2_m2", "raft3_mcr2_t2_m1_prefix", "raft3_mcr4_t2_m1_prefix_small", "raft2_mcr1_t3_m1",
                                  "raft3_mcr4_t2_m1_bench", "raft5_mcr6_t2_m1_prefix", "raft3_mcr4_t3_m2_prefix"])
def test_raft_golden_levels_on_gpu(amd, name):
    """Committed oracle fixtures (tests/golden/make_golden.py), incl. the bench workload's full size."""
    try:
        c = _golden(name)
    except StopIteration:
        pytest.skip(f"fixture {name} not generated")
    cap = max(1 << 20, 2 * c["distinct"])
    eng = amd.Engine("raft", c["params"], table_capacity=4 * c["distinct"], arena_capacity=cap, chunk_states=1 << 19,
                     max_distinct=c["max_distinct"], trace=False)
    r = eng.run()
    assert r.levels == c["levels"]
    assert (r.distinct, r.generated, r.depth, r.verdict) == (c["distinct"], c["generated"], c["depth"], c["verdict"])
    eng.close()


@pytest.mark.parametrize("name,caps", [("raft3_mcr4_t2_m1_k5_complete", (0, 0, 0)), ("raft3_mcr4_t2_m1_k7_complete", (7, 1, 4)),
                                       ("raft3_mcr4_t3_m2_k5_complete", (0, 0, 0))])
def test_raft_complete_graphs_on_gpu(amd, oracle, name, caps):
    """COMPLETE 3-server graphs (verdict ok, nothing left on the queue): StateConstraint's MaxMsgKeys conjunct makes the graph
    finite; per-level counts of the oracle's exact-dedup run (tests/golden/raft_levels.json)."""
    c = _golden(name)
    params = oracle.raft_device_params(c["params"], *caps)
    eng = amd.Engine("raft", params, table_capacity=2 * c["distinct"], arena_capacity=c["distinct"] + (1 << 16), chunk_states=1 << 19, trace=False)
    r = eng.run()
    assert r.levels == c["levels"]
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (c["distinct"], c["generated"], c["depth"], "ok", 0)
    eng.close()


def test_bench_workload_complete_on_gpu(amd, oracle):
    """rounds 1-2's bench workload (`bench.py --workload k10`) = specs/MCraft.cfg: the COMPLETE graph of raft.tla with 3 servers and
    MaxTerm = 2, 102 586 254 states, with the slot capacities bench.py uses (10 / 1 / 4 = the oracle's maxima; W = 128 B since the
    compact layout of round 3, 288 B before) and a seen-set load of 0.76."""
    c = _golden("raft3_mcr4_t2_m1_k10_complete")
    params = oracle.raft_device_params(c["params"], 10, 1, 4)
    assert params == [3, 4, 2, 3, 1, 1, 10, 1, 4, 10] and amd.state_bytes("raft", params) == 128
    assert c["max_stat"][:3] == [10, 1, 4]
    eng = amd.Engine("raft", params, table_capacity=1 << 27, arena_capacity=c["distinct"] + (1 << 16), chunk_states=1 << 22, trace=False)
    r = eng.run()
    assert r.levels == c["levels"]
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (102586254, 1217433925, 33, "ok", 0)
    eng.close()


def test_contract_bench_workload_complete_on_gpu(amd, oracle):
    """bench.py's DEFAULT workload since round 3 = specs/MCraft_t3.cfg: MaxTerm = 3 (two elections, leader changes, conflict-truncate
    reachable), MaxMsgKeys = 8: the COMPLETE graph, 525 782 408 states / 6 708 500 293 generated / depth 33, every one of the 33
    per-level counts equal to the exact-dedup oracle's (run on the GPU box's host: tests/golden/raft_levels.json `source`)."""
    c = _golden("raft3_mcr4_t3_m1_k8_complete")
    params = oracle.raft_device_params(c["params"], 8, 2, 4)
    assert params == [3, 4, 3, 3, 1, 1, 8, 2, 4, 8] and amd.state_bytes("raft", params) == 136 and c["max_stat"][:3] == [8, 2, 4]
    eng = amd.Engine("raft", params, table_capacity=17 << 26, arena_capacity=c["distinct"] + (1 << 16), chunk_states=1 << 22, trace=False)
    r = eng.run()
    assert r.levels == c["levels"]
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (525782408, 6708500293, 33, "ok", 0)
    eng.close()


def test_config4_model_one_billion_states_on_one_gpu(amd):
    """BASELINE config 4's model (examples/raft.tla, Server = {s1..s5}, MaxClientRequests = 6 => log <= 5; raft.tla:11-24) at the
    size its 8-GPU run is budgeted for, on ONE GPU: 18 BFS levels = 924 041 864 states resident in HBM (W = 192 B with the compact
    layout of round 3: 177 GB; the 944 B of round 2 would have needed 872 GB).  Level 19 alone has 1.26e9 states: 18 levels is what
    288 GB can hold whatever the arena policy.  Levels 1-15 (63 297 104 states) equal the exact-dedup oracle's — level 15 is where
    round 2's 64-bit guard mask lost 216 states (AppendEntries of a leader s4 / s5); beyond them the fused engine and the sharded
    engine (mc_shard_run over RCCL at world size 1: route-mode expand, packed exchange with itself, keep) must agree level by level."""
    c = _golden("raft5_mcr6_t2_m1_levels18")   # round 4: ALL 18 levels from the exact-dedup oracle (GPU box's host, 274 s) — no count here comes from the GPU
    assert c["levels"][:15] == _golden("raft5_mcr6_t2_m1_levels15")["levels"] and "oracle_mc" in c["source"]
    params = [5, 6, 2, 5, 1, 1, 18, 1, 4]
    assert amd.state_bytes("raft", params) == 192 and c["max_stat"][:3] <= [18, 1, 4]
    eng = amd.Engine("raft", params, table_capacity=3 << 29, arena_capacity=1_300_000_000, chunk_states=1 << 22, max_levels=18, trace=False)
    r = eng.run()
    eng.close()
    assert r.levels == c["levels"] and r.verdict == "budget" and r.depth == 18
    assert (r.distinct, r.generated) == (c["distinct"], c["generated"]) == (924041864, 10345499171)
    from tla_rust_amd.binding import Comm
    comm = Comm(Comm.unique_id(), 0, 1, 0)
    eng = amd.Engine("raft", params, table_capacity=3 << 29, arena_capacity=1_300_000_000, chunk_states=1 << 21, trace=False, shard_rank=0, shard_count=1)
    s, st = comm.shard_run(eng, chunk_states=1 << 21, max_levels=18)
    eng.close()
    comm.close()
    assert (s.levels, s.distinct, s.generated, s.verdict) == (r.levels, r.distinct, r.generated, "budget") and st["stay_levels"] >= 3


def test_parked_overflow_writes_every_state_in_wave(amd, oracle):
    """MC_F_PARK (round 6): the PARK instantiation of the by-family kernel — a wavefront whose survivor list fills up
    parks 64 survivors in the new-list's memory and the workgroup's own tail writes them in later rounds, nothing goes through
    k_materialise.  Config 4's five-server model, 16 levels (158 M states; 11 % of them overflow the lists): the oracle's per-level counts,
    and every state of the large levels written in-wave (without the flag: 89 %); the same on a 3-server complete graph with a trace kept (parent pointers of parked states)."""
    c = _golden("raft5_mcr6_t2_m1_levels18")
    params = [5, 6, 2, 5, 1, 1, 18, 1, 4]
    eng = amd.Engine("raft", params, table_capacity=1 << 30, arena_capacity=170_000_000, chunk_states=1 << 22, max_levels=16, trace=False, debug_flags=32768)
    r = eng.run()
    ks = eng.kernel_stats()
    eng.close()
    assert r.levels == c["levels"][:16] and r.verdict == "budget" and r.distinct == sum(c["levels"][:16])
    # (everything but what Init enumerates; the batched small levels run the PARK instantiation too)
    assert r.distinct - 64 <= ks["inwave_states"] < r.distinct, ks
    k = _golden("raft3_mcr4_t2_m1_k7_complete")
    for flags in (32768, 0):
        eng = amd.Engine("raft", oracle.raft_device_params(k["params"], 7, 1, 4), table_capacity=1 << 24, arena_capacity=2_400_000, chunk_states=1 << 16, trace=True, debug_flags=flags)
        r = eng.run()
        eng.close()
        assert (r.distinct, r.generated, r.depth, r.verdict, r.levels) == (k["distinct"], k["generated"], k["depth"], "ok", k["levels"])


def test_next_complete_graph_on_gpu(amd):
    """MaxMsgKeys = 11: 336 581 097 states / 3 913 649 887 generated / depth 35, the largest complete graph the exact-dedup oracle
    has verified (on the GPU box's host: the build container cannot hold it; tests/golden/raft_levels.json `source`).  116 GB arena."""
    c = _golden("raft3_mcr4_t2_m1_k11_complete")
    assert c["max_stat"][:3] == [11, 1, 4]
    params = [3, 4, 2, 3, 1, 1, 11, 1, 4, 11]
    eng = amd.Engine("raft", params, table_capacity=5 << 27, arena_capacity=c["distinct"] + (1 << 16), chunk_states=1 << 22, trace=False)
    r = eng.run()
    assert r.levels == c["levels"]
    assert (r.distinct, r.generated, r.depth, r.verdict, r.queue_left) == (336581097, 3913649887, 35, "ok", 0)
    eng.close()


def test_bench_workload_with_tuned_capacities(amd):
    """bench.py runs the same model with slot-array capacities sized from the oracle's maxima
    (16 / 2 / 8 instead of the defaults 40 / 4 / 16): W changes, the state graph must not."""
    c = _golden("raft3_mcr4_t2_m1_bench")
    eng = amd.Engine("raft", c["params"] + [16, 2, 8], table_capacity=1 << 27, arena_capacity=30_000_000, chunk_states=1 << 19,
                     max_distinct=c["max_distinct"], trace=False)
    assert amd.state_bytes("raft", c["params"] + [16, 2, 8]) == 176
    r = eng.run()
    assert r.levels == c["levels"] and (r.distinct, r.generated) == (c["distinct"], c["generated"])
    eng.close()


def test_capacity_too_small_overflows(amd):
    eng = amd.Engine("raft", [2, 2, 2, 9, 2, 1, 8, 1, 2], table_capacity=1 << 21, arena_capacity=1 << 19)
    with pytest.raises(amd.McError) as ei:
        eng.run()
    assert ei.value.code == -3
    eng.close()


def test_overflow_is_reported_not_dropped(amd, oracle):
    """overflow of a slot array must raise MC_EOVERFLOW, never silently drop a state.  Three servers with
    8 message slots and no MaxMsgKeys bound: the ninth key appears on level 16, deep inside the run (692 605 states are stored by
    then); a run that stops before that level is unaffected."""
    params = [3, 4, 2, 3, 1, 1, 8, 2, 8]
    eng = amd.Engine("raft", params, table_capacity=1 << 24, arena_capacity=1 << 23, chunk_states=1 << 16, max_distinct=3_000_000, trace=False)
    with pytest.raises(amd.McError) as ei:
        eng.run()
    assert ei.value.code == -3
    eng.close()
    o = oracle.oracle_run("raft", params[:6], max_levels=12)
    eng = amd.Engine("raft", params, table_capacity=1 << 24, arena_capacity=1 << 23, chunk_states=1 << 16, max_levels=12, trace=False)
    r = eng.run()
    assert (r.verdict, r.levels, r.generated) == ("budget", o["levels"], o["generated"]) and o["max_stat"][0] <= 8
    eng.close()


def test_no_two_leaders_negative_control_on_gpu(amd, tmp_path):
    """NoTwoLeaders (raft.tla:500-507) holds in every reachable state, so no graph test ever sees the kernels raise it.  Negative
    control through the checkpoint door: the one-state checkpoint of Init is edited by hand into "s1 Leader of term 2, s2
    Candidate of term 2 holding the votes {s2, s3}" and recovered; BecomeLeader(s2) must be reported as a violation of
    invariant 0 with a 2-state counterexample; with s2 in term 3 the same run finds nothing."""
    import struct
    params = [3, 4, 3, 3, 1, 1]
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    e = amd.Engine("raft", params, max_levels=1, **kw)
    assert e.run().distinct == 1
    e.checkpoint(tmp_path / "init")
    e.close()
    data = (tmp_path / "init").read_bytes()
    hdr = 8 + 4 + 4 + 16 * 8 + 4 + 4 + 6 * 8      # CkHeader (engine.hip), then the level table (1 entry), then the arena's blocks
    arena = hdr + 8
    word = lambda buf, w: struct.unpack_from("<Q", buf, arena + w * 64 * 8)[0]   # state 0 of block 0: word w at (w * 64 + 0)
    for same_term in (True, False):
        buf = bytearray(data)
        sv0, sv1 = word(buf, 2), word(buf, 4)       # spec_raft.h: W_SRV(i) = 2 + 2 i; term[0,3) state[3,5) votesGranted[8,13)
        struct.pack_into("<Q", buf, arena + 2 * 64 * 8, (sv0 & ~0x1f) | 2 | (2 << 3))
        struct.pack_into("<Q", buf, arena + 4 * 64 * 8, (sv1 & ~(0x1f | (0x1f << 8))) | (2 if same_term else 3) | (1 << 3) | (0b110 << 8))
        (tmp_path / "edited").write_bytes(buf)
        e = amd.Engine("raft", params, max_levels=2, **kw)
        e.restore(tmp_path / "edited")
        r = e.run()
        if same_term:
            assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", 0, 2)
            tr = e.trace()
            assert tr[-1][0] == "BecomeLeader" and tr[-1][1].count("Leader") >= 2
        else:
            assert r.verdict == "budget"
        e.close()


def test_table_full_is_an_error(amd):
    eng = amd.Engine("atomic_add", [16], table_capacity=1 << 12, arena_capacity=1 << 17)
    with pytest.raises(amd.McError) as ei:
        eng.run()
    assert ei.value.code == -4
    eng.close()


@pytest.mark.parametrize("find", [1, 2, 3, 4, 5, 6, 7])
def test_ssi_expected_violations_on_gpu(amd, oracle, find):
    """serializableSnapshotIsolation.tla:81-96: each 'interesting history' is reachable; same shortest trace length as the oracle."""
    o = oracle.oracle_run("ssi", [3, 2, 127, find])
    eng = amd.Engine("ssi", [3, 2, 127, find], table_capacity=1 << 25, arena_capacity=1 << 24, chunk_states=1 << 18)
    r = eng.run()
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", 7, len(o["trace"]))
    tr = eng.trace()
    assert len(tr) == r.trace_len and tr[0][0] == "Initial predicate" and tr[0][1] == o["trace"][0][1]
    eng.close()


@pytest.mark.parametrize("params", [[2, 2, 127, 2], [2, 2, 127, 3], [3, 2, 127, 6]])
def test_ssi_invariants_checked_on_the_level_a_budget_stops_at_on_gpu(amd, oracle, params):
    """invariants of the SI models are evaluated at expansion; a run cut by max_levels exactly at the depth
    of a violation must report it (k_check_frontier over the unexpanded last level), as TLC's check-on-generation does."""
    L = len(oracle.oracle_run("ssi", params)["trace"])
    o = oracle.oracle_run("ssi", params, max_levels=L)
    eng = amd.Engine("ssi", params, table_capacity=1 << 24, arena_capacity=1 << 23, chunk_states=1 << 16, max_levels=L)
    r = eng.run()
    assert o["verdict"] == "invariant"
    assert (r.verdict, r.violated_invariant, r.trace_len, r.levels) == ("invariant", o["violated_invariant"], L, o["levels"])
    assert len(eng.trace()) == L
    eng.close()
    eng = amd.Engine("ssi", params, table_capacity=1 << 24, arena_capacity=1 << 23, chunk_states=1 << 16, max_levels=L - 1)
    assert eng.run().verdict == "budget"
    eng.close()


def test_textbook_si_write_skew_found_when_cut_at_its_depth_on_gpu(amd):
    """textbook SI write skew is a 13-state history: max_levels = 13 stops before level 13 is expanded and must still find it"""
    eng = amd.Engine("ssi", [3, 2, 32, 0, 1], table_capacity=1 << 26, arena_capacity=40_000_000, chunk_states=1 << 19, max_levels=13)
    r = eng.run()
    assert (r.verdict, r.violated_invariant, r.trace_len, r.depth) == ("invariant", 5, 13, 13)
    eng.close()


def test_ssi_2x3_complete_graph_on_gpu(amd):
    """SSI 2 txns x 3 keys = 7 910 565 distinct / 13 246 749 generated / 17 levels (oracle-verified), all invariants on."""
    eng = amd.Engine("ssi", [2, 3, 127, 0], table_capacity=1 << 25, arena_capacity=9_000_000, chunk_states=1 << 19, trace=False)
    r = eng.run()
    assert (r.verdict, r.distinct, r.generated, r.depth) == ("ok", 7910565, 13246749, 17)
    assert r.levels == [1, 2, 12, 60, 354, 1968, 9318, 35286, 102408, 222552, 381444, 641376, 1118376, 1616976, 1824552, 1405080, 550800]
    eng.close()


def test_ssi_4x3_prefix_on_gpu(amd):
    """BASELINE config 5 (4 txns x 3 keys), levels 1-9."""
    eng = amd.Engine("ssi", [4, 3, 127, 0], table_capacity=1 << 27, arena_capacity=24_000_000, chunk_states=1 << 19, max_levels=9, trace=False)
    r = eng.run()
    assert r.levels == [1, 4, 32, 264, 2532, 24576, 236844, 2189052, 18810792] and r.verdict == "budget"
    eng.close()


def test_ssi_4x3_ten_levels_on_gpu(amd):
    """BASELINE config 5 to the depth the bench (`--workload ssi4x3`) and profiles/ quote: 10 levels = 168 052 153 states, every
    per-level count and `generated` from the exact-dedup oracle on the GPU box's host (tests/golden/ssi_levels.json `source`)."""
    c = next(x for x in json.loads((GOLDEN / "ssi_levels.json").read_text())["cases"] if x["name"] == "ssi_4x3_levels10")
    assert "oracle_mc" in c["source"]
    eng = amd.Engine("ssi", c["params"], table_capacity=9 << 26, arena_capacity=c["distinct"] + (1 << 20), chunk_states=1 << 21, max_levels=10, trace=False)
    r = eng.run()
    eng.close()
    assert r.levels == c["levels"] and (r.distinct, r.generated, r.verdict) == (c["distinct"], c["generated"], "budget")


def _ssi_case(name):
    c = next(x for x in json.loads((GOLDEN / "ssi_levels.json").read_text())["cases"] if x["name"] == name)
    assert "oracle_mc" in c["source"]
    return c


def test_ssi_4x3_eleven_levels_on_gpu(amd):
    """config 5 one level deeper than the bench runs it: 11 levels = 1 184 049 193 states (95 GB of states resident), per-level counts and
    `generated` from the oracle on the GPU box's host (75 s there, 0.3 s here)"""
    c = _ssi_case("ssi_4x3_levels11")
    eng = amd.Engine("ssi", c["params"], table_capacity=57 << 26, arena_capacity=c["distinct"] + (1 << 20), chunk_states=1 << 23, max_levels=11, trace=False)
    r = eng.run()
    eng.close()
    assert r.levels == c["levels"] and (r.distinct, r.generated, r.verdict) == (c["distinct"], c["generated"], "budget")


def test_ssi_4x3_symmetry_thirteen_levels_on_gpu(amd):
    """config 5 as the spec's run-book sets it up (SYMMETRY over TxnId and Key, :38-44), 13 levels = 267 790 850 orbits: the number
    profiles/ has quoted since round 1 is now the oracle's (brute-force least image over 144 permutations per successor, 154 s on the GPU
    box's host), level by level"""
    c = _ssi_case("ssi_4x3_sym_levels13")
    eng = amd.Engine("ssi", c["params"], table_capacity=13 << 26, arena_capacity=c["distinct"] + (1 << 20), chunk_states=1 << 22, max_levels=13, trace=False)
    r = eng.run()
    eng.close()
    assert r.levels == c["levels"] and (r.distinct, r.generated, r.verdict) == (c["distinct"], c["generated"], "budget")


@pytest.mark.parametrize("case", ["2x3", "3x2", "4x3_prefix"])
def test_ssi_symmetry_golden_on_gpu(amd, case):
    """cfg SYMMETRY on the SSI model, Key and TxnId symmetry sets as the spec's run-book prescribes (:38-44): per-level orbit
    counts of tests/golden/ssi_symmetry.json (oracle, brute force over |TxnId|! x |Key|! permutations per successor;
    tests/golden/make_ssi_symmetry_golden.py).  3 x 2: 80 807 116 states -> 6 734 049 orbits."""
    g = json.loads((GOLDEN / "ssi_symmetry.json").read_text())[case]
    eng = amd.Engine("ssi", g["params"], table_capacity=1 << 25, arena_capacity=8_000_000, chunk_states=1 << 19,
                     max_distinct=g["max_distinct"], trace=False)
    r = eng.run()
    assert (r.verdict, r.distinct, r.generated, r.depth) == (g["verdict"], g["distinct"], g["generated"], g["depth"])
    assert r.levels == g["levels"]
    eng.close()


def test_textbook_si_write_skew_under_symmetry_on_gpu(amd):
    """Symmetry reduction keeps the verdict and the length of the shortest counterexample (13-state write skew)."""
    g = json.loads((GOLDEN / "ssi_symmetry.json").read_text())["textbook_3x2_cahill"]
    eng = amd.Engine("ssi", g["params"], table_capacity=1 << 24, arena_capacity=4_000_000, chunk_states=1 << 19)
    r = eng.run()
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", 5, g["trace_len"])
    tr = eng.trace()
    assert len(tr) == 13 and tr[-1][1].count('"commit"') >= 2
    eng.close()


@pytest.mark.parametrize("mask,inv", [(32, 5), (64, 6)])
def test_textbook_si_write_skew_on_gpu(amd, mask, inv):
    """textbookSnapshotIsolation.tla, 3 txns x 2 keys: serializability is violated by a 13-state history (oracle:
    tests/test_oracle_golden.py), found by both Cahill's and Bernstein's formulation."""
    eng = amd.Engine("ssi", [3, 2, mask, 0, 1], table_capacity=1 << 26, arena_capacity=40_000_000, chunk_states=1 << 19)
    r = eng.run()
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", inv, 13)
    tr = eng.trace()
    assert len(tr) == 13 and tr[-1][1].count('"commit"') >= 2
    eng.close()


@pytest.mark.parametrize("slots", [3 << 12, 40000, 1 << 16, 100032])
def test_seen_set_of_any_size(amd, oracle, slots):
    """the seen-set is sized in slots, not in powers of two (home bucket = multiply-shift of the fingerprint's low 32 bits):
    same counts for every table size that holds the states, MC_ETABLEFULL for one that does not"""
    o = oracle.oracle_run("raft", [2, 2, 2, 9, 1, 1])          # 13 634 distinct
    eng = amd.Engine("raft", [2, 2, 2, 9, 1, 1], table_capacity=slots, arena_capacity=1 << 16, chunk_states=1 << 10)
    if slots < o["distinct"]:
        with pytest.raises(amd.McError) as e:
            eng.run()
        assert e.value.code == -4
    else:
        r = eng.run()
        assert (r.distinct, r.generated, r.depth, r.levels) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    eng.close()


# ---------------------------------------------------------------------------------------------- random configurations (round 4)
@pytest.mark.parametrize("seed", range(20))
def test_raft_random_configuration_prefix_on_gpu(amd, oracle, seed):
    """the seeded random raft configurations of tests/test_lowering_sweep.py (server counts, term / log / message bounds, MaxMsgKeys,
    invariant masks) through the HIP engine against the C oracle run beside it: counters, verdict, depth, every per-level count"""
    from test_lowering_sweep import raft_config
    dev = raft_config(seed)
    o = oracle.oracle_run("raft", oracle.raft_oracle_params(dev), max_distinct=60000)
    eng = amd.Engine("raft", dev, table_capacity=1 << 21, arena_capacity=1 << 20, chunk_states=1 << 13, max_distinct=60000, trace=False)
    r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels"):
        assert o[k] == r[k], (k, dev)
    eng.close()


@pytest.mark.parametrize("seed", range(12))
def test_ssi_random_configuration_prefix_on_gpu(amd, oracle, seed):
    from test_lowering_sweep import ssi_config
    params = ssi_config(seed)
    o = oracle.oracle_run("ssi", params, max_distinct=40000)
    eng = amd.Engine("ssi", params, table_capacity=1 << 21, arena_capacity=1 << 20, chunk_states=1 << 13, max_distinct=40000, trace=False)
    r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels"):
        assert o[k] == r[k], (k, params)
    eng.close()


def _random_engine_case(seed):
    import random
    from test_lowering_sweep import raft_config, ssi_config
    r = random.Random(6000 + seed)
    kind = r.choice(["raft", "raft", "raft", "ssi", "atomic_add", "pcal_intro"])
    if kind == "raft":
        params = raft_config(r.randrange(40))
    elif kind == "ssi":
        params = ssi_config(r.randrange(24))
    elif kind == "atomic_add":
        params = [r.randrange(3, 13)]
    else:
        params = r.choice([[0, 1, 20, 2], [1, 1, 20, 2], [0, 1, 7, 3]])
    flags = 0
    for bit, p in ((256, 0.3), (65536, 0.3), (131072, 0.25), (8192, 0.25), (32, 0.15)):   # NOBATCH, NOINWAVE, WAVETAIL, NOFILTER, NOFAMILY
        if r.random() < p:
            flags |= bit
    cfg = dict(chunk_states=r.choice([64, 200, 1024, 5000, 1 << 15]), table_capacity=r.choice([1 << 20, 1 << 21, 1 << 23]), arena_capacity=r.choice([1 << 19, 1 << 21]),
               trace=r.random() < 0.5, timing=r.random() < 0.3, debug_flags=flags)
    return kind, params, cfg, r.choice([0, 0, 1, 3])


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("TLAMC_SWEEP", "48"))))
def test_random_model_and_engine_settings_on_gpu(amd, oracle, seed):
    """seeded random combinations of a model (raft / SI configurations of tests/test_lowering_sweep.py, the PlusCal root specs) with the
    engine's settings — states per launch from 64 to 2^15, seen-set sizes on both sides of the bucket / single-slot switch, parent
    tracking on and off, kernel timing, and the A/B flags (one round trip per level, every state through the new-list, the
    wavefront tail, no duplicate filter, slot-by-slot expansion) — and with the search taken in steps of 1 or 3 levels
    (mc_engine_step) instead of one run: counters, verdict, depth and every per-level count are the oracle's in all of them"""
    kind, params, cfg, step = _random_engine_case(seed)
    oparams = oracle.raft_oracle_params(params) if kind == "raft" else params
    o = oracle.oracle_run(kind, oparams, max_distinct=40000)
    eng = amd.Engine(kind, params, max_distinct=40000, **cfg)
    if step:
        r = eng.step(step)
        for _ in range(4096):
            if not (r.verdict == "budget" and r.distinct < 40000 and r.queue_left):
                break
            r = eng.step(step)
    else:
        r = eng.run()
    for k in ("distinct", "generated", "depth", "verdict", "levels"):
        assert o[k] == r[k], (k, kind, params, cfg, step)
    eng.close()
