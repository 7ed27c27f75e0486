"""Checkpoint / recover on the GPU (TLC: "-- Checkpointing of run states/... completed.", testout1:10, and -recover):
a search stopped on a budget, written to a file, reloaded into a NEW engine and continued must report exactly what the
uninterrupted search reports — counters, depth, per-level counts, verdict, counterexample."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
KEYS = ("distinct", "generated", "depth", "verdict", "levels", "queue_left")
KW = dict(table_capacity=1 << 22, arena_capacity=1 << 20, chunk_states=1 << 12)


@pytest.fixture(scope="module")
def amd():
    import tla_rust_amd
    assert tla_rust_amd.device_count() >= 1, "no HIP device visible"
    return tla_rust_amd


@pytest.mark.parametrize("spec,params,cut", [
    ("atomic_add", [11], 5), ("pcal_intro", [0, 1, 20, 2], 3), ("raft", [2, 2, 2, 9, 1, 1], 9), ("raft", [2, 2, 2, 9, 1, 1], 1),
    ("ssi", [2, 2, 127, 0], 7), ("ssi", [3, 1, 127, 0, 0, 3], 6), ("ssi", [2, 2, 127, 0, 1], 12)])
@pytest.mark.parametrize("trace", [True, False])
def test_recovered_run_equals_uninterrupted_run(amd, oracle, tmp_path, spec, params, cut, trace):
    o = oracle.oracle_run(spec, params)
    a = amd.Engine(spec, params, max_levels=cut, trace=trace, **KW)
    ra = a.run()
    assert ra.verdict == "budget" and ra.depth == cut and ra.levels == o["levels"][:cut]
    ck = tmp_path / "run.ckpt"
    a.checkpoint(ck)
    states_a = a.read_states(0, ra.distinct)
    a.close()
    b = amd.Engine(spec, params, trace=trace, **KW)
    b.restore(ck)
    rb = b.run()
    for k in KEYS:
        assert rb[k] == o[k], k
    # the checkpointed levels are the states the first engine found, in its order
    assert b.read_states(0, ra.distinct) == states_a
    # a second run() of the same engine starts from Init again
    rc = b.run()
    for k in KEYS:
        assert rc[k] == o[k], k
    b.close()


def test_checkpoint_of_a_checkpoint(amd, oracle, tmp_path):
    params = [2, 2, 127, 0]
    o = oracle.oracle_run("ssi", params)
    ck = tmp_path / "c"
    e = amd.Engine("ssi", params, max_levels=4, **KW)
    e.run(); e.checkpoint(ck); e.close()
    e = amd.Engine("ssi", params, max_levels=9, **KW)        # budgets are absolute: 5 more levels
    e.restore(ck)
    r = e.run()
    assert r.verdict == "budget" and r.levels == o["levels"][:9]
    e.checkpoint(ck); e.close()
    e = amd.Engine("ssi", params, **KW)
    e.restore(ck)
    r = e.run()
    for k in KEYS:
        assert r[k] == o[k], k
    e.close()


def test_counterexample_found_after_recovery(amd, oracle, tmp_path):
    """the parent pointers travel with the checkpoint: the shortest counterexample is rebuilt through the checkpointed levels"""
    params = [3, 2, 127, 3]     # ~AtLeastNTxnsAbortedDueToReason(1, deadlock prevention): serializableSnapshotIsolation.tla:81-96
    o = oracle.oracle_run("ssi", params)
    kw = dict(table_capacity=1 << 24, arena_capacity=1 << 22, chunk_states=1 << 16)
    a = amd.Engine("ssi", params, max_levels=len(o["trace"]) - 3, **kw)
    assert a.run().verdict == "budget"
    a.checkpoint(tmp_path / "c"); a.close()
    b = amd.Engine("ssi", params, **kw)
    b.restore(tmp_path / "c")
    r = b.run()
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", 7, len(o["trace"]))
    tr = b.trace()
    assert len(tr) == r.trace_len and tr[0][0] == "Initial predicate" and tr[0][1] == o["trace"][0][1]
    assert "forced by deadlock-prevention" in tr[-1][1]
    b.close()


def test_checkpoint_misuse_is_refused(amd, tmp_path):
    e = amd.Engine("ssi", [2, 2, 127, 0], max_levels=5, **KW)
    with pytest.raises(amd.McError):
        e.checkpoint(tmp_path / "early")          # nothing has run yet
    e.run()
    e.checkpoint(tmp_path / "ok")
    e.close()
    for spec, params in (("ssi", [2, 2, 127, 0, 0, 3]), ("ssi", [3, 2, 127, 0]), ("atomic_add", [5])):
        other = amd.Engine(spec, params, **KW)
        with pytest.raises(amd.McError):
            other.restore(tmp_path / "ok")          # another model: other constants / SYMMETRY / spec
        other.close()
    small = amd.Engine("ssi", [2, 2, 127, 0], table_capacity=1 << 22, arena_capacity=64)
    with pytest.raises(amd.McError):
        small.restore(tmp_path / "ok")
    small.close()
    data = (tmp_path / "ok").read_bytes()
    (tmp_path / "cut").write_bytes(data[: len(data) // 2])
    (tmp_path / "junk").write_bytes(b"not a checkpoint" * 100)
    e = amd.Engine("ssi", [2, 2, 127, 0], **KW)
    for bad in ("cut", "junk", "missing"):
        with pytest.raises(amd.McError):
            e.restore(tmp_path / bad)
    viol = amd.Engine("pcal_intro", [1, 0, 20, 2], **KW)    # README variant: the assertion fails
    assert viol.run().verdict == "assert"
    with pytest.raises(amd.McError):
        viol.checkpoint(tmp_path / "v")              # a run that ended in an error is not continued
    viol.close()
    e.close()


def test_cli_checkpoint_and_recover(amd, tmp_path):
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    tla, cfg, ck = str(ROOT / "specs" / "MCssi.tla"), str(ROOT / "specs" / "MCssi_2x2.cfg"), str(tmp_path / "ssi.ckpt")
    p = subprocess.run([str(mc), tla, "-config", cfg, "-maxlevels", "6", "-checkpoint", ck], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert f"-- Checkpointing of run {ck} completed." in p.stdout                  # testout1:10
    p = subprocess.run([str(mc), tla, "-config", cfg, "-recover", ck], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "50121 states generated, 29629 distinct states found, 0 states left on queue." in p.stdout
    assert "The depth of the complete state graph search is 13." in p.stdout
    p = subprocess.run([str(mc), tla, "-config", str(ROOT / "specs" / "MCssi_2x2_sym.cfg"), "-recover", ck], capture_output=True, text=True)
    assert p.returncode == 1 and "other constants" in p.stderr


def test_large_checkpoint_round_trip(amd, tmp_path):
    """SSI 2 x 3 (7 910 565 states): stop after level 13 (2.5 M states, 200 MB), recover, finish."""
    kw = dict(table_capacity=1 << 25, arena_capacity=9_000_000, chunk_states=1 << 19, trace=False)
    a = amd.Engine("ssi", [2, 3, 127, 0], max_levels=13, **kw)
    ra = a.run()
    assert ra.verdict == "budget" and ra.distinct == sum([1, 2, 12, 60, 354, 1968, 9318, 35286, 102408, 222552, 381444, 641376, 1118376])
    a.checkpoint(tmp_path / "big"); a.close()
    assert (tmp_path / "big").stat().st_size >= ra.distinct * 80
    b = amd.Engine("ssi", [2, 3, 127, 0], **kw)
    b.restore(tmp_path / "big")
    r = b.run()
    assert (r.verdict, r.distinct, r.generated, r.depth) == ("ok", 7910565, 13246749, 17)
    b.close()


def test_checkpoint_of_a_compiled_program_is_bound_to_that_program(amd, tmp_path):
    """a checkpoint of a compiled PlusCal program must be refused by ANY other program (other algorithm, other
    constants, other invariants), not only by one with another state width: the header carries a hash of the program image."""
    S = ROOT / "specs" / "pluscal"
    text = (S / "cas_counter.tla").read_text()
    cfg = (S / "cas_counter.cfg").read_text()
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    prog = amd.Program(text, cfg)
    full = amd.Engine("pcal", prog.params, **kw)
    want = full.run()
    full.close()
    a = amd.Engine("pcal", prog.params, max_levels=6, **kw)
    assert a.run().verdict == "budget"
    ck = tmp_path / "cas.ckpt"
    a.checkpoint(ck)
    a.close()
    b = amd.Engine("pcal", prog.params, **kw)      # the same program: accepted, and the continued run equals the uninterrupted one
    b.restore(ck)
    rb = b.run()
    assert (rb.distinct, rb.generated, rb.depth, rb.verdict, rb.levels) == (want.distinct, want.generated, want.depth, want.verdict, want.levels)
    b.close()
    # the same algorithm with another constant (same variables, same state width) and with one invariant less: both refused
    import re
    other_n = amd.Program(text, re.sub(r"N\s*=\s*\d+", "N = 3", cfg))
    assert " SeenIsOld" in cfg
    fewer_inv = amd.Program(text, cfg.replace(" SeenIsOld", ""))
    for other in (other_n, fewer_inv):
        assert amd.state_bytes("pcal", other.params) == amd.state_bytes("pcal", prog.params)
        e = amd.Engine("pcal", other.params, **kw)
        with pytest.raises(amd.McError) as ei:
            e.restore(ck)
        assert "another compiled program" in str(ei.value)
        e.close()
        other.close()
    prog.close()


def test_corrupt_level_table_is_refused(amd, tmp_path):
    """restore() validates the level table (starts at 0, strictly increasing, consistent with the frontier)"""
    import struct
    e = amd.Engine("ssi", [2, 2, 127, 0], max_levels=6, **KW)
    r = e.run()
    e.checkpoint(tmp_path / "ok")
    e.close()
    data = bytearray((tmp_path / "ok").read_bytes())
    # header: magic[8] spec_id nparams params[16] words has_trace distinct generated cells lo hi nlevels, then the level table
    hdr = 8 + 4 + 4 + 16 * 8 + 4 + 4 + 6 * 8
    nlevels = struct.unpack_from("<Q", data, hdr - 8)[0]
    assert nlevels == r.depth
    for mutate in (lambda t: [t[0] + 1] + t[1:], lambda t: t[:2] + [t[1]] + t[3:], lambda t: t[:-1] + [t[-1] + 1]):
        bad = bytearray(data)
        table = list(struct.unpack_from(f"<{nlevels}Q", bad, hdr))
        struct.pack_into(f"<{nlevels}Q", bad, hdr, *mutate(table))
        (tmp_path / "bad").write_bytes(bad)
        e = amd.Engine("ssi", [2, 2, 127, 0], **KW)
        with pytest.raises(amd.McError) as ei:
            e.restore(tmp_path / "bad")
        assert ei.value.code == -8
        e.close()
    e = amd.Engine("ssi", [2, 2, 127, 0], max_levels=10, **KW)
    assert e.run().distinct * 2 > 1 << 10
    e.checkpoint(tmp_path / "ten")
    e.close()
    small_table = amd.Engine("ssi", [2, 2, 127, 0], table_capacity=1 << 10, arena_capacity=1 << 20, chunk_states=1 << 12)
    with pytest.raises(amd.McError) as ei:      # the states of ten levels do not fit a 1024-slot seen-set at load 1/2
        small_table.restore(tmp_path / "ten")
    assert ei.value.code == -4
    small_table.close()


@pytest.mark.parametrize("spec,params,stride", [("ssi", [2, 2, 127, 0], 3), ("raft", [2, 2, 2, 9, 1, 1], 5), ("paxos", [0, 3, 2, 2, 15, 3, 1], 4),
                                                  ("atomic_add", [11], 1)])
def test_incremental_steps_equal_one_run(amd, oracle, spec, params, stride):
    """mc_engine_step (the optional incremental entry point): the search advanced `stride` levels at a time, in
    place — arena, seen-set and parent pointers stay in HBM between the calls — ends with the counters, the per-level counts
    and the state sets of one uninterrupted run; every intermediate result is a prefix of it"""
    o = oracle.oracle_run(spec, params, check_deadlock=(spec != "paxos"))
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 10, deadlock=(spec != "paxos"))
    e = amd.Engine(spec, params, **kw)
    seen_levels, r = 0, None
    for _ in range(100):
        r = e.step(stride)
        assert r.levels == o["levels"][:len(r.levels)] and len(r.levels) >= min(len(o["levels"]), seen_levels + 1)
        seen_levels = len(r.levels)
        if r.verdict != "budget":
            break
    assert (r.verdict, r.distinct, r.generated, r.depth, r.levels, r.queue_left) == \
           (o["verdict"], o["distinct"], o["generated"], o["depth"], o["levels"], o["queue_left"])
    full = amd.Engine(spec, params, **kw)
    rf = full.run()
    assert sorted(e.state_texts(0, r.distinct)) == sorted(full.state_texts(0, rf.distinct))
    r2 = e.step(2)                     # after the end: starts over
    assert r2.levels == o["levels"][:2] and r2.verdict == "budget"
    e.close(); full.close()


def test_counterexample_found_by_a_later_step(amd, oracle):
    params = [3, 2, 127, 3]
    o = oracle.oracle_run("ssi", params)
    e = amd.Engine("ssi", params, table_capacity=1 << 24, arena_capacity=1 << 22, chunk_states=1 << 16)
    r = e.step(len(o["trace"]) - 3)
    assert r.verdict == "budget"
    r = e.step(2)
    assert r.verdict == "budget"
    r = e.step(5)
    assert (r.verdict, r.violated_invariant, r.trace_len) == ("invariant", 7, len(o["trace"]))
    tr = e.trace()
    assert len(tr) == r.trace_len and tr[0][0] == "Initial predicate" and "forced by deadlock-prevention" in tr[-1][1]
    e.close()


def test_a_progress_callback_can_stop_the_run_and_steps_finish_it(amd, oracle):
    """mc_engine_request_stop (round 6: how `mc` leaves the device interpreter once a program's generated code is built): called from a
    progress callback, the run ends before its next BFS level like a run whose budget is spent — a prefix of the oracle's levels, verdict
    "budget", the frontier left on the queue — and mc_engine_step continues it in place to the counters of one uninterrupted run"""
    params = [2, 2, 127, 0]
    o = oracle.oracle_run("ssi", params)
    e = amd.Engine("ssi", params, table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 10)
    calls = []
    def fn(lv, g, d, q):
        calls.append(lv)
        if len(calls) == 1:
            e.request_stop()
    e.set_progress(fn, 0.0)
    r = e.run()
    assert calls and r.verdict == "budget" and 0 < len(r.levels) < len(o["levels"]) and r.levels == o["levels"][:len(r.levels)]
    assert r.queue_left == r.levels[-1]
    e.set_progress(None)
    r = e.step(100)
    assert (r.verdict, r.distinct, r.generated, r.depth, r.levels, r.queue_left) == \
           (o["verdict"], o["distinct"], o["generated"], o["depth"], o["levels"], o["queue_left"])
    r = e.run()                           # a request does not outlive its run
    assert (r.verdict, r.distinct) == (o["verdict"], o["distinct"])
    e.close()
