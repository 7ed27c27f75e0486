"""The sharded path on the GPU — always the library's ONE level loop (tla_rust_amd/csrc/shard_loop.h): (a) one process, one
shard (routing + compaction + probe + send-materialise + ingest with nranks = 1) against the fused single-GPU run; (b) 2 / 3 /
4 / 8 processes sharing GPU 0 with torch.distributed's gloo collectives as the loop's transport, against the oracle; (c) the
hip-rccl back-end itself (mc_comm_* / mc_shard_run / mc_shard_trace: `mc X.tla -gpus P`, `bench.py --gpus N`) with P = 2 / 4 /
8 ranks on this one GPU through the librccl stand-in of tests/_fakerccl ($TLAMC_RCCL) — RCCL refuses two ranks on one device."""
import pytest

from test_sharded_gloo import run_dist

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("spec,params,opts", [("atomic_add", [12], {}), ("pcal_intro", [0, 1, 20, 2], {}),
                                               ("raft", [2, 2, 2, 9, 2, 1], {}), ("raft", [3, 4, 2, 3, 1, 1, 16, 2, 8], {"max_distinct": 1000000})])
def test_single_shard_step_api_equals_fused_run(spec, params, opts):
    import tla_rust_amd as amd
    from tla_rust_amd.sharded import ShardedChecker
    eng = amd.Engine(spec, params, table_capacity=1 << 23, arena_capacity=1 << 21, chunk_states=1 << 14, max_distinct=opts.get("max_distinct", 0))
    a = eng.run()
    eng.close()
    chk = ShardedChecker(spec, params, device=0, chunk_states=1 << 14, max_distinct=opts.get("max_distinct", 0), table_capacity=1 << 23,
                         arena_capacity=1 << 21, move_fanout=40)
    b = chk.run()
    chk.close()
    assert (a.distinct, a.generated, a.depth, a.levels, a.verdict) == (b.distinct, b.generated, b.depth, b.levels, b.verdict)


@pytest.mark.parametrize("spec,params,opts", [("raft", [2, 2, 2, 9, 2, 1], {"chunk": 1 << 12}),
                                               ("raft", [3, 2, 2, 9, 1, 1], {"max_distinct": 300000, "chunk": 1 << 14, "table": 1 << 22, "arena": 1 << 20}),
                                               ("pcal_intro", [1, 0, 20, 2], {"chunk": 512})])
def test_two_ranks_on_one_gpu_equal_oracle(oracle, tmp_path, spec, params, opts):
    o = oracle.oracle_run(spec, params, max_distinct=opts.get("max_distinct", 0))
    r = run_dist("hip", 2, spec, params, tmp_path, opts)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == \
           (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
    assert sum(r["shares"]) == o["distinct"]


def test_two_ranks_stay_mode_on_gpu(oracle, tmp_path):
    """stay levels of two engines on one GPU, several rounds per level (chunk 2^14): the default form (host-paced rounds with exact sizes,
    mc_shard_expand_finish / _probe / _keep_slot, the next round's expand in flight meanwhile); the fixed-capacity forms
    (mc_shard_expand_pack / _probe_pack / _keep_pack, counts in band, three streams and their events) are test_the_three_forms_... below"""
    params = [3, 2, 2, 9, 1, 1]
    o = oracle.oracle_run("raft", params, max_distinct=300000)
    r = run_dist("hip", 2, "raft", params, tmp_path, {"max_distinct": 300000, "chunk": 1 << 14, "table": 1 << 22, "arena": 1 << 20,
                                                       "stay_threshold": 200, "rebalance_ratio": 1.5})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert r["phases"].get("stay_levels", 0) >= 3


@pytest.mark.parametrize("form,extra", [("fixed", {"exchange": "packed"}), ("measured", {"exchange": "measured", "cap_safety_pct": 140}), ("exact", {"exchange": "exact"})])
def test_the_three_forms_of_a_stay_level_on_gpu(oracle, tmp_path, form, extra):
    """the stay levels of two HIP engines in their three forms (include/tlamc.h: MC_SHARD_PACKED | MC_SHARD_FIXED_CAPS, MC_SHARD_PACKED + cap_safety_pct, the default):
    buckets from packed_fanout, buckets from the fill of the previous level (the in-band counts read back by the loop), host-paced rounds
    with exact sizes (mc_shard_expand_finish + mc_shard_probe + mc_shard_keep_slot) — the oracle's counters and per-level counts each
    time; 9 bytes per routed candidate in the exact form"""
    params = [3, 2, 2, 9, 1, 1]
    o = oracle.oracle_run("raft", params, max_distinct=300000)
    r = run_dist("hip", 2, "raft", params, tmp_path, dict({"max_distinct": 300000, "chunk": 1 << 14, "table": 1 << 22, "arena": 1 << 20,
                                                           "stay_threshold": 200, "rebalance_ratio": 1.5}, **extra))
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    st = r["stats"]
    assert st["stay_levels"] >= 3 and st["restarts"] == 0 and st["routed_candidates"] > 100000, st
    if form == "exact":
        assert st["fp_answer_bytes"] == 9 * st["routed_candidates"] and st["measured_levels"] == 0, st
    elif form == "measured":
        assert st["measured_levels"] >= 2 and st["fp_answer_bytes"] < 2.5 * 9 * st["routed_candidates"], st
    else:
        assert st["measured_levels"] == 0 and st["fp_answer_bytes"] > 9 * st["routed_candidates"], st


@pytest.mark.parametrize("exchange", ["packed", "measured", "exact"])
def test_native_rccl_loop_exchange_forms(exchange):
    """`mc X.tla -gpus 4 -samedevice -exchange packed | measured | exact` (exact is the default the other tests run): the native loop over
    the nccl* entry points with each form of the stay levels — the one-GPU run's counter line, and the report's exchange line"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    args = [S / "MCraft.tla", "-config", S / "MCraft.cfg", "-maxdistinct", 3000000, "-tablelog2", 24, "-arena", 6000000, "-chunk", 65536]
    q = _mc(*args, "-noprogress")
    line = next((ln for ln in q.stdout.splitlines() if "distinct states found" in ln), None)
    p = _mc(*args, "-gpus", 4, "-samedevice", "-exchange", exchange, env=_fake_env())
    assert p.returncode == q.returncode, (p.stdout[-1500:], p.stderr[-1500:])
    assert line and line in p.stdout, (line, p.stdout[-800:], p.stderr[-800:])
    x = next(ln for ln in p.stdout.splitlines() if ln.startswith("(exchange"))
    ratio = float(x.split("candidates = ")[1].split(" x")[0])
    assert ratio == 1.0 if exchange == "exact" else ratio > 1.0, x


def test_a_full_exchange_bucket_restarts_the_search_on_gpu(oracle, tmp_path):
    """round 4: an allowance of ONE in-model successor per state (packed_fanout = move_fanout = 1) overflows the route buckets of the
    HIP engines (DEV_EROUTE / "send buffer too small" -> MC_EROUTE on every rank, nothing truncated); mc_shard_run_transport starts
    the search over with twice the allowance until it fits: the oracle's counters, and mc_shard_stats.restarts says how often"""
    params = [3, 2, 2, 9, 1, 1]
    o = oracle.oracle_run("raft", params, max_distinct=100000)
    r = run_dist("hip", 2, "raft", params, tmp_path, {"max_distinct": 100000, "chunk": 1 << 13, "table": 1 << 22, "arena": 1 << 20,
                                                       "stay_threshold": 200, "rebalance_ratio": 1.5, "packed_fanout": 1, "move_fanout": 1})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert r["stats"]["restarts"] >= 1


@pytest.mark.parametrize("exchange", ["packed", "exact"])
def test_a_full_route_sub_bucket_restarts_the_search_at_a_production_chunk_size(oracle, tmp_path, exchange):
    """with rounds of 2^17 states the +4096 slack of a route sub-bucket no longer hides an allowance that is too
    small — the expand kernel itself finds its sub-bucket full (DEV_EROUTE, not DEV_EARENA: `check_dev_error` runs before the host's own
    count test) and every rank restarts with twice the allowance instead of reporting a full arena.  Levels of > 100 000 states of the
    3-server model, one in-model successor per state allowed where four are generated."""
    import tla_rust_amd as amd
    params = [3, 4, 2, 3, 1, 1, 16, 2, 8]   # (with slot capacities: the reference run is the fused one-GPU engine, itself pinned to the oracle)
    eng = amd.Engine("raft", params, table_capacity=1 << 24, arena_capacity=1 << 22, chunk_states=1 << 17, max_distinct=1500000)
    o = eng.run()
    eng.close()
    r = run_dist("hip", 2, "raft", params, tmp_path, {"max_distinct": 1500000, "chunk": 1 << 17, "table": 1 << 24, "arena": 1 << 22,
                                                       "stay_threshold": 200, "rebalance_ratio": 1.5, "packed_fanout": 1, "move_fanout": 1,
                                                       "exchange": exchange})
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o.distinct, o.generated, o.depth, list(o.levels))
    assert o.distinct > 1500000 and max(o.levels) > 100000
    assert r["stats"]["restarts"] >= 1


@pytest.mark.parametrize("world,replicate_until", [(2, 0), (3, 0), (3, 40)])
def test_counterexample_walked_back_across_ranks(oracle, tmp_path, world, replicate_until):
    """README.md:267-321 on several ranks with MC_F_TRACE engines: the parents of states that moved to their owner travelled
    with them (mc_shard_materialise_parents / _ingest_parents), and the failing Assert's behaviour is rebuilt by walking
    (rank, index) pointers with mc_shard_fetch — no single-GPU re-run.  Shortest length (the oracle's), the README's last state."""
    params = [1, 0, 20, 2]   # the README variant: money 1..20, the assertion fails
    o = oracle.oracle_run("pcal_intro", params)
    r = run_dist("hip", world, "pcal_intro", params, tmp_path, {"chunk": 512, "trace": True, "replicate_until": replicate_until})
    assert r["verdict"] == "assert" == o["verdict"]
    tr = r["trace"]
    assert tr is not None and len(tr) == len(o["trace"]) == 6
    assert tr[0][0] == "Initial predicate" and all(isinstance(a, str) and a and a != "?" for a, _ in tr)
    # README.md:305-311: the state the Assert fails in — alice overdrawn, one process at C (which of the equally short
    # behaviours is found depends on which rank reports first: money = <<1, 10>> in the README, any <<m, 10>> with m < 10 here)
    assert "alice_account = -" in tr[-1][1] and '"C"' in tr[-1][1].split("pc = ")[1].split("\n")[0]
    assert len({t for _, t in tr}) == 6


@pytest.mark.parametrize("spec,params,last", [("pcal_intro", [1, 1, 20, 2], "account_total"), ("ssi", [2, 2, 127, 3], "history")])
def test_invariant_counterexample_across_ranks(oracle, tmp_path, spec, params, last):
    """an INVARIANT violated by a successor (pcal_intro's MoneyInvariant, README variant: the violating state is stored nowhere,
    it is rebuilt from its parent with mc_state_apply) and one found when the state is expanded (the SI models' expected
    violations: SLOT_PARENT), three ranks, move and stay levels: the oracle's shortest length, a behaviour without repeats"""
    o = oracle.oracle_run(spec, params)
    assert o["verdict"] == "invariant"
    r = run_dist("hip", 3, spec, params, tmp_path, {"chunk": 256, "trace": True, "stay_threshold": 40, "rebalance_ratio": 2.5}, timeout=900)
    assert r["verdict"] == "invariant"
    tr = r["trace"]
    assert tr is not None and len(tr) == len(o["trace"]) and len({t for _, t in tr}) == len(tr)
    assert tr[0][0] == "Initial predicate" and all(a and a != "?" for a, _ in tr) and last in tr[-1][1]
    if spec == "pcal_intro":   # MoneyInvariant == alice_account + bob_account = account_total: broken in the last state only
        def broken(text):
            v = {ln.split(" = ")[0].strip("/\\ "): ln.split(" = ")[1] for ln in text.splitlines() if " = " in ln}
            return int(v["alice_account"]) + int(v["bob_account"]) != int(v["account_total"])
        assert broken(tr[-1][1]) and not any(broken(t) for _, t in tr[:-1])


@pytest.mark.parametrize("world", [4, 8])
def test_eight_engines_on_one_gpu(oracle, tmp_path, world):
    """N = 4 and 8 HIP step engines (the widths of the driver's scaling run) sharing this one GPU, exchange staged
    through gloo: exercises the 8-owner routing / compaction / block plans of the device code"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=40000)
    r = run_dist("hip", world, "raft", params, tmp_path, {"max_distinct": 40000, "chunk": 1 << 11, "table": 1 << 18, "arena": 1 << 17,
                                                         "stay_threshold": 30, "rebalance_ratio": 2.5}, timeout=900)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert len(r["shares"]) == world and sum(r["shares"]) == o["distinct"]
    assert r["phases"].get("stay_levels", 0) >= 1 and r["phases"].get("move_levels", 0) >= 3


@pytest.mark.parametrize("world,until", [(1, 300), (2, 300), (4, 50)])
def test_replicated_prefix_on_gpu(oracle, tmp_path, world, until):
    """mc_shard_begin_replicated: the fused single-GPU BFS on every rank for the small levels, then slices"""
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params, max_distinct=40000)
    r = run_dist("hip", world, "raft", params, tmp_path, {"max_distinct": 40000, "chunk": 1 << 11, "table": 1 << 18, "arena": 1 << 17,
                                                         "stay_threshold": 60, "rebalance_ratio": 2.0, "replicate_until": until}, timeout=900)
    assert (r["distinct"], r["generated"], r["depth"], r["levels"]) == (o["distinct"], o["generated"], o["depth"], o["levels"])
    assert sum(r["shares"]) == o["distinct"]


def _mc_multi(world, *args, timeout=300):
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    import helpers

    def cmd(port):
        c = [sys.executable]
        if world > 1:
            c += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port)]
        c += ["-m", "tla_rust_amd.mc_multi"] + [str(a) for a in args]
        if world > 1:
            c += ["-backend", "gloo", "-device", "0"]      # one GPU box: the ranks share GPU 0, buckets staged over gloo
        return c
    return helpers.run_with_master_port(cmd, capture_output=True, text=True, timeout=timeout, cwd=root)


@pytest.mark.parametrize("world", [1, 2])
def test_multi_gpu_front_door(world):
    """`python -m tla_rust_amd.mc_multi X.tla` = `tlc X.tla` across ranks: TLC's counter / depth lines, TLC's exit codes,
    the numbers of the one-GPU `mc` (tests/test_frontend.py)."""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    small = ["-chunk", 4096, "-tablelog2", 22, "-arena", 1 << 20]
    p = _mc_multi(world, S / "MCssi.tla", "-config", S / "MCssi_2x2_sym.cfg", *small)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "12558 states generated, 7419 distinct states found, 0 states left on queue." in p.stdout      # SYMMETRY Perms
    assert "The depth of the complete state graph search is 13." in p.stdout
    p = _mc_multi(world, S / "MCraft.tla", "-config", S / "MCraft_small.cfg", *small)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "104515 states generated, 13634 distinct states found, 0 states left on queue." in p.stdout
    p = _mc_multi(world, S / "pluscal" / "peterson.tla", *small)                                         # compiled PlusCal program
    assert p.returncode == 0 and "58 distinct states found" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    p = _mc_multi(world, S / "readme_variant" / "pcal_intro.tla", *small)                                # README.md:267-321: the assertion fails
    assert "Assert evaluated to FALSE" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
    # the behaviour is walked back across the ranks' arenas (no one-GPU re-run): the README's 6 states, ending where the Assert fails
    assert "Error: The behavior up to this point is:" in p.stdout and p.stdout.count("\nState ") == 6       # README.md:270-311
    assert "State 1: <Initial predicate>" in p.stdout and "alice_account = -" in p.stdout.split("State 6:")[1]
    assert "one-GPU run" not in p.stdout
    if world == 1:   # -rerun: the one-GPU report with TLC's positions on top
        q = _mc_multi(world, S / "readme_variant" / "pcal_intro.tla", "-rerun", *small)
        assert '"Failure of assertion at line 16, column 4."' in q.stdout and "counterexample rebuilt by a one-GPU run" in q.stdout
    if world == 1:
        assert p.returncode == 12                     # TLC's exit code for a safety violation
    else:                                             # every rank exits 12; the launcher itself reports 1
        assert p.returncode != 0 and "exitcode  : 12" in p.stderr


@pytest.mark.parametrize("door", ["rccl", "torch"])
def test_mc_gpus_option(door):
    """`mc X.tla -gpus 1`: the C++ CLI forks its ranks and runs the search over the hip-rccl back-end of the C ABI (mc_comm_* /
    mc_shard_run: RCCL, world 1 on this box, no Python in the process); `-torch` hands over to the torch.distributed front door"""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    mc = str(root / "tla_rust_amd" / "_build" / "mc")
    extra = ["-torch"] if door == "torch" else []
    p = subprocess.run([mc, str(root / "specs" / "MCssi.tla"), "-config", str(root / "specs" / "MCssi_2x2_sym.cfg"), "-gpus", "1", *extra,
                        "-tablelog2", "22", "-arena", "1048576", "-chunk", "4096"], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "12558 states generated, 7419 distinct states found, 0 states left on queue." in p.stdout
    assert "The depth of the complete state graph search is 13." in p.stdout
    if door == "rccl":
        assert "over RCCL" in p.stdout
        # a model large enough to leave the replicated prefix: sharded rounds (fixed-capacity exchange over RCCL), the 1-GPU counts
        p = subprocess.run([mc, str(root / "specs" / "MCraft.tla"), "-config", str(root / "specs" / "MCraft.cfg"), "-gpus", "1", "-maxdistinct", "3000000",
                            "-tablelog2", "24", "-arena", "6000000", "-chunk", "262144"], capture_output=True, text=True, timeout=300, cwd="/tmp")
        q = subprocess.run([mc, str(root / "specs" / "MCraft.tla"), "-config", str(root / "specs" / "MCraft.cfg"), "-maxdistinct", "3000000", "-noprogress",
                            "-tablelog2", "24", "-arena", "6000000", "-chunk", "262144"], capture_output=True, text=True, timeout=300, cwd="/tmp")
        assert p.returncode == 0 and q.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:] + q.stdout[-1500:] + q.stderr[-1500:]
        line = next(ln for ln in q.stdout.splitlines() if "distinct states found" in ln)
        assert line in p.stdout, (line, p.stdout[-800:])
        # a violation: TLC's exit code from every rank, the verdict line from rank 0
        p = subprocess.run([mc, str(root / "specs" / "readme_variant" / "pcal_intro.tla"), "-gpus", "1", "-tablelog2", "20", "-arena", "100000"],
                           capture_output=True, text=True, timeout=300, cwd="/tmp")
        assert p.returncode == 12 and "Assert evaluated to FALSE" in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]


# ------------------------------------------------------------------------------------------ the hip-rccl back-end with P > 1
def _fake_env():
    import os
    import helpers
    return dict(os.environ, TLAMC_RCCL=str(helpers.build_fakerccl()))


def _mc(*args, env=None, timeout=600):
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    return subprocess.run([str(root / "tla_rust_amd" / "_build" / "mc")] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout,
                          cwd="/tmp", env=env)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_rccl_loop_with_several_ranks(world):
    """`mc X.tla -gpus P -samedevice`: P forked ranks, mc_comm_create / mc_shard_run over the nccl* entry points (stand-in), the
    C++ level loop with replicated prefix, move and stay levels: the one-GPU `mc`'s counter line — raft (3 servers, 3 M states:
    several rounds per level), the SI model under SYMMETRY, the compiled PlusCal path, Paxos under SYMMETRY with its PROPERTY"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    env = _fake_env()
    cases = [([S / "MCraft.tla", "-config", S / "MCraft.cfg", "-maxdistinct", 3000000, "-tablelog2", 24, "-arena", 6000000, "-chunk", 65536], None),
             ([S / "MCssi.tla", "-config", S / "MCssi_2x2_sym.cfg", "-tablelog2", 22, "-arena", 1 << 20, "-chunk", 4096],
              "12558 states generated, 7419 distinct states found, 0 states left on queue."),
             ([S / "pluscal" / "peterson.tla", "-tablelog2", 20, "-arena", 1 << 18], "58 distinct states found"),
             ([S / "paxos" / "MCPaxos3.tla", "-tablelog2", 22, "-arena", 1 << 20, "-deadlock"], None)]
    for args, want in cases:
        q = _mc(*args, "-noprogress")
        line = next((ln for ln in q.stdout.splitlines() if "distinct states found" in ln), None)
        p = _mc(*args, "-gpus", world, "-samedevice", env=env)
        assert p.returncode == q.returncode, (args, p.stdout[-1500:], p.stderr[-1500:], q.stdout[-500:], q.stderr[-500:])
        assert line and line in p.stdout, (line, p.stdout[-800:], p.stderr[-800:])
        assert want is None or want in p.stdout
        assert f"({world} GPUs over RCCL" in p.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_native_counterexample_across_ranks(world):
    """README.md:267-321 through `mc -gpus P`: every rank exits 12, rank 0 prints the 6-state behaviour walked back across the
    ranks by mc_shard_trace (no Python, no one-GPU re-run); an INVARIANT broken by a successor is rebuilt from its parent"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    p = _mc(S / "readme_variant" / "pcal_intro.tla", "-gpus", world, "-samedevice", "-tablelog2", 20, "-arena", 100000, env=_fake_env())
    assert p.returncode == 12, p.stdout[-1500:] + p.stderr[-1500:]
    assert "Assert evaluated to FALSE" in p.stdout and "Error: The behavior up to this point is:" in p.stdout
    assert p.stdout.count("\nState ") == 6 and "State 1: <Initial predicate>" in p.stdout and "alice_account = -" in p.stdout.split("State 6:")[1]


def test_native_loop_survives_a_failing_rank():
    """a rank-local failure (here: an arena too small for the rank's share, MC_EARENA at the end of a level) must
    not leave the other ranks waiting in a collective — every rank's status travels with the per-level all-gather, all leave
    together, `mc -gpus P` exits 1 (and would SIGTERM the siblings of a rank that died)"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    p = _mc(S / "MCraft.tla", "-config", S / "MCraft.cfg", "-gpus", 2, "-samedevice", "-maxdistinct", 3000000, "-tablelog2", 24, "-arena", 700000,
            "-chunk", 65536, env=_fake_env(), timeout=300)
    assert p.returncode == 1 and ("arena" in p.stderr.lower() or "failed" in p.stderr.lower()), p.stdout[-800:] + p.stderr[-1500:]


def test_native_default_chunk_beyond_one_launch():
    """`mc X.tla -gpus P` WITHOUT -chunk on a model whose per-rank frontier exceeds the engine's default chunk
    (2^18 states): the loop clamps its rounds to what one launch of the engine takes"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    args = [S / "MCraft.tla", "-config", S / "MCraft.cfg", "-maxdistinct", 12000000, "-tablelog2", 26, "-arena", 30000000]
    q = _mc(*args, "-noprogress")
    p = _mc(*args, "-gpus", 2, "-samedevice", env=_fake_env())
    line = next(ln for ln in q.stdout.splitlines() if "distinct states found" in ln)
    assert p.returncode == 0 and line in p.stdout, (line, p.stdout[-800:], p.stderr[-1500:])


@pytest.mark.parametrize("world", [2, 8])
def test_bench_contract_invocation_with_several_ranks(world):
    """`python bench.py --gpus N` — the contract invocation, no launcher: bench.py spawns its N ranks, the ranks build the RCCL
    communicator (id through a file in a directory of the job's own: no port to lose, round 3's GPUTEST failure), run mc_shard_run on
    the COMPLETE bench graph and rank 0 prints the line.  Here the
    ranks share this one GPU (--share-gpu + stand-in): a functional run, the line says so.  P = 8: the frontier stays balanced."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--share-gpu", "--workload", "k10",
                        "--exchange", "exact"], capture_output=True, text=True, timeout=1500, env=_fake_env(), cwd=str(root))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
    c = line["config"]
    assert line["n_gpus"] == world and (c["distinct"], c["generated"], c["depth"], c["verdict"]) == (102586254, 1217433925, 33, "ok")
    assert len(c["shares"]) == world and sum(c["shares"]) == 102586254 and c["levels"]["stay_levels"] >= 10
    assert c["frontier_imbalance"] <= 1.25 and max(c["shares"]) <= 1.25 * 102586254 / world
    assert "xgmi" in line and line["xgmi"]["sent_bytes_per_step_per_gpu"] > 0
    # the default (exact) form of the stay levels moves 9 bytes per routed candidate and nothing else
    x = line["xgmi"]
    assert x["exchange"] == "exact" and x["routed_candidates_per_step"] > 10 ** 8 and x["fp_answer_bytes_per_step"] == x["model_bytes_per_step"] and x["sent_over_model"] == 1.0


@pytest.mark.parametrize("workload,levels,prefix", [
    ("raft5", 13, [1, 6, 40, 205, 775, 2851, 10000, 32015, 97215, 287510, 816406, 2225540, 5913945]),
    ("ssi4x3", 9, [1, 4, 32, 264, 2532, 24576, 236844, 2189052, 18810792]),   # (eight ranks replicate up to a frontier of 2^18: level 8 is the first sharded expansion)
])
def test_the_eight_rank_deep_command_lines_of_configs_4_and_5_at_a_reduced_budget(workload, levels, prefix):
    """`python bench.py --gpus 8 --workload raft5 | ssi4x3` — eight ranks, --deep by default: the budgets the first
    real 8-GPU run will use — here exactly those command lines with `--levels L` (a prefix of the same golden, every level gated by
    bench.py itself; seen-set and arenas sized for it) and the eight ranks on this one GPU (--share-gpu + stand-in): the N-rank engine
    walks BASELINE config 4's five-server raft model and config 5's 4 x 3 snapshot-isolation model through replicated, stay and move
    levels and reproduces the oracle's per-level counts."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--workload", workload, "--steps", "1", "--warmup", "0", "--share-gpu",
                        "--levels", str(levels)], capture_output=True, text=True, timeout=1500, env=_fake_env(), cwd=str(root))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
    c = line["config"]
    assert line["n_gpus"] == 8 and c["distinct"] == sum(prefix) and c["depth"] == levels and c["verdict"] == "budget"
    assert "REDUCED" in c["workload"] and "NOT_A_MEASUREMENT" in c
    assert len(c["shares"]) == 8 and sum(c["shares"]) >= sum(prefix) and c["levels"]["stay_levels"] + c["levels"]["move_levels"] >= 1


def test_bench_under_the_drivers_launcher():
    """the driver's N > 1 command line, word for word: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` — the communicator id travels through the launcher's own TCP store
    (the agent listens before any rank starts); here with the two ranks on this one GPU (--share-gpu + stand-in)"""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    import helpers
    p = helpers.run_with_master_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                   "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--share-gpu",
                                                   "--workload", "k10"],
                                     capture_output=True, text=True, timeout=1500, env=_fake_env(), cwd=str(root))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{")))
    c = line["config"]
    assert (line["n_gpus"], line["steps"], line["warmup"], line["scaling"]) == (2, 2, 1, "strong")
    assert (c["distinct"], c["generated"], c["depth"], c["verdict"]) == (102586254, 1217433925, 33, "ok") and sum(c["shares"]) == 102586254
    # round 5: the N-rank line describes itself — the default `--exchange auto` ran one untimed step in the exact
    # and in the measured form and timed the faster one; per rank: the GPU time of its three kinds of kernels and the host time of its
    # level loop inside engine calls and inside collectives
    assert c["exchange"] in ("exact", "measured") and set(c["exchange_trial_ms"]) == {"exact", "measured"}
    assert c["exchange"] == min(c["exchange_trial_ms"], key=c["exchange_trial_ms"].get) == line["xgmi"]["exchange"]
    pr = line["per_rank"]
    for k in ("expand_ms", "probe_ms", "keep_ms", "engine_host_ms", "collective_host_ms", "collectives", "host_ms_per_round"):
        assert len(pr[k]) == 2 and all(v > 0 for v in pr[k]), (k, pr)
    assert pr["rounds"] > 0 and max(pr["expand_ms"]) < 1e3 * 60


def test_bench_under_a_launcher_at_world_size_1_is_the_fused_line():
    """`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`: the N = 1 point of a launched scaling run is the
    one-GPU line itself (fused engine: a roofline object with its dominant kernel, no exchange), not the N-rank engine at world size 1
    (+9 % in round 4); --force-shard still runs that one (RCCL at world size 1: the real librccl accepts one rank per device)."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    lines = []
    import helpers
    for extra in ([], ["--force-shard"]):
        p = helpers.run_with_master_port(lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                                                       "--master-port", str(port), str(root / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "k10",
                                                       "--no-cpu-baseline"] + extra,
                                         capture_output=True, text=True, timeout=900, cwd=str(root))
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        lines.append(json.loads(next(ln for ln in p.stdout.splitlines() if ln.startswith("{"))))
    fused, shard = lines
    assert fused["n_gpus"] == 1 and fused["roofline"]["kernel"].startswith("k_expand_family<SpecRaft<3>") and "xgmi" not in fused
    assert fused["config"]["distinct"] == shard["config"]["distinct"] == 102586254
    assert "xgmi" in shard and shard["config"]["exchange"] == "exact" and len(shard["per_rank"]["expand_ms"]) == 1
    assert fused["ms_per_step"] < 1.05 * shard["ms_per_step"]   # (the fused engine is never the slower one)


# ------------------------------------------------------------------------------------------ one checkpoint file per rank
@pytest.mark.parametrize("world", [1, 2, 4])
def test_native_checkpoint_per_rank_and_recover(world, tmp_path):
    """`mc X.tla -gpus P -checkpoint F` after a budget stop writes F.rank<r>of<P> on every rank (arena, parent pointers, the rank's
    seen-set slice as it lies in HBM, counters, level table; testout1:10's line is printed once all files are written);
    `mc X.tla -gpus P -recover F` in NEW processes continues with the unexpanded frontier: the counter line of the continued run
    equals the uninterrupted one-GPU run's.  The stop lies in the sharded part (3 M of 6.7 M states: stay levels)"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    env = _fake_env()
    base = [S / "MCraft.tla", "-config", S / "MCraft.cfg", "-tablelog2", 24, "-arena", 12000000, "-chunk", 65536]
    q = _mc(*base, "-maxdistinct", 6000000, "-noprogress")
    line = next(ln for ln in q.stdout.splitlines() if "distinct states found" in ln)
    ck = tmp_path / "run"
    p = _mc(*base, "-maxdistinct", 1500000, "-gpus", world, "-samedevice", "-checkpoint", ck, env=env)
    assert p.returncode == 0 and f"-- Checkpointing of run {ck} completed." in p.stdout, p.stdout[-1500:] + p.stderr[-1500:]
    files = [tmp_path / f"run.rank{r}of{world}" for r in range(world)]
    assert all(f.stat().st_size > (1 << 24) * 8 for f in files)           # at least the seen-set slice
    p = _mc(*base, "-maxdistinct", 6000000, "-gpus", world, "-samedevice", "-recover", ck, env=env)
    assert p.returncode == 0 and line in p.stdout, (line, p.stdout[-800:], p.stderr[-1500:])
    # another world size, or a table of another size, is refused on every rank
    if world == 2:
        p = _mc(*base, "-gpus", 4, "-samedevice", "-recover", ck, env=env)
        assert p.returncode == 1 and "cannot read" in p.stderr
        p = _mc(S / "MCraft.tla", "-config", S / "MCraft.cfg", "-tablelog2", 25, "-arena", 12000000, "-gpus", 2, "-samedevice", "-recover", ck, env=env)
        assert p.returncode == 1 and "table_capacity differs" in p.stderr


def test_native_counterexample_after_recover(tmp_path):
    """parent pointers (index, slot, RANK) are part of a rank's file: a violation found after -recover is walked back across the
    ranks into the checkpointed part, down to the initial state"""
    from pathlib import Path
    S = Path(__file__).resolve().parent.parent / "specs"
    env = _fake_env()
    base = [S / "readme_variant" / "pcal_intro.tla", "-tablelog2", 20, "-arena", 100000, "-gpus", 2, "-samedevice"]
    ck = tmp_path / "run"
    p = _mc(*base, "-maxlevels", 4, "-checkpoint", ck, env=env)
    assert p.returncode == 0 and "Checkpointing of run" in p.stdout, p.stdout[-800:] + p.stderr[-800:]
    p = _mc(*base, "-recover", ck, env=env)
    assert p.returncode == 12 and p.stdout.count("\nState ") == 6 and "State 1: <Initial predicate>" in p.stdout, p.stdout[-1500:] + p.stderr[-800:]


def test_torch_door_checkpoint_per_rank(oracle, tmp_path):
    """the same through the torch.distributed transport (ShardedChecker.checkpoint / restore over gloo, two HIP engines on one GPU)"""
    from test_sharded_gloo import run_dist
    params = [2, 2, 2, 9, 2, 1]
    o = oracle.oracle_run("raft", params)
    r = run_dist("hip", 2, "raft", params, tmp_path, {"max_levels": 12, "chunk": 2048, "checkpoint": str(tmp_path / "ck"), "stay_threshold": 50,
                                                        "rebalance_ratio": 1.6, "trace": True})
    assert r["first"]["verdict"] == "budget"
    assert (r["distinct"], r["generated"], r["depth"], r["levels"], r["verdict"]) == (o["distinct"], o["generated"], o["depth"], o["levels"], o["verdict"])
