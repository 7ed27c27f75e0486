"""Front-end of libtlamc.so: the .cfg parser (ConfigFileGrammar.tla:4-32) on every cfg file of the
reference tree, spec resolution, and (GPU) the `tlc X.tla` end-to-end path with its report text."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")


@pytest.fixture(scope="module")
def amd():
    import tla_rust_amd
    return tla_rust_amd


def test_parses_every_reference_cfg(amd):
    """all 22 cfg files of the reference tree (needs /root/reference)."""
    if not REF.exists():
        pytest.skip("reference tree not present on this box")
    files = sorted(REF.rglob("*.cfg")) + sorted(REF.rglob("*.cfg.alt"))
    assert len(files) == 22
    for f in files:
        c = amd.cfg_parse(f.read_text())
        assert isinstance(c["CONSTANTS"], list), f


FEATURES = r'''
(* boxed
   multi-line comment *)
  SPECIFICATION Spec  \* trailing comment
CONSTANTS
  a1=a1  a2=a2        \* several on one line, no spaces
  Acceptor <- MCAcceptor
  Ballot <-[Voting] MCBallot
  Proc = {p1, p2}     Val = {"a", "b", "c"}   MaxQLen = 1   Neg = -3
  Empty = {}
INVARIANT Inv1 Inv2
INVARIANTS Inv3
PROPERTY Live
CONSTRAINT Constraint
ACTION-CONSTRAINT ActC
SYMMETRY Perms
VIEW V
INIT Init
NEXT Next
'''


def test_cfg_features(amd):
    c = amd.cfg_parse(FEATURES)
    assert c["SPECIFICATION"] == "Spec" and c["INIT"] == "Init" and c["NEXT"] == "Next"
    assert c["INVARIANTS"] == ["Inv1", "Inv2", "Inv3"] and c["PROPERTIES"] == ["Live"]
    assert c["CONSTRAINTS"] == ["Constraint"] and c["ACTION_CONSTRAINTS"] == ["ActC"]
    assert c["SYMMETRY"] == "Perms" and c["VIEW"] == "V"
    k = {x["name"]: x for x in c["CONSTANTS"]}
    assert k["a1"]["value"] == {"model_value": "a1"} and k["a2"]["value"] == {"model_value": "a2"}
    assert k["Acceptor"]["replace_by"] == "MCAcceptor" and "module" not in k["Acceptor"]
    assert k["Ballot"]["replace_by"] == "MCBallot" and k["Ballot"]["module"] == "Voting"
    assert k["Proc"]["value"] == {"set": [{"model_value": "p1"}, {"model_value": "p2"}]}
    assert k["Val"]["value"] == {"set": ["a", "b", "c"]} and k["MaxQLen"]["value"] == 1 and k["Neg"]["value"] == -3
    assert k["Empty"]["value"] == {"set": []}


@pytest.mark.parametrize("bad", ["SPECIFICATION", "CONSTANTS x", "CONSTANT x = {a, }", "FOO Bar", "(* never closed", "CONSTANT x <- 3",
                                 'CONSTANT s = "open'])
def test_cfg_rejects_malformed(amd, bad):
    with pytest.raises(amd.McError) as e:
        amd.cfg_parse(bad)
    assert e.value.code == -8


def test_cfg_check_deadlock_statement(amd):
    """TLC2's `CHECK_DEADLOCK FALSE` (= the command line's -deadlock; not in the 2001 grammar of TLC/ConfigFileGrammar.tla:4-32):
    specs/pluscal/paxos_soup.cfg uses it — its acceptors never stop"""
    assert amd.cfg_parse("SPECIFICATION Spec\nCHECK_DEADLOCK FALSE\nINVARIANT I\n")["CHECK_DEADLOCK"] is False
    assert amd.cfg_parse("CHECK_DEADLOCK TRUE\n")["CHECK_DEADLOCK"] is True
    assert "CHECK_DEADLOCK" not in amd.cfg_parse("SPECIFICATION Spec\n")
    with pytest.raises(amd.McError) as e:
        amd.cfg_parse("CHECK_DEADLOCK maybe\n")
    assert e.value.code == -8


def test_empty_cfg_is_valid(amd):
    c = amd.cfg_parse("(* only a comment *)\n\\* and another\n")
    assert c["SPECIFICATION"] == "" and c["CONSTANTS"] == []


def test_symmetry_resolution(amd):
    """SYMMETRY is accepted for the snapshot-isolation models (the spec's run-book makes Key and TxnId symmetry sets,
    serializableSnapshotIsolation.tla:38-44) and refused — never ignored — for every other lowering."""
    cfg = (ROOT / "specs" / "MCssi_2x2_sym.cfg").read_text()
    sid, params = amd.spec_resolve("MCssi", cfg)
    assert sid == amd.SPEC_IDS["ssi"] and params[:2] == [2, 2] and params[5] == 3
    sid, params = amd.spec_resolve("MCssi", (ROOT / "specs" / "MCssi_2x2.cfg").read_text())
    assert params[5] == 0
    with pytest.raises(amd.McError):
        amd.spec_resolve("MCraft", (ROOT / "specs" / "MCraft_small.cfg").read_text() + "\nSYMMETRY Perms\n")
    with pytest.raises(amd.McError):
        amd.spec_resolve("pcal_intro", (ROOT / "specs" / "pcal_intro.cfg").read_text() + "\nSYMMETRY Perms\n")


def test_constraint_is_never_silently_ignored(amd):
    """a CONSTRAINT in the cfg of a hand-lowered PlusCal module is not dropped: the registry declines (the CLI then compiles
    the module); MCraft accepts exactly its own StateConstraint; the SI models define none"""
    for module, cfg in (("pcal_intro", "pcal_intro.cfg"), ("atomic_add", "atomic_add.cfg")):
        with pytest.raises(amd.McError):
            amd.spec_resolve(module, (ROOT / "specs" / cfg).read_text() + "\nCONSTRAINT Small\n")
    with pytest.raises(amd.McError):
        amd.spec_resolve("MCraft", (ROOT / "specs" / "MCraft_small.cfg").read_text() + "\nCONSTRAINT Other\n")
    with pytest.raises(amd.McError):
        amd.spec_resolve("MCssi", (ROOT / "specs" / "MCssi_2x2.cfg").read_text() + "\nCONSTRAINT Small\n")


def test_resolve_files_for_a_multi_gpu_host(amd):
    """mc_resolve_files = the front half of `tlc X.tla` (no GPU needed): the descriptor every rank of a sharded run creates
    its engine from — same registry, same text checks, same PlusCal compiler as the one-GPU CLI."""
    S = ROOT / "specs"
    r = amd.ResolvedSpec(S / "MCssi.tla", S / "MCssi_2x2_sym.cfg")
    assert (r.spec, r.params[:2], r.params[5]) == ("ssi", [2, 2], 3)
    r = amd.ResolvedSpec(S / "pcal_intro.tla")                       # X.cfg beside X.tla (README.md:356)
    assert (r.spec, r.params) == ("pcal_intro", [0, 1, 20, 2])
    r = amd.ResolvedSpec(S / "readme_variant" / "pcal_intro.tla")    # README.md:220-243: labels A / B
    assert (r.spec, r.params[0]) == ("pcal_intro", 1)
    r = amd.ResolvedSpec(S / "MCraft.tla", S / "MCraft_small.cfg")
    assert r.spec == "raft" and r.params[0] == 2
    for f in ("peterson.tla", "growing_counters.tla"):               # no hand lowering: compiled (the handle keeps the program)
        r = amd.ResolvedSpec(S / "pluscal" / f)
        assert r.spec == "pcal" and r.params[0] != 0
        r.close()
    r = amd.ResolvedSpec(S / "pcal_intro.tla", generic=True)
    assert r.spec == "pcal"
    r.close()
    with pytest.raises(amd.McError):
        amd.ResolvedSpec(ROOT / "include" / "tlamc.h")
    with pytest.raises(amd.McError):
        amd.ResolvedSpec(S / "MCssi.tla", S / "MCraft.cfg")


def test_multi_gpu_front_door_report_and_options():
    from tla_rust_amd import mc_multi
    from tla_rust_amd.binding import Result
    o = mc_multi.parse(["X.tla", "-config", "Y.cfg", "-maxlevels", "7", "-workers", "8", "-backend", "gloo", "-device", "0"])
    assert (o["tla"], o["config"], o["maxlevels"], o["backend"], o["device"]) == ("X.tla", "Y.cfg", 7, "gloo", 0)
    with pytest.raises(SystemExit):
        mc_multi.parse(["-nonsense"])
    assert mc_multi.parse(["X.tla", "-exchange", "measured"])["exchange"] == "measured" and "exchange" not in mc_multi.parse(["X.tla"])
    assert mc_multi.parse(["X.tla", "-fanout", "40"])["fanout"] == 40
    with pytest.raises(SystemExit):
        mc_multi.parse(["X.tla", "-exchange", "sideways"])
    rep = mc_multi.report(Result(distinct=3800, generated=5850, queue_left=0, depth=5, verdict="ok", levels=[400, 1250, 900, 800, 450]), 8, 0.5)
    assert "Finished computing initial states: 400 distinct states generated." in rep
    assert "5850 states generated, 3800 distinct states found, 0 states left on queue." in rep         # README.md:319 format
    assert "The depth of the complete state graph search is 5." in rep                                 # README.md:320 format
    rep = mc_multi.report(Result(distinct=10, generated=20, queue_left=3, depth=4, verdict="assert", levels=[1, 2, 3, 4]), 2, 0.1)
    assert "Assert evaluated to FALSE" in rep and "no behavior" in rep  # engines without parent pointers
    rep = mc_multi.report(Result(distinct=10, generated=20, queue_left=3, depth=4, verdict="assert", levels=[1, 2, 3, 4]), 2, 0.1,
                          [("Initial predicate", "/\\ x = 0"), ("Step", "/\\ x = 1")])
    assert "Error: The behavior up to this point is:" in rep and "State 1: <Initial predicate>" in rep and "State 2: <Step>" in rep  # README.md:270-311


def test_mc_gpus_without_a_gpu_refuses_loudly(tmp_path):
    """`mc X.tla -gpus P` starts P ranks of itself over the hip-rccl back-end of the C ABI; `-gpus P -torch` replaces itself by
    torch.distributed.run -m tla_rust_amd.mc_multi (found through the binary's own location, whatever the working directory).
    Without a GPU both refuse loudly — there is no CPU fallback."""
    import os
    if os.path.exists("/dev/kfd"):   # (not torch.cuda.is_available(): that initialises torch's HIP runtime inside this long-lived
        pytest.skip("GPU present: covered by tests/test_gpu_sharded.py")   # process, which without a device node leaves threads behind)
    import tla_rust_amd.build as b
    b.build()
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    args = [str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(ROOT / "specs" / "MCssi_2x2_sym.cfg"), "-gpus", "1"]
    p = subprocess.run(args, capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert p.returncode == 1 and "mc[0]:" in p.stderr and "states generated" not in p.stdout
    p = subprocess.run(args + ["-torch"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert p.returncode != 0 and "no HIP device visible" in p.stderr + p.stdout
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-gpus", "2", "-dump", "x"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "not available with -gpus" in p.stderr
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-gpus", "0"], capture_output=True, text=True)
    assert p.returncode == 1
    # -exchange names one of the three forms of a stay level's exchange (include/tlamc.h); anything else is refused by every rank
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-gpus", "2", "-exchange", "sideways"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "-exchange exact | measured | packed" in p.stderr
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-gpus", "2", "-fanout", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "-fanout needs a number" in p.stderr


def test_cli_is_built():
    import tla_rust_amd.build as b
    b.build()
    assert (ROOT / "tla_rust_amd" / "_build" / "mc").exists()


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_tlc_dropin_committed_pcal_intro(amd):
    r, rep = amd.check_files(ROOT / "specs" / "pcal_intro.tla")
    assert r.verdict == "ok" and (r.distinct, r.generated, r.depth) == (3800, 5850, 5)
    assert "Finished computing initial states: 400 distinct states generated." in rep
    assert "Model checking completed. No error has been found." in rep                       # testout2:260
    assert "5850 states generated, 3800 distinct states found, 0 states left on queue." in rep   # README.md:319 format
    assert "The depth of the complete state graph search is 5." in rep                        # README.md:320 format


@pytest.mark.gpu
def test_tlc_dropin_readme_variant_report(amd):
    r, rep = amd.check_files(ROOT / "specs" / "readme_variant" / "pcal_intro.tla")
    assert r.verdict == "assert" and r.trace_len == 6
    lines = rep.splitlines()
    assert lines[1] == "The first argument of Assert evaluated to FALSE; the second argument was:"   # README.md:268
    assert lines[2] == '"Failure of assertion at line 16, column 4."'                                # README.md:269
    assert lines[3] == "Error: The behavior up to this point is:"                                    # README.md:270
    assert lines[4] == "State 1: <Initial predicate>"                                                # README.md:271
    assert rep.count("\nState ") == 6
    # action locations exactly as TLC printed them for this file layout (README.md:278,285,292,299,306)
    hdr = [l for l in lines if l.startswith("State ") and "Initial" not in l]
    allowed = {"Transfer": "<Action line 35, col 19 to line 40, col 42 of module pcal_intro>",
               "A": "<Action line 42, col 12 to line 45, col 63 of module pcal_intro>",
               "B": "<Action line 47, col 12 to line 50, col 65 of module pcal_intro>"}
    assert len(hdr) == 5 and all(h.split(": ", 1)[1] in allowed.values() for h in hdr)
    assert sorted(h.split(": ", 1)[1] for h in hdr) == sorted([allowed["Transfer"]] * 2 + [allowed["A"]] * 2 + [allowed["B"]])
    # README.md:313-316
    i = lines.index("Error: The error occurred when TLC was evaluating the nested")
    assert lines[i + 1] == "expressions at the following positions:"
    assert lines[i + 2] == "0. Line 52, column 15 to line 52, column 28 in pcal_intro"
    assert lines[i + 3] == "1. Line 53, column 15 to line 54, column 66 in pcal_intro"
    assert "The depth of the complete state graph search is 7." in rep                               # README.md:320


@pytest.mark.gpu
def test_cli_exit_codes_and_raft(amd):
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    p = subprocess.run([str(mc), str(ROOT / "specs" / "atomic_add.tla")], capture_output=True, text=True)
    assert p.returncode == 0 and "7 states generated, 5 distinct states found, 0 states left on queue." in p.stdout
    p = subprocess.run([str(mc), str(ROOT / "specs" / "readme_variant" / "pcal_intro.tla")], capture_output=True, text=True)
    assert p.returncode == 12
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCraft.tla"), "-config", str(ROOT / "specs" / "MCraft_small.cfg")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "104515 states generated, 13634 distinct states found, 0 states left on queue." in p.stdout
    p = subprocess.run([str(mc), str(ROOT / "specs" / "atomic_add_n.tla")], capture_output=True, text=True)
    assert p.returncode == 0 and f"{2**20 + 1} distinct states found" in p.stdout
    p = subprocess.run([str(mc), str(ROOT / "include" / "tlamc.h")], capture_output=True, text=True)
    assert p.returncode == 1


@pytest.mark.gpu
def test_cli_ssi_symmetry(amd, tmp_path):
    """`SYMMETRY Perms` end to end: which sets Perms permutes is read from the model module."""
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(ROOT / "specs" / "MCssi_2x2_sym.cfg")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "12558 states generated, 7419 distinct states found, 0 states left on queue." in p.stdout
    assert "The depth of the complete state graph search is 13." in p.stdout
    for name, want in (("TxnPerms", "25062 states generated, 14815 distinct"), ("KeyPerms", "25113 states generated, 14837 distinct")):
        cfg = tmp_path / f"{name}.cfg"
        cfg.write_text((ROOT / "specs" / "MCssi_2x2_sym.cfg").read_text().replace("SYMMETRY Perms", f"SYMMETRY {name}"))
        p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(cfg)], capture_output=True, text=True)
        assert p.returncode == 0 and want in p.stdout, p.stdout + p.stderr
    cfg = tmp_path / "bad.cfg"
    cfg.write_text((ROOT / "specs" / "MCssi_2x2_sym.cfg").read_text().replace("SYMMETRY Perms", "SYMMETRY WellFormed"))
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(cfg)], capture_output=True, text=True)
    assert p.returncode != 0 and "Permutations" in (p.stdout + p.stderr)


@pytest.mark.gpu
def test_cli_ssi(amd):
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(ROOT / "specs" / "MCssi_2x2.cfg")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "50121 states generated, 29629 distinct states found, 0 states left on queue." in p.stdout
    assert "The depth of the complete state graph search is 13." in p.stdout


@pytest.mark.gpu
def test_progress_lines_like_tlc(amd):
    """testout2:4-259: "Progress(5): 6117 states generated, 195 distinct states found, 1 states left on queue." while the search
    runs (mc prints them by default, at most one per second; TLAMC_PROGRESS_INTERVAL=0 reports after every host-visible level)"""
    import os
    import re
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    env = dict(os.environ, TLAMC_PROGRESS_INTERVAL="0")
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(ROOT / "specs" / "MCssi_2x2.cfg")], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr
    rows = [tuple(map(int, m.groups())) for m in re.finditer(r"^Progress\((\d+)\): (\d+) states generated, (\d+) distinct states found, (\d+) states left on queue\.$", p.stdout, re.M)]
    assert len(rows) >= 2 and rows == sorted(rows)                      # every counter only grows
    assert rows[-1][0] == 13 and rows[-1][1:] == (50121, 29629, 0)      # the last report is the final state of the search
    assert p.stdout.index("Progress(") < p.stdout.index("Model checking completed")
    p = subprocess.run([str(mc), str(ROOT / "specs" / "MCssi.tla"), "-config", str(ROOT / "specs" / "MCssi_2x2.cfg"), "-noprogress"], capture_output=True, text=True, env=env)
    assert p.returncode == 0 and "Progress(" not in p.stdout
    # the same through the C ABI
    seen = []
    eng = amd.Engine("ssi", [2, 2, 127, 0], table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    eng.set_progress(lambda lv, g, d, q: seen.append((lv, g, d, q)), 0.0)
    r = eng.run()
    assert seen and seen[-1] == (r.depth, r.generated, r.distinct, 0) and seen == sorted(seen)
    eng.close()


def test_wrapper_and_extended_module_are_verified_or_refused(amd, tmp_path, monkeypatch):
    """`mc specs/MCraft.tla` must not hand a changed StateConstraint / raft.tla to the hard-wired lowering.
    The wrapper body is hashed; the EXTENDed module must be found and match, or the run is refused unless the caller
    explicitly accepts the built-in lowering (MC_F_UNVERIFIED / -unverified / TLAMC_UNVERIFIED=1)."""
    import shutil
    S = ROOT / "specs"
    cfg = S / "MCraft_small.cfg"
    monkeypatch.delenv("TLA_PATH", raising=False)
    monkeypatch.delenv("TLAMC_UNVERIFIED", raising=False)
    # 1. EXTENDed module not found: refused, accepted only when asked for
    with pytest.raises(amd.McError) as e:
        amd.ResolvedSpec(S / "MCraft.tla", cfg)
    assert e.value.code == -9 and "raft" in str(e.value)
    with pytest.raises(amd.McError):
        amd.ResolvedSpec(S / "MCssi.tla", S / "MCssi_2x2.cfg")
    r = amd.ResolvedSpec(S / "MCraft.tla", cfg, unverified=True)
    assert r.spec == "raft" and r.params[9] == 64
    monkeypatch.setenv("TLAMC_UNVERIFIED", "1")
    assert amd.ResolvedSpec(S / "MCssi.tla", S / "MCssi_2x2.cfg").spec == "ssi"
    # 2. a one-character change of the wrapper is refused even then
    for name, old, new in (("MCraft", "currentTerm[i] <= MaxTerm", "currentTerm[i] < MaxTerm"), ("MCssi", "CahillSerializable(history)", "CahillSerializable(history) ")):
        d = tmp_path / name
        d.mkdir()
        text = (S / f"{name}.tla").read_text()
        assert old in text
        (d / f"{name}.tla").write_text(text.replace(old, new))
        c2 = cfg if name == "MCraft" else S / "MCssi_2x2.cfg"
        if old.strip() == new.strip():   # whitespace only: still the same wrapper
            assert amd.ResolvedSpec(d / f"{name}.tla", c2).spec == "ssi"
        else:
            with pytest.raises(amd.McError) as e:
                amd.ResolvedSpec(d / f"{name}.tla", c2)
            assert e.value.code == -9 and "wrapper" in str(e.value)
    # 3. a changed raft.tla beside the wrapper is refused (build container only: needs the reference's file)
    ref = Path("/root/reference/examples/raft.tla")
    if ref.exists():
        d = tmp_path / "with_raft"
        d.mkdir()
        shutil.copy(S / "MCraft.tla", d / "MCraft.tla")
        (d / "raft.tla").write_text(ref.read_text().replace("clientRequests < MaxClientRequests", "clientRequests <= MaxClientRequests"))
        with pytest.raises(amd.McError) as e:
            amd.ResolvedSpec(d / "MCraft.tla", cfg)
        assert e.value.code == -9 and "differs" in str(e.value)
        (d / "raft.tla").write_text(ref.read_text())
        assert amd.ResolvedSpec(d / "MCraft.tla", cfg).spec == "raft"


def test_edited_atomic_add_n_takes_the_compiled_path(amd, tmp_path):
    """a module NAMED atomic_add_n whose algorithm differs from specs/atomic_add_n.tla is not the N-adder
    hand lowering: it is compiled like any other PlusCal module"""
    S = ROOT / "specs"
    r = amd.ResolvedSpec(S / "atomic_add_n.tla")
    assert r.spec == "atomic_add"
    text = (S / "atomic_add_n.tla").read_text()
    assert "await global_counter = N" in text
    (tmp_path / "atomic_add_n.tla").write_text(text.replace("await global_counter = N", "await global_counter >= N - 1"))
    (tmp_path / "atomic_add_n.cfg").write_text((S / "atomic_add_n.cfg").read_text())
    r = amd.ResolvedSpec(tmp_path / "atomic_add_n.tla")
    assert r.spec == "pcal"
    r.close()


def test_paxos_models_are_resolved_from_the_module_text(amd, tmp_path, monkeypatch):
    """examples/Paxos/MCVoting.cfg:3-6, MCPaxos.cfg:4-9: the cfg replaces Acceptor / Value / Quorum / Ballot by DEFINITIONS of
    the model module, so the sizes come from the text (MCAcceptor == {a1, a2, a3}; MCPaxos.tla:7-9 as committed: one
    acceptor, one value).  Voting.tla / Paxos.tla / Consensus.tla are verified by hash where they are found."""
    import shutil
    P = ROOT / "specs" / "paxos"
    monkeypatch.delenv("TLA_PATH", raising=False)
    monkeypatch.delenv("TLAMC_UNVERIFIED", raising=False)
    with pytest.raises(amd.McError) as e:                      # Paxos.tla not found: refused
        amd.ResolvedSpec(P / "MCPaxos3.tla")
    assert e.value.code == -9 and "Paxos" in str(e.value)
    r = amd.ResolvedSpec(P / "MCPaxos3.tla", unverified=True)
    assert (r.spec, r.params) == ("paxos", [0, 3, 2, 2, 15, 3, 1, 3, 3, 5, 6])
    r = amd.ResolvedSpec(P / "MCVoting3.tla", unverified=True)
    assert (r.spec, r.params) == ("paxos", [1, 3, 2, 3, 1, 3, 1, 3, 3, 5, 6])
    with pytest.raises(amd.McError) as e:                      # quorums that do not intersect: TLC evaluates the ASSUME first
        amd.ResolvedSpec(P / "MCVotingBadQuorum.tla", unverified=True)
    assert "QuorumAssumption" in str(e.value)
    with pytest.raises(amd.McError) as e:                      # another next-state relation than Paxos.tla's own
        amd.ResolvedSpec(P / "MCPaxosBad.tla", unverified=True)
    assert e.value.code == -9
    # an invariant / property the lowering does not implement is refused, never ignored
    d = tmp_path / "m"
    d.mkdir()
    shutil.copy(P / "MCPaxos3.tla", d / "MCPaxos3.tla")
    (d / "MCPaxos3.cfg").write_text((P / "MCPaxos3.cfg").read_text().replace("Inv4", "Inv4 MCLiveness"))
    with pytest.raises(amd.McError):
        amd.ResolvedSpec(d / "MCPaxos3.tla", unverified=True)
    (d / "MCPaxos3.cfg").write_text((P / "MCPaxos3.cfg").read_text().replace("Ballot <-[Voting] MCBallot", ""))
    with pytest.raises(amd.McError) as e:
        amd.ResolvedSpec(d / "MCPaxos3.tla", unverified=True)
    assert "<-[Voting]" in str(e.value)
    ref = REF / "examples" / "Paxos"
    if ref.exists():   # build container: the reference's own model files, and an edited Voting.tla
        r = amd.ResolvedSpec(ref / "MCVoting.tla")
        assert (r.spec, r.params) == ("paxos", [1, 3, 2, 2, 1, 3, 1, 3, 3, 5, 6])
        r = amd.ResolvedSpec(ref / "MCPaxos.tla")             # MCPaxos.tla:7-9 as committed
        assert (r.spec, r.params) == ("paxos", [0, 1, 1, 2, 15, 3, 1, 1, 1])
        monkeypatch.setenv("TLA_PATH", str(ref))
        assert amd.ResolvedSpec(P / "MCPaxos3.tla").spec == "paxos"
        monkeypatch.delenv("TLA_PATH")
        for f in ("Paxos.tla", "Voting.tla"):
            shutil.copy(ref / f, d / f)
        (d / "MCPaxos3.cfg").write_text((P / "MCPaxos3.cfg").read_text())
        assert amd.ResolvedSpec(d / "MCPaxos3.tla").spec == "paxos"
        (d / "Voting.tla").write_text((ref / "Voting.tla").read_text().replace("maxBal[a] \\leq b", "maxBal[a] < b"))
        with pytest.raises(amd.McError) as e:
            amd.ResolvedSpec(d / "MCPaxos3.tla")
        assert e.value.code == -9 and "Voting.tla differs" in str(e.value)


def test_mutated_paxos_models_are_refused_or_resolved_never_crash(amd, tmp_path):
    """the sizes of a Paxos model are parsed out of module text by hand-written C++: 300 randomly damaged copies of the model and
    its cfg must come back as an error code or a descriptor within the lowering's limits"""
    import random
    P = ROOT / "specs" / "paxos"
    tla, cfg = (P / "MCPaxos3.tla").read_text(), (P / "MCPaxos3.cfg").read_text()
    rnd = random.Random(7)
    outcomes = {"resolved": 0, "refused": 0}
    for _ in range(300):
        t, c = tla, cfg
        for _ in range(rnd.randint(1, 3)):
            in_tla = rnd.random() < 0.6
            src = t if in_tla else c
            a = rnd.randrange(len(src))
            b = min(len(src), a + rnd.randint(1, 12))
            how = rnd.choice(["del", "dup", "chr"])
            src = src[:a] + src[b:] if how == "del" else src[:a] + src[a:b] * 2 + src[b:] if how == "dup" else \
                src[:a] + rnd.choice("{}(),=<>\\!|.x1 ") + src[a + 1:]
            if in_tla:
                t = src
            else:
                c = src
        (tmp_path / "MCPaxos3.tla").write_text(t)
        (tmp_path / "MCPaxos3.cfg").write_text(c)
        try:
            r = amd.ResolvedSpec(tmp_path / "MCPaxos3.tla", unverified=True)
            assert r.spec == "paxos" and 1 <= r.params[1] <= 4 and 1 <= r.params[2] <= 3 and 1 <= r.params[3] <= 4
            outcomes["resolved"] += 1
        except amd.McError:
            outcomes["refused"] += 1
    assert outcomes["refused"] > 100 and outcomes["resolved"] > 20
