"""GPU leg of tests/test_codegen.py: compiled PlusCal programs as GENERATED code on the device (MC_F_JIT: tla_rust_amd/csrc/pcal_codegen.cpp
writes the program as C++ for spec_gen.h, hipcc builds it into an engine library when the engine is created) against the bytecode
interpreter on the same device — counters, verdict, depth, per-level counts, the per-level SETS of states and the counterexample of a violating
model (rows LEAVE both back-ends in the interpreter's layout; the generated code STORES them packed to its cells' inferred ranges, so the
comparison also proves pack + export on every state of these graphs) — and, through tests/test_gpu_pcal.py's chain, against oracle/tla_eval.py.  (Sorts behind the other GPU files: the
first engine of each program pays ~20 s of compilation.)"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_pcal import cfg_text  # noqa: E402
from test_pcal import CASES  # noqa: E402

pytestmark = pytest.mark.gpu
# (pagecache, epoch_gc, io_buffer, ms_queue_counted: 16 and more slots per state — the slot-SLICED launches of the slot-by-slot kernel must not be
#  mixed with the by-pairs kernel: round 6's first device run reported a deadlock for every parent there)
PICK = ["pcal_intro", "atomic_add", "peterson", "ticket_lock", "treiber_stack", "ms_queue", "bounded_queue", "mailboxes", "recursive_sum", "radix_tree",
        "pagecache", "epoch_gc", "io_buffer", "ms_queue_counted", "two_phase_channels"]
JIT_CASES = []
for stem in PICK:
    c = next((c for c in CASES if c[0].stem == stem), None)
    if c is not None:
        JIT_CASES.append(c)


@pytest.fixture(scope="module")
def amd():
    import tla_rust_amd
    assert tla_rust_amd.device_count() >= 1, "no HIP device visible"
    return tla_rust_amd


def level_sets(eng, r):
    out, first = [], 0
    for n in r["levels"]:
        out.append(sorted(eng.read_states(first, n)))
        first += n
    return out


@pytest.mark.parametrize("path,invs,consts", JIT_CASES, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_generated_code_on_gpu_equals_the_interpreter(amd, path, invs, consts, capfd):
    prog = amd.Program(path.read_text(), cfg_text(invs, consts))
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    a = amd.Engine("pcal", prog.params, jit=True, **kw)                       # generated code, the by-pairs kernel (pairs sorted by label)
    assert "interpreting the program" not in capfd.readouterr().err, "MC_F_JIT fell back to the interpreter"
    c = amd.Engine("pcal", prog.params, jit=True, debug_flags=32, **kw)       # generated code, slot by slot (MC_F_NOFAMILY)
    b = amd.Engine("pcal", prog.params, **kw)                                 # the interpreter
    ra, rc, rb = a.run(), c.run(), b.run()
    for r in (ra, rc):
        for k in ("distinct", "generated", "queue_left", "depth", "levels", "verdict"):
            assert getattr(r, k) == getattr(rb, k), (k, getattr(r, k), getattr(rb, k))
    if ra.verdict == "ok" and ra.distinct <= 60000:
        want = level_sets(b, rb)
        assert level_sets(a, ra) == want and level_sets(c, rc) == want
    if ra.verdict not in ("ok", "budget"):
        tb = b.trace()
        assert tb and a.trace() == tb and c.trace() == tb   # the same counterexample, state texts and action names
    a.close()
    b.close()
    c.close()
    prog.close()


def test_a_program_the_translator_refuses_is_interpreted(amd, capfd):
    """sets of records (the message soup) are not translated: MC_F_JIT says so on stderr and the interpreter runs — same counts as without the flag"""
    c = next(c for c in CASES if c[0].stem == "two_phase_soup")
    prog = amd.Program(c[0].read_text(), cfg_text(c[1], c[2]))
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    a = amd.Engine("pcal", prog.params, jit=True, **kw)
    assert "sets of records" in capfd.readouterr().err
    b = amd.Engine("pcal", prog.params, **kw)
    ra, rb = a.run(), b.run()
    assert (ra.distinct, ra.generated, ra.depth, ra.verdict) == (rb.distinct, rb.generated, rb.depth, rb.verdict)
    a.close(); b.close(); prog.close()


def test_mc_moves_a_long_search_to_generated_code_on_its_own(amd, tmp_path, capfd, monkeypatch):
    """`mc X.tla` on a compiled PlusCal program starts on the device interpreter and, once the search has lasted $TLAMC_AUTOJIT_AFTER
    seconds, builds the generated code beside it; when the library is there first, the run starts over with it (frontend.cpp).  Here the
    library is in the cache already (a -jit run built it) and the wait is 0 s, so the move happens at the first report: same report text
    and counters as the interpreter alone ($TLAMC_AUTOJIT=0) and as the golden."""
    import json
    import shutil
    G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
    tla = tmp_path / "pagecache.tla"
    shutil.copy(ROOT / "specs" / "pluscal" / "pagecache.tla", tla)
    cfg = tmp_path / "pagecache.cfg"
    cfg.write_text("CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n")
    kw = dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21)
    monkeypatch.setenv("TLAMC_JIT_CACHE", str(tmp_path / "cache"))
    monkeypatch.setenv("TLAMC_JIT", "1")                      # generated code from the first state: builds the library
    r0, rep0 = amd.check_files(tla, cfg, **kw)
    assert "interpreting the program" not in capfd.readouterr().err
    monkeypatch.delenv("TLAMC_JIT")
    monkeypatch.setenv("TLAMC_AUTOJIT_AFTER", "0")
    r1, rep1 = amd.check_files(tla, cfg, **kw)
    err = capfd.readouterr().err
    assert "started over with it" in err, err
    monkeypatch.setenv("TLAMC_AUTOJIT", "0")
    r2, rep2 = amd.check_files(tla, cfg, **kw)
    assert "started over" not in capfd.readouterr().err
    for r in (r0, r1, r2):
        assert (r.verdict, r.distinct, r.generated, r.depth) == ("ok", G["distinct"], G["generated"], G["depth"])
    assert r0.levels == r1.levels == r2.levels and rep0 == rep1 == rep2


def test_generated_code_stores_packed_rows_and_hands_out_the_interpreters(amd, tmp_path, capfd):
    """Round 6: the generated code packs every cell into the bits of its inferred range (pcal_codegen.cpp "cell ranges"): the engine's
    state_bytes shrink (two_phase_channels RM = 3: 45 words -> 6), mc_engine_read_states / traces still hand out mc_state_bytes(spec) per
    state; $TLAMC_JIT_PACK=0 keeps the interpreter's rows (same counts either way); a checkpoint of packed rows is refused by the interpreter's
    engine and continued by another engine of the same generated code; the sharded entry points refuse a packed engine."""
    import os
    c = next(c for c in CASES if c[0].stem == "two_phase_channels" and c[2].get("RM") == 3 and not c[2].get("Eager"))
    prog = amd.Program(c[0].read_text(), cfg_text(c[1], c[2]))
    kw = dict(table_capacity=1 << 20, arena_capacity=1 << 18, chunk_states=1 << 12)
    b = amd.Engine("pcal", prog.params, timing=True, **kw)
    rb = b.run()
    public = amd.state_bytes("pcal", prog.params)
    assert b.kernel_stats()["state_bytes"] == public
    a = amd.Engine("pcal", prog.params, jit=True, timing=True, **kw)
    ra = a.run()
    assert "interpreting the program" not in capfd.readouterr().err
    stored = a.kernel_stats()["state_bytes"]
    assert stored * 3 <= public, (stored, public)
    assert (ra.distinct, ra.generated, ra.depth, ra.levels) == (rb.distinct, rb.generated, rb.depth, rb.levels)
    assert sorted(a.read_states(0, ra.distinct)) == sorted(b.read_states(0, rb.distinct))
    with pytest.raises(RuntimeError, match="ONE GPU"):
        a.shard_begin()
    # checkpoint at a budget, continue in a second engine of the same code; the interpreter's engine refuses the file
    a2 = amd.Engine("pcal", prog.params, jit=True, max_levels=6, **kw)
    r2 = a2.run()
    assert r2.verdict == "budget"
    ck = tmp_path / "packed.ck"
    a2.checkpoint(ck)
    a3 = amd.Engine("pcal", prog.params, jit=True, **kw)
    a3.restore(ck)
    r3 = a3.run()
    assert (r3.distinct, r3.generated, r3.depth, r3.verdict) == (rb.distinct, rb.generated, rb.depth, rb.verdict)
    b2 = amd.Engine("pcal", prog.params, **kw)
    with pytest.raises(RuntimeError, match="another"):
        b2.restore(ck)
    for e in (a, a2, a3, b, b2):
        e.close()
    os.environ["TLAMC_JIT_PACK"] = "0"
    try:
        d = amd.Engine("pcal", prog.params, jit=True, timing=True, **kw)
        rd = d.run()
        assert d.kernel_stats()["state_bytes"] == public
        assert (rd.distinct, rd.generated, rd.depth, rd.levels) == (rb.distinct, rb.generated, rb.depth, rb.levels)
        d.close()
    finally:
        del os.environ["TLAMC_JIT_PACK"]
    prog.close()


def test_mc_recovers_a_checkpoint_of_generated_code(amd, tmp_path):
    """`mc -checkpoint` of a run on generated code (here -jit; a long run moves there by itself) writes packed rows; `mc -recover` starts on the
    interpreter's engine, which refuses the file, and continues with the generated code instead: the golden's counts at the end."""
    import json
    import subprocess
    G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
    mc = ROOT / "tla_rust_amd" / "_build" / "mc"
    cfg = tmp_path / "pagecache.cfg"
    cfg.write_text("CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n")
    ck = tmp_path / "pc.ck"
    env = dict(os.environ, TLAMC_JIT_CACHE=str(tmp_path / "cache"))
    common = [str(mc), str(ROOT / "specs" / "pluscal" / "pagecache.tla"), "-config", str(cfg), "-tablelog2", "27", "-arena", str(22 << 20), "-noprogress"]
    a = subprocess.run(common + ["-jit", "-maxlevels", "20", "-checkpoint", str(ck)], env=env, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and ck.exists(), a.stdout + a.stderr
    b = subprocess.run(common + ["-recover", str(ck)], env=env, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout + b.stderr
    assert f"{G['generated']} states generated, {G['distinct']} distinct states found, 0 states left on queue." in b.stdout, b.stdout
