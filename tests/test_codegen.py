"""The GENERATED lowering of compiled PlusCal programs (tla_rust_amd/csrc/pcal_codegen.cpp -> spec_gen.h: what MC_F_JIT builds for the
device) against the bytecode interpreter (spec_vm.h), on the host: tests/_gen/harness.cpp runs a breadth-first search with the
interpreter and asks BOTH back-ends for the status, the fingerprint and the successor row of every (reachable state, slot) pair, and for
every initial state.  Every program of tests/test_pcal.py's CASES goes through it; the ones the translator refuses (sets of records) must
be refused with a message that says why — mc_engine_create then falls back to the interpreter."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import helpers  # noqa: E402
from test_pcal import CASES  # noqa: E402


@pytest.mark.parametrize("path,invs,consts", CASES, ids=lambda v: v.stem if isinstance(v, Path) else None)
def test_generated_code_equals_the_interpreter_state_by_state(path, invs, consts):
    prog = helpers.ShimProgram(path.read_text(), invs, consts)
    try:
        try:
            text = helpers.program_codegen(prog)
        except RuntimeError as e:
            assert "sets of records" in str(e), str(e)   # the one documented gap: such a program is interpreted
            return
        assert "struct GenProg" in text and "using SpecGen = SpecGenT<GenProg>;" in text
        r = helpers.gen_check(prog, max_states=20000)
        assert r["mismatches"] == 0, r
        assert r["states_checked"] > 0 and r["pairs_checked"] >= r["states_checked"] and r["distinct"] > 0
    finally:
        prog.close()


def test_the_two_root_specs_of_the_reference_translate_completely():
    """pcal_intro.tla and atomic_add.tla (the reference's own PlusCal modules) take the generated path under `mc -generic` + MC_F_JIT: the
    whole graph, every pair, and the counts of the README's TLC run (9097 generated / 6164 distinct would be the buggy variant; the fixed
    module's counts come from the interpreter here, pinned to the oracle elsewhere)"""
    for path, invs in ((ROOT / "specs" / "pcal_intro.tla", ["MoneyInvariant"]), (ROOT / "specs" / "atomic_add.tla", [])):
        prog = helpers.ShimProgram(path.read_text(), invs, {})
        try:
            r = helpers.gen_check(prog)
            s = helpers.shim_run("pcal", prog.params)
            assert r["mismatches"] == 0 and r["distinct"] == s["distinct"] and r["depth"] == s["depth"], (r, s)
        finally:
            prog.close()


def test_cells_are_packed_to_their_inferred_ranges(monkeypatch):
    """Round 6 ("variables bit-packed to their inferred ranges (a pc in ceil(log2 labels) bits, booleans in 1)"): the
    interval analysis of pcal_codegen.cpp gives the stored row of the generated code — two-phase commit over channels, RM = 3: 45 words of
    32-bit cells -> 6; the harness packs and exports EVERY reachable state (kind 9 = a cell outside its range) and compares every successor in
    both layouts.  $TLAMC_JIT_PACK=0 is the interpreter's layout, where rows and fingerprints are the interpreter's bit for bit."""
    c = next(c for c in CASES if c[0].stem == "two_phase_channels" and c[2].get("RM") == 3 and not c[2].get("Eager"))
    prog = helpers.ShimProgram(c[0].read_text(), c[1], c[2])
    try:
        text = helpers.program_codegen(prog)
        assert "PACKED = true" in text
        r = helpers.gen_check(prog, max_states=20000)
        assert r["mismatches"] == 0 and r["vm_words"] == 45 and r["stored_words"] <= 6, r
        monkeypatch.setenv("TLAMC_JIT_PACK", "0")
        text0 = helpers.program_codegen(prog)
        assert "PACKED = false" in text0
        r0 = helpers.gen_check(prog, max_states=20000)
        assert r0["mismatches"] == 0 and r0["stored_words"] == r0["vm_words"] == 45, r0
        assert (r0["distinct"], r0["generated"], r0["pairs_checked"]) == (r["distinct"], r["generated"], r["pairs_checked"])
    finally:
        prog.close()


def test_a_counter_keeps_its_32_bits_and_the_default_value_costs_one_code():
    """what the analysis must NOT narrow: `x := x + 1` under a test is not bounded by the test here (widened to 32 bits: euclid's x, y keep the
    interpreter's layout), and a variable declared without an initial value (defaultInitValue, a huge negative cell) is one extra code of
    its cell, not 32 bits (peterson: 3 words -> 1)"""
    import re
    for stem, want_packed in (("euclid", False), ("peterson", True)):
        c = next(c for c in CASES if c[0].stem == stem)
        prog = helpers.ShimProgram(c[0].read_text(), c[1], c[2])
        try:
            text = helpers.program_codegen(prog)
            assert ("PACKED = true" in text) == want_packed, stem
            m = re.search(r"NW = (\d+), VMW = (\d+)", text)
            assert (int(m.group(1)) < int(m.group(2))) == want_packed
        finally:
            prog.close()
