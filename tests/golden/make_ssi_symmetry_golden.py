"""Generates tests/golden/ssi_symmetry.json: per-level distinct / generated counts of the SSI model under
cfg SYMMETRY (serializableSnapshotIsolation.tla:38-44), from the CPU oracle's BRUTE-FORCE canonicalisation
(oracle/spec_ssi.c s_canonical: all |TxnId|! x |Key|! permutations per successor).  Runs a few minutes.
Usage: python tests/golden/make_ssi_symmetry_golden.py"""
import json
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
import helpers as h  # noqa: E402

h.build_oracle()
CASES = [("2x3", [2, 3, 127, 0, 0, 3], 0), ("3x2", [3, 2, 127, 0, 0, 3], 0), ("4x3_prefix", [4, 3, 127, 0, 0, 3], 1_000_000),
         ("textbook_3x2_cahill", [3, 2, 32, 0, 1, 3], 0)]
out = {}
for name, params, maxd in CASES:
    o = h.oracle_run("ssi", params, max_distinct=maxd)
    out[name] = {"params": params, "max_distinct": maxd, "distinct": o["distinct"], "generated": o["generated"], "depth": o["depth"],
                 "verdict": o["verdict"], "violated_invariant": o["violated_invariant"], "trace_len": len(o["trace"]), "levels": o["levels"]}
    print(name, out[name], flush=True)
(HERE / "ssi_symmetry.json").write_text(json.dumps(out, indent=1) + "\n")
