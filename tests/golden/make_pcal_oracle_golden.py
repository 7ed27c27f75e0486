"""tests/golden/pcal_oracle.json: ORACLE-MADE goldens of the larger PlusCal models — oracle/tlaplus.py, the
TLC-like evaluator under oracle/ (test infrastructure; no code shared with the product), on the HAND-WRITTEN pcal2tla-style translations
under tests/golden/pcal_records/ (written from the algorithm text of specs/pluscal/*.tla, not from the product's translation), at the sizes
tests/golden/pcal_channels.json holds PRODUCT-made numbers for.  tests/test_pcal.py asserts that the two files agree entry by entry, so
every test that compares the compiled program (host VM, GPU interpreter, generated code) with pcal_channels.json is pinned to the oracle:
another text, another evaluator.  One worker process per model; the largest (radix tree, four inserters: 3.4 M states) takes the evaluator
about ten minutes and a few GB.  pagecache N = 3 (20 M states) stays product-made: out of this evaluator's reach (its N = 2 is evaluated
live in tests/test_pcal.py).

    python tests/golden/make_pcal_oracle_golden.py [case ...]"""
import json
import multiprocessing as mp
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
F = ROOT / "tests" / "golden" / "pcal_records"
OUT = ROOT / "tests" / "golden" / "pcal_oracle.json"

CASES = {
    # case (= the key in pcal_channels.json): fixture module, cfg text
    "two_phase_channels_rm4": ("TwoPhaseChannels", "SPECIFICATION Spec\nCONSTANT RM = 4\nCONSTANT Eager = FALSE\n"
                               "INVARIANT Consistent CommitNeedsAllVotes InboxHoldsVotes FromTheCoordinator AtMostTwoWaiting\n"),
    "two_phase_soup_rm6": ("TwoPhaseSoup", "SPECIFICATION Spec\nCONSTANT RM = 6\nCONSTANT Hasty = FALSE\n"
                           "INVARIANT Consistent OneDecision PreparedWereSent KnownMessages SoupIsSmall\n"),
    "two_phase_soup_rm7": ("TwoPhaseSoup", "SPECIFICATION Spec\nCONSTANT RM = 7\nCONSTANT Hasty = FALSE\n"
                           "INVARIANT Consistent OneDecision PreparedWereSent KnownMessages SoupIsSmall\n"),
    "epoch_gc_n3": ("EpochGc", "SPECIFICATION Spec\nCONSTANT N = 3\nCONSTANT Grace = 2\nINVARIANT HeadIsLive NoDanglingReader EpochInRange\n"),
    "io_buffer_n4": ("IoBuffer", "SPECIFICATION Spec\nCONSTANT N = 4\nCONSTANT Cap = 2\nCONSTANT Patient = TRUE\nINVARIANT HeaderInRange SealedIsFull FlushedFull\n"),
    "radix_tree_n4": ("RadixTree", "SPECIFICATION Spec\nCONSTANT N = 4\nCONSTANT Plain = FALSE\nINVARIANT InsertedKeysAreFound NoLeak ChildrenAreNodes\n"),
}


def work(case):
    sys.path.insert(0, str(ROOT / "oracle"))
    import tlaplus as T
    fixture, cfg = CASES[case]
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(cfg)
    t0 = time.time()
    p = T.Checker(F / f"{fixture}.tla", cfg_path=f.name, search=[]).run_levels(keep_states=False)
    assert p["verdict"] == "ok", (case, p["verdict"], p.get("error"))
    return case, dict(distinct=p["distinct"], generated=p["generated"], depth=p["depth"], levels=p["levels"], seconds=round(time.time() - t0, 1),
                      source=f"ORACLE-MADE: oracle/tlaplus.py on the HAND-WRITTEN pcal2tla-style translation tests/golden/pcal_records/{fixture}.tla "
                             f"(tests/golden/make_pcal_oracle_golden.py); cfg: {cfg.strip()!r}")


if __name__ == "__main__":
    todo = sys.argv[1:] or list(CASES)
    out = json.loads(OUT.read_text()) if OUT.exists() else {}
    with mp.Pool(min(len(todo), 6)) as pool:
        for case, entry in pool.imap_unordered(work, todo):
            out[case] = entry
            print(case, entry["distinct"], entry["generated"], entry["depth"], entry["seconds"], flush=True)
            OUT.write_text(json.dumps({k: out[k] for k in sorted(out)}, indent=1) + "\n")
