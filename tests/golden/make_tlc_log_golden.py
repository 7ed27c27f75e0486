"""Makes tests/golden/tlc_log_mcinnerserial.json: the ONE complete TLC run the reference tree holds a log of —
examples/SpecifyingSystems/AdvancedExamples/testout2 (TLC 1.57 on MCInnerSerial.tla + .cfg: "6181 states generated, 195
distinct states found", "The state graph has diameter 5", testout2:265-266; 22 hours in 2001, testout2:267) — reproduced by
oracle/tlaplus.py evaluating the reference's module texts where they lie.

The next-state relation quantifies over SUBSET (opId' \\X opId') (InnerSerial.tla): one state of level 4 or 5 takes the pure
Python evaluator one to several minutes, so the frontier of every level is expanded by a pool of forked workers (the BFS
bookkeeping — de-duplication, CONSTRAINT, INVARIANT, counters — stays in the parent and is exactly Checker.run_levels's).
Run in the build container (about an hour on 8 cores):   python tests/golden/make_tlc_log_golden.py [workers]
"""
import json
import multiprocessing as mp
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
D = Path("/root/reference/examples/SpecifyingSystems/AdvancedExamples")

CHECKER = None
FRONTIER = []


def expand(i):
    t0 = time.time()
    succ = list(CHECKER.spec.successors(FRONTIER[i], CHECKER.nextf))
    return i, succ, time.time() - t0


def main(workers):
    global CHECKER, FRONTIER
    import tlaplus as T
    c = CHECKER = T.Checker(D / "MCInnerSerial.tla", cfg_text=(D / "MCInnerSerial.cfg").read_text(),
                            search=[D, D.parent / "Standard", D.parent / "CachingMemory", D.parent / "TLC"])
    t0 = time.time()
    seen, levels, generated, frontier, verdict = set(), [], 0, [], "ok"
    for st in c.spec.init_states(c.init_name):
        generated += 1
        if st in seen:
            continue
        if c.violated(st) >= 0:
            verdict = "invariant"
        if c.in_model(st):
            seen.add(st)
            frontier.append(st)
    levels.append(len(frontier))
    progress = []
    while frontier and verdict == "ok":
        FRONTIER = frontier
        with mp.get_context("fork").Pool(workers) as pool:   # forked per level: the workers inherit this level's frontier
            results = {}
            for i, succ, dt in pool.imap_unordered(expand, range(len(frontier))):
                results[i] = succ
                print(f"  level {len(levels)}: state {i + 1}/{len(frontier)} has {len(succ)} successors ({dt:.0f} s)", file=sys.stderr, flush=True)
        new = []
        for i in range(len(frontier)):          # merged in frontier order: the counters are those of a sequential BFS
            if not results[i]:
                verdict = "deadlock"
            for s2 in results[i]:
                generated += 1
                if s2 in seen:
                    continue
                if c.violated(s2) >= 0:
                    verdict = "invariant"
                if c.in_model(s2):
                    seen.add(s2)
                    new.append(s2)
        frontier = new
        if new:
            levels.append(len(new))
        progress.append(dict(levels=len(levels), generated=generated, distinct=len(seen), seconds=round(time.time() - t0)))
        print(f"Progress({len(levels)}): {generated} states generated, {len(seen)} distinct states found", file=sys.stderr, flush=True)
    out = dict(module="examples/SpecifyingSystems/AdvancedExamples/MCInnerSerial.tla", generated=generated, distinct=len(seen),
               depth=len(levels), levels=levels, verdict=verdict, progress=progress, workers=workers, seconds=round(time.time() - t0))
    (ROOT / "tests" / "golden" / "tlc_log_mcinnerserial.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else mp.cpu_count())
