"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle.

  readme_pcal_intro_trace.json : the six states of the README's TLC counterexample,
                                 transcribed BY HAND from reference README.md:272-311 (not from
                                 the oracle) — this script only re-checks it.
  raft_levels.json             : per-level distinct counts of raft configurations, incl. the
                                 bench workload's budgeted prefix (takes a few minutes).
Usage: python tests/golden/make_golden.py [--big]
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import helpers  # noqa: E402

HERE = Path(__file__).resolve().parent

CASES = [
    ("raft2_mcr1_t2_m1", [2, 1, 2, 9, 1, 1], 0),
    ("raft2_mcr2_t2_m1", [2, 2, 2, 9, 1, 1], 0),
    ("raft2_mcr2_t2_m2", [2, 2, 2, 9, 2, 1], 0),
    ("raft2_mcr3_t2_m1", [2, 3, 2, 9, 1, 3], 0),
    ("raft3_mcr2_t2_m1_prefix", [3, 2, 2, 9, 1, 1], 300000),
    ("raft3_mcr4_t2_m1_prefix_small", [3, 4, 2, 3, 1, 1], 1000000),
    ("raft5_mcr6_t2_m1_prefix", [5, 6, 2, 5, 1, 1], 2000000),       # BASELINE config 4's model (5 servers, log <= 5), small prefix
    ("raft3_mcr4_t3_m2_prefix", [3, 4, 3, 3, 2, 3], 2000000),       # more terms, two messages in flight, both invariants
    # COMPLETE 3-server graphs: StateConstraint's fourth conjunct Cardinality(DOMAIN messages) <= MaxMsgKeys (specs/MCraft.tla)
    # makes the graph finite; params = {n, MCR, MaxTerm, MaxLogLen, MaxMsgs, invMask, naive, MaxMsgKeys}
    ("raft3_mcr4_t2_m1_k5_complete", [3, 4, 2, 3, 1, 1, 0, 5], 0),
    ("raft3_mcr4_t2_m1_k7_complete", [3, 4, 2, 3, 1, 1, 0, 7], 0),
    ("raft3_mcr4_t3_m2_k5_complete", [3, 4, 3, 3, 2, 3, 0, 5], 0),
]
BIG = [
    ("raft2_mcr1_t3_m1", [2, 1, 3, 9, 1, 1], 0),
    ("raft3_mcr4_t2_m1_bench", [3, 4, 2, 3, 1, 1], 25000000),   # round-1 bench.py workload (budgeted prefix)
    # bench.py workload (BASELINE config 3): specs/MCraft.cfg, COMPLETE graph, 102 586 254 states (about 3 minutes on 8 threads,
    # 17 GB; run with the multi-threaded oracle, whose counts equal the single-threaded one's: tests/test_oracle_mt.py)
    ("raft3_mcr4_t2_m1_k10_complete", [3, 4, 2, 3, 1, 1, 0, 10], 0),
]
# raft3_mcr4_t2_m1_k11_complete (MaxMsgKeys = 11: 336 581 097 states, a 55.5 GB oracle arena) is NOT regenerated here: it was
# made by the same oracle binary on the GPU box's host — `oracle_mc raft 3 4 2 3 1 1 0 11 --threads 192 --levels-out`, 248 s,
# profiles/r02zf_cmd.sh / profiles/r02zf_oracle_k11.txt — and its entry (with a `source` field) is kept as it is.

if __name__ == "__main__":
    cases = CASES + (BIG if "--big" in sys.argv else [])
    path = HERE / "raft_levels.json"
    old = {c["name"]: c for c in json.loads(path.read_text())["cases"]} if path.exists() else {}
    out = []
    for name, params, maxd in cases:
        if name.endswith("k10_complete"):
            import os
            r = helpers.oracle_run_mt("raft", params, os.cpu_count() or 1, max_distinct=maxd)
        else:
            r = helpers.oracle_run("raft", params, max_distinct=maxd)
        print(name, r["distinct"], r["generated"], r["depth"], r["verdict"], f"{r['seconds']:.1f}s", flush=True)
        old[name] = dict(name=name, params=params, max_distinct=maxd, distinct=r["distinct"], generated=r["generated"],
                         depth=r["depth"], verdict=r["verdict"], levels=r["levels"], max_stat=r["max_stat"][:5])
    path.write_text(json.dumps(dict(cases=list(old.values())), indent=1) + "\n")
