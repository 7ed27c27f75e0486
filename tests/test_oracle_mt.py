"""The multi-threaded oracle (oracle/bfs_mt.c — also bench.py's cpu_baseline on all host cores) against the
single-threaded one: identical counters, depth, per-level counts, verdict and violated invariant.  CPU only."""
import pytest

CASES = [
    ("atomic_add", [10], {}),
    ("pcal_intro", [0, 1, 20, 2], {}),
    ("pcal_intro", [1, 1, 20, 2], {}),                    # MoneyInvariant violated
    ("pcal_intro", [1, 0, 20, 2], {}),                    # Assert fails
    ("raft", [2, 2, 2, 9, 1, 1], {}),
    ("raft", [2, 3, 2, 9, 1, 3], {}),                     # CommittedLogStable violated
    ("raft", [3, 4, 2, 3, 1, 1, 0, 5], {}),               # 3 servers, MaxMsgKeys = 5: complete
    ("raft", [3, 4, 2, 3, 1, 1], dict(max_distinct=200_000)),
    ("raft", [3, 4, 2, 3, 1, 1], dict(max_levels=12)),
    ("ssi", [2, 2, 127, 0], {}),
    ("ssi", [3, 1, 127, 0], {}),
    ("ssi", [2, 2, 127, 3], {}),                          # an "expected to be violated" predicate
]


@pytest.mark.parametrize("spec,params,kw", CASES)
@pytest.mark.parametrize("threads", [1, 4])
def test_mt_oracle_equals_single_threaded(oracle, spec, params, kw, threads):
    a = oracle.oracle_run(spec, params, **kw)
    b = oracle.oracle_run_mt(spec, params, threads, **kw)
    for k in ("distinct", "generated", "queue_left", "depth", "verdict", "levels", "max_stat"):
        assert a[k] == b[k], (k, a[k], b[k])
    if a["verdict"] == "invariant":
        assert a["violated_invariant"] == b["violated_invariant"]
