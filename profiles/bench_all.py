"""Throughput of every lowered spec on one MI355X (not the contract bench: see bench.py).
Prints one JSON line per workload: distinct states/s, generated/s, kernel times, algorithmic GB/s."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

WORKLOADS = [
    ("atomic_add N=24 (config 2 series)", "atomic_add", [24], dict(table_capacity=1 << 26, arena_capacity=(1 << 24) + 4096)),
    ("atomic_add N=28", "atomic_add", [28], dict(table_capacity=1 << 30, arena_capacity=(1 << 28) + 4096)),
    # SURVEY 8d config 2's series is N in {20, 24, 28, 30}: 2^30 + 1 states, an 8.6 GB arena, a 16 GB seen-set at load 0.5
    ("atomic_add N=20", "atomic_add", [20], dict(table_capacity=1 << 22, arena_capacity=(1 << 20) + 4096)),
    ("atomic_add N=30", "atomic_add", [30], dict(table_capacity=1 << 31, arena_capacity=(1 << 30) + 4096)),
    ("pcal_intro committed (config 1)", "pcal_intro", [0, 1, 20, 2], dict(table_capacity=1 << 16, arena_capacity=1 << 14)),
    ("raft 3 servers, 25M budget (round 1's bench prefix)", "raft", [3, 4, 2, 3, 1, 1, 16, 2, 8],
     dict(table_capacity=1 << 28, arena_capacity=30_000_000, max_distinct=25_000_000)),
    ("raft 3 servers complete, MaxMsgKeys=9 (29.7M)", "raft", [3, 4, 2, 3, 1, 1, 9, 0, 0, 9], dict(table_capacity=3 << 24, arena_capacity=31_000_000)),
    ("raft 5 servers, log <= 5, 20M budget (config 4's model on one GPU)", "raft", [5, 6, 2, 5, 1, 1],
     dict(table_capacity=3 << 26, arena_capacity=70_000_000, max_distinct=20_000_000)),
    # BASELINE config 4's model at the size the 8-GPU run is budgeted for, on ONE GPU: 18 levels = 924 041 864 states resident in HBM
    # (W = 192 B with 18 / 1 / 4 slots: the compact layout of round 3; 944 B per state in round 2 would have needed 872 GB)
    ("raft 5 servers, log <= 5 (config 4's model), 18 levels = 9.24e8 states on ONE GPU", "raft", [5, 6, 2, 5, 1, 1, 18, 1, 4],
     dict(table_capacity=3 << 29, arena_capacity=1_300_000_000, max_levels=18)),
    ("raft 2 servers MaxTerm=3 complete (4.3M)", "raft", [2, 1, 3, 9, 1, 1], dict(table_capacity=1 << 25, arena_capacity=5_000_000)),
    ("SSI 2x3 complete (7.9M), 7 invariants", "ssi", [2, 3, 127, 0], dict(table_capacity=1 << 26, arena_capacity=9_000_000)),
    ("SSI 4x3 levels 1-10 (config 5 prefix), 7 invariants", "ssi", [4, 3, 127, 0],
     dict(table_capacity=1 << 29, arena_capacity=200_000_000, max_levels=10)),
    # cfg SYMMETRY Perms (Key and TxnId symmetry sets, serializableSnapshotIsolation.tla:38-44): orbits instead of states
    ("SSI 3x2 complete under SYMMETRY (6.7M orbits of 80.8M states), 7 invariants", "ssi", [3, 2, 127, 0, 0, 3],
     dict(table_capacity=1 << 25, arena_capacity=8_000_000)),
    ("SSI 4x3 under SYMMETRY (config 5 as the run-book sets it up), 60M budget, 7 invariants", "ssi", [4, 3, 127, 0, 0, 3],
     dict(table_capacity=1 << 30, arena_capacity=400_000_000, max_distinct=60_000_000)),
]
WORKLOADS += [
    # the Paxos family (examples/Paxos): small graphs, launch-bound — listed for completeness, not as a throughput claim
    ("Paxos 3 acceptors x 2 values x ballots 0..2 (185 369 states), Inv1-4 + V!Spec", "paxos", [0, 3, 2, 3, 15, 0, 1],
     dict(table_capacity=1 << 20, arena_capacity=1 << 18, deadlock=False)),
    ("Paxos 3 x 2 x 0..2 under SYMMETRY (17 153 orbits)", "paxos", [0, 3, 2, 3, 15, 3, 1], dict(table_capacity=1 << 18, arena_capacity=1 << 16, deadlock=False)),
]
if len(sys.argv) > 1:   # substring filter
    WORKLOADS = [w for w in WORKLOADS if sys.argv[1] in w[0]]

for name, spec, params, kw in WORKLOADS:
    try:
        eng = amd.Engine(spec, params, chunk_states=1 << 20, trace=False, timing=True, **kw)
        if kw.get("arena_capacity", 0) < 1_000_000_000:   # (the 10^9-state run is timed cold: 1.2 s of it is enough)
            eng.run()
        t0 = time.perf_counter()
        r = eng.run()
        dt = time.perf_counter() - t0
        ks = eng.kernel_stats()
        W = ks["state_bytes"]
        kms = {k: round(ks[k]["ms_total"], 3) for k in ("expand", "insert", "materialise")}
        alg = 2 * W * r.distinct + 8 * r.generated
        print(json.dumps(dict(workload=name, distinct=r.distinct, generated=r.generated, depth=r.depth, verdict=r.verdict, levels=r.levels[-4:],
                              ms=round(dt * 1e3, 2), distinct_per_s=round(r.distinct / dt), generated_per_s=round(r.generated / dt),
                              W=W, kernel_ms=kms, alg_GBs=round(alg / dt / 1e9, 1))), flush=True)
        eng.close()
    except Exception as e:  # noqa: BLE001
        print(json.dumps(dict(workload=name, error=str(e))), flush=True)
