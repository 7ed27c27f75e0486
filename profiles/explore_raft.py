"""Explore complete-graph sizes of the 3-server raft model on the GPU (no oracle): python profiles/explore_raft.py MaxTerm K [K ...]
Prints one JSON line per model; the counts are NOT golden until the exact-dedup oracle has reproduced them (tests/golden/make_golden.py)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

T = int(sys.argv[1])
for K in [int(x) for x in sys.argv[2:]]:
    params = [3, 4, T, 3, 1, 1, K, 2, 6, K]
    try:
        eng = amd.Engine("raft", params, table_capacity=5 << 28, arena_capacity=550_000_000, chunk_states=1 << 22, trace=False, timing=False)
        t0 = time.perf_counter()
        r = eng.run()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(params=params, distinct=r.distinct, generated=r.generated, depth=r.depth, verdict=r.verdict, seconds=round(dt, 3),
                              levels=r.levels)), flush=True)
        eng.close()
    except Exception as e:  # noqa: BLE001
        print(json.dumps(dict(params=params, error=str(e))), flush=True)
