"""Exploration aid (round 2): which 3-server raft model has a COMPLETE graph of 1e8..1e9 states?
Runs candidate models on the GPU engine until verdict ok / budget and prints per-level counts.  The chosen model is then
verified by the CPU oracle (tests/golden/make_golden.py).  argv: JSON device parameter vectors."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

for a in sys.argv[1:]:
    params = json.loads(a)
    W = amd.state_bytes("raft", params)
    arena = min(int(225e9 // W), (1 << 32) - (1 << 22))
    t0 = time.time()
    eng = None
    try:
        eng = amd.Engine("raft", params, device=0, table_capacity=1 << 30, arena_capacity=arena, chunk_states=1 << 22,
                         max_distinct=int(arena * 0.8), trace=False, timing=False)
        r = eng.run()
        print(json.dumps(dict(params=params, W=W, arena=arena, distinct=r.distinct, generated=r.generated, depth=r.depth,
                              verdict=r.verdict, seconds=r.seconds, wall=time.time() - t0, levels=r.levels)), flush=True)
    except amd.McError as e:
        print(json.dumps(dict(params=params, W=W, arena=arena, error=str(e), wall=time.time() - t0)), flush=True)
    if eng is not None:
        eng.close()
