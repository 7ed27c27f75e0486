"""world-1 run of the sharded level loop (mc_shard_run over RCCL with one rank) next to the fused engine, with the engine's own kernel timers:
which kernel does the sharded path spend its time in?  python profiles/sharded_w1_probe.py raft5|t3 [chunk_log2]"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
from tla_rust_amd.binding import Comm

which = sys.argv[1] if len(sys.argv) > 1 else "raft5"
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 21
if which == "raft5":
    params, table, arena, ml = [5, 6, 2, 5, 1, 1, 18, 1, 4], 3 << 29, 1_300_000_000, 18
else:
    params, table, arena, ml = [3, 4, 3, 3, 1, 1, 8, 2, 4, 8], 40 << 26, 527_000_000, 0
comm = Comm(Comm.unique_id(), 0, 1, 0)
eng = amd.Engine("raft", params, table_capacity=table, arena_capacity=arena, chunk_states=1 << ch, trace=False, shard_rank=0, shard_count=1, timing=True)
for k in range(2):
    t = time.perf_counter()
    s, st = comm.shard_run(eng, chunk_states=1 << ch, max_levels=ml)
    dt = time.perf_counter() - t
    ks = eng.kernel_stats()
    print(json.dumps(dict(which=which, run=k, seconds=round(dt, 3), distinct=s.distinct, verdict=s.verdict, stats=st,
                          kernel_ms={a: round(ks[a]["ms_total"], 1) for a in ("expand", "insert", "materialise")},
                          launches={a: ks[a]["launches"] for a in ("expand", "insert", "materialise")})), flush=True)
eng.close()
comm.close()
