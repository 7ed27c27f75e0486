#!/usr/bin/env python
"""profiles/pcal_level_overhead.py — what a BFS level of the deep, narrow PlusCal graphs costs outside the expand kernel: the same generated-code engine
with and without per-kernel HIP events (MC_F_TIMING), and with the batched levels' blind grid at 2^18 .. 2^21 states ($TLAMC_BLIND_LOG2, read once per process:
this script re-executes itself per value)."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
MSQ = "INVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n"
JOBS = [("pagecache N=3", "pagecache.tla", "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n", dict(table_capacity=1 << 27, arena_capacity=22 << 20)),
        ("ms_queue_counted N=3 K=3", "ms_queue_counted.tla", "CONSTANTS N = 3 K = 3 Counted = TRUE\n" + MSQ, dict(table_capacity=1 << 28, arena_capacity=40 << 20))]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import tla_rust_amd as amd
    for name, f, cfg, kw in JOBS:
        for timing in (True, False):
            prog = amd.Program((ROOT / "specs" / "pluscal" / f).read_text(), cfg)
            eng = amd.Engine("pcal", prog.params, trace=False, timing=timing, jit=True, chunk_states=1 << 21, **kw)
            eng.run()
            t0 = time.perf_counter()
            for _ in range(5):
                r = eng.run()
            dt = (time.perf_counter() - t0) / 5
            print(json.dumps({"model": name, "blind_log2": os.environ.get("TLAMC_BLIND_LOG2", "20"), "hip_events": timing, "ms": round(1e3 * dt, 3), "levels": len(r.levels),
                              "us_per_level": round(1e6 * dt / len(r.levels), 1), "distinct": r.distinct, "states_per_s_G": round(r.distinct / dt / 1e9, 3)}), flush=True)
            eng.close()
            prog.close()
else:
    for b in ("20", "18", "19", "21"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, TLAMC_BLIND_LOG2=b), check=False)
