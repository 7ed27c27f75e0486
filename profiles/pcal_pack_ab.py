#!/usr/bin/env python
"""profiles/pcal_pack_ab.py — generated PlusCal code with rows PACKED to the cells' inferred ranges (round 6, pcal_codegen.cpp "cell ranges")
against the interpreter's rows ($TLAMC_JIT_PACK=0), same device, same call: ms per complete search, stored bytes per state, counts equal.
(`expand_ms` = the expand kernel's HIP-event time of the LAST search; the files of calls zt .. zz carry a quarter of it: the engine resets its kernel
statistics per run and this script divided by its four runs.)"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import tla_rust_amd as amd  # noqa: E402

MSQ = "INVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n"
JOBS = [("pagecache N=3", "pagecache.tla", "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n", dict(table_capacity=1 << 27, arena_capacity=22 << 20)),
        ("ms_queue_counted N=3 K=3", "ms_queue_counted.tla", "CONSTANTS N = 3 K = 3 Counted = TRUE\n" + MSQ, dict(table_capacity=1 << 28, arena_capacity=40 << 20)),
        ("ms_queue_counted N=3 K=4", "ms_queue_counted.tla", "CONSTANTS N = 3 K = 4 Counted = TRUE\n" + MSQ, dict(table_capacity=1 << 30, arena_capacity=140 << 20)),
        ("two_phase_channels RM=4", "two_phase_channels.tla", "CONSTANTS RM = 4 Eager = FALSE\nINVARIANTS Consistent\n", dict(table_capacity=1 << 27, arena_capacity=16 << 20)),
        ("radix_tree N=4", "radix_tree.tla", "CONSTANTS N = 4 Plain = FALSE\n", dict(table_capacity=1 << 26, arena_capacity=8 << 20)),
        ("epoch_gc N=3", "epoch_gc.tla", "CONSTANTS N = 3 Grace = 2\n", dict(table_capacity=1 << 25, arena_capacity=4 << 20))]
ONLY = os.environ.get("PACK_AB_ONLY")          # "1": the packed form only (shape / knob sweeps through $TLAMC_JIT_DEFS)
NJOBS = int(os.environ.get("PACK_AB_JOBS", "99"))
for name, f, cfg, kw in JOBS[:NJOBS]:
    got = {}
    for pack in ((ONLY,) if ONLY else ("1", "0")):
        os.environ["TLAMC_JIT_PACK"] = pack
        try:
            prog = amd.Program((ROOT / "specs" / "pluscal" / f).read_text(), cfg)
            t0 = time.perf_counter()
            eng = amd.Engine("pcal", prog.params, trace=False, timing=True, jit=True, chunk_states=1 << 21, **kw)
            build = time.perf_counter() - t0
            eng.run()
            t0 = time.perf_counter()
            for _ in range(3):
                r = eng.run()
            dt = (time.perf_counter() - t0) / 3
            ks = eng.kernel_stats()
            got[pack] = (r.distinct, r.generated, r.depth, r.verdict)
            print(json.dumps({"model": name, "packed": pack == "1", "ms": round(1e3 * dt, 3), "states_per_s_G": round(r.distinct / dt / 1e9, 3), "state_bytes": ks["state_bytes"],
                              "state_bytes_interpreter": amd.state_bytes("pcal", prog.params), "distinct": r.distinct, "generated": r.generated, "depth": r.depth,
                              "verdict": r.verdict, "defs": os.environ.get("TLAMC_JIT_DEFS", ""), "expand_ms": round(ks["expand"]["ms_total"], 3), "engine_create_s": round(build, 1)}), flush=True)
            eng.close()
            prog.close()
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"model": name, "packed": pack == "1", "error": str(e)[:300]}), flush=True)
    if len(got) == 2 and got["1"] != got["0"]:
        print(json.dumps({"model": name, "MISMATCH": [got["1"], got["0"]]}), flush=True)
        sys.exit(1)
