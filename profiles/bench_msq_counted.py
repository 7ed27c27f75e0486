"""The compiled-PlusCal path on a model worth a GPU: specs/pluscal/ms_queue_counted.tla (the Michael-Scott queue with counted pointers
as nested records, nodes freed and reused) with three threads.  Expected counts = the SAME compiled program on the host build of the
interpreter (tests/_shim: 240 s for K = 3, 885 s for K = 4, one core).  Run on the GPU box: python profiles/bench_msq_counted.py"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import tla_rust_amd as amd

HOST = {3: (35263910, 99861367, 105, 240.0), 4: (124636796, 353102053, 105, 884.9)}
src = (ROOT / "specs" / "pluscal" / "ms_queue_counted.tla").read_text()
for k in (3, 4):
    prog = amd.Program(src, f"CONSTANTS N = 3 K = {k} Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n")
    best, r = 1e9, None
    for _ in range(2):
        eng = amd.Engine("pcal", prog.params, table_capacity=1 << (28 if k == 3 else 30), arena_capacity=(40 if k == 3 else 130) << 20,
                         chunk_states=1 << 21, trace=False)
        t0 = time.perf_counter()
        r = eng.run()
        best = min(best, time.perf_counter() - t0)
        eng.close()
    d, g, depth, host_s = HOST[k]
    print(json.dumps(dict(workload=f"ms_queue_counted N=3 K={k}", distinct=r.distinct, generated=r.generated, depth=r.depth, verdict=r.verdict,
                          equals_host_vm=(r.distinct, r.generated, r.depth) == (d, g, depth), seconds=round(best, 3),
                          Mstates_s=round(r.distinct / best / 1e6, 1), host_vm_one_core_s=host_s, speedup_vs_host_vm=round(host_s / best, 1),
                          state_bytes=amd.state_bytes("pcal", prog.params))), flush=True)
    prog.close()
