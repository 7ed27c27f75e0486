"""Throughput of the compiled-PlusCal path (spec_vm.h interpreter) next to the hand lowering of the same spec.
Run on the GPU box: python profiles/bench_pcal.py > gpurun_out/bench_pcal.jsonl"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import tla_rust_amd as amd


def timed(make, reps=3):
    best, res = 1e9, None
    for _ in range(reps):
        eng = make()
        t0 = time.perf_counter()
        res = eng.run()
        best = min(best, time.perf_counter() - t0)
        eng.close()
    return best, res


src = (ROOT / "specs" / "atomic_add_n.tla").read_text()
for n in (16, 20, 22):
    prog = amd.Program(src, f"CONSTANT N = {n}\n")
    kw = dict(table_capacity=1 << 25, arena_capacity=(1 << n) + 4096, chunk_states=1 << 18, trace=False)
    tv, rv = timed(lambda: amd.Engine("pcal", prog.params, **kw))
    th, rh = timed(lambda: amd.Engine("atomic_add", [n], **kw))
    assert (rv.distinct, rv.generated, rv.depth) == (rh.distinct, rh.generated, rh.depth) == ((1 << n) + 1, n * (1 << (n - 1)) + 3, n + 2)
    print(json.dumps(dict(workload=f"atomic_add_n N={n}", distinct=rv.distinct, generated=rv.generated,
                          compiled_ms=round(tv * 1e3, 2), compiled_Mstates_s=round(rv.distinct / tv / 1e6, 1),
                          hand_ms=round(th * 1e3, 2), hand_Mstates_s=round(rh.distinct / th / 1e6, 1),
                          state_bytes_compiled=amd.state_bytes("pcal", prog.params))), flush=True)
    prog.close()
src = (ROOT / "specs" / "pluscal" / "cas_counter.tla").read_text()
for w, n in ((3, 3), (4, 2), (4, 3)):
    prog = amd.Program(src, f"CONSTANTS Workers = {w} N = {n}\nINVARIANTS NeverTooMany SeenIsOld\n")
    t, r = timed(lambda: amd.Engine("pcal", prog.params, table_capacity=1 << 25, arena_capacity=1 << 23, chunk_states=1 << 17, trace=False))
    print(json.dumps(dict(workload=f"cas_counter Workers={w} N={n}", distinct=r.distinct, generated=r.generated, depth=r.depth,
                          verdict=r.verdict, ms=round(t * 1e3, 2), Mstates_s=round(r.distinct / t / 1e6, 1),
                          state_bytes=amd.state_bytes("pcal", prog.params))), flush=True)
    prog.close()
src = (ROOT / "specs" / "pluscal" / "treiber_stack.tla").read_text()
for n in (3, 4):
    prog = amd.Program(src, f"CONSTANT N = {n}\nINVARIANTS PoppedOnce TopIsNode Conservation\n")
    t, r = timed(lambda: amd.Engine("pcal", prog.params, table_capacity=1 << 25, arena_capacity=1 << 23, chunk_states=1 << 17, trace=False))
    print(json.dumps(dict(workload=f"treiber_stack N={n}", distinct=r.distinct, generated=r.generated, depth=r.depth,
                          verdict=r.verdict, ms=round(t * 1e3, 2), Mstates_s=round(r.distinct / t / 1e6, 1),
                          state_bytes=amd.state_bytes("pcal", prog.params))), flush=True)
    prog.close()
