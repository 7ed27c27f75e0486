"""The compiled-PlusCal path as GENERATED code (MC_F_JIT: pcal_codegen.cpp -> spec_gen.h, hipcc when the engine is created) against the
bytecode interpreter on the device, on the path's larger models: ms_queue_counted.tla N = 3, K = 3 / 4 (35 M / 125 M states; expected counts:
the same program on the host VM, profiles/bench_msq_counted.py) and pagecache.tla N = 3 (20 M states; golden tests/golden/pcal_channels.json).
One engine per back-end; the first engine of a program pays the build (reported), runs are timed best of 3.
Run on the GPU box: python profiles/bench_jit.py [msq3 msq4 pagecache]"""
import json, os, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import tla_rust_amd as amd

HOST = {3: (35263910, 99861367, 105), 4: (124636796, 353102053, 105)}
G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())
which = sys.argv[1:] or ["msq3", "msq4", "pagecache"]
JOBS = []
if "msq3" in which or "msq4" in which:
    src = (ROOT / "specs" / "pluscal" / "ms_queue_counted.tla").read_text()
    for k in (3, 4):
        if f"msq{k}" in which:
            JOBS.append((f"ms_queue_counted N=3 K={k}", src, f"CONSTANTS N = 3 K = {k} Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n",
                         dict(table_capacity=1 << (28 if k == 3 else 30), arena_capacity=(40 if k == 3 else 130) << 20, chunk_states=1 << 21), HOST[k]))
if "pagecache" in which:
    g = G["pagecache_n3"]
    JOBS.append(("pagecache N=3", (ROOT / "specs" / "pluscal" / "pagecache.tla").read_text(), "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n",
                 dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21), (g["distinct"], g["generated"], g["depth"])))
for name, src, cfg, kw, want in JOBS:
    prog = amd.Program(src, cfg)
    for jit, flags, label in ((True, 0, "generated code, pairs sorted by label (MC_F_JIT)"), (True, 32, "generated code, slot by slot (MC_F_JIT | MC_F_NOFAMILY)"),
                              (False, 0, "bytecode interpreter")):
        t0 = time.perf_counter()
        eng = amd.Engine("pcal", prog.params, trace=False, timing=True, jit=jit, debug_flags=flags, **kw)
        build_s = time.perf_counter() - t0
        best, r = 1e9, None
        for _ in range(3):
            t0 = time.perf_counter()
            r = eng.run()
            best = min(best, time.perf_counter() - t0)
        ks = eng.kernel_stats()
        eng.close()
        print(json.dumps(dict(workload=name, backend=label, distinct=r.distinct, generated=r.generated, depth=r.depth,
                              verdict=r.verdict, equals_expected=(r.distinct, r.generated, r.depth) == tuple(want), seconds=round(best, 4),
                              Mstates_s=round(r.distinct / best / 1e6, 1), Msuccessors_s=round(r.generated / best / 1e6, 1), engine_create_s=round(build_s, 2),
                              kernel_ms={k: round(ks[k]["ms_total"], 2) for k in ("expand", "insert", "materialise")}, state_bytes=ks["state_bytes"])), flush=True)
    prog.close()
