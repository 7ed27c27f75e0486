"""The register / workgroup shape of the by-pairs kernel in a GENERATED translation unit ($TLAMC_JIT_DEFS): wavefronts per SIMD the register
allocation leaves room for (MC_PAIR_MINW: 2 = up to 256 VGPRs) x wavefronts per workgroup (MC_PAIR_WAVES), on the two models of the driver
line's `pcal` object.  One process per shape (the knob is read when the engine is built).  python profiles/jit_defs_ab.py"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT))
    import time
    import tla_rust_amd as amd
    G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
    jobs = [("pagecache N=3", (ROOT / "specs" / "pluscal" / "pagecache.tla").read_text(), "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n",
             dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21), (G["distinct"], G["generated"], G["depth"])),
            ("ms_queue_counted N=3 K=3", (ROOT / "specs" / "pluscal" / "ms_queue_counted.tla").read_text(),
             "CONSTANTS N = 3 K = 3 Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n",
             dict(table_capacity=1 << 28, arena_capacity=40 << 20, chunk_states=1 << 21), (35263910, 99861367, 105))]
    for name, src, cfg, kw, want in jobs:
        prog = amd.Program(src, cfg)
        t0 = time.perf_counter()
        eng = amd.Engine("pcal", prog.params, trace=False, timing=True, jit=True, **kw)
        build_s = time.perf_counter() - t0
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            r = eng.run()
            best = min(best, time.perf_counter() - t0)
        eng.close()
        print(json.dumps(dict(workload=name, shape=os.environ.get("TLAMC_JIT_DEFS", "(default) -DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1"), ok=(r.distinct, r.generated, r.depth) == want,
                              ms=round(1e3 * best, 2), G_states_s=round(r.distinct / best / 1e9, 3), build_s=round(build_s, 1))), flush=True)
    sys.exit(0)
for minw, waves in ((2, 1), (3, 1), (4, 1), (2, 2), (3, 2), (2, 4)):
    env = dict(os.environ, TLAMC_JIT_DEFS=f"-DMC_PAIR_MINW={minw} -DMC_PAIR_WAVES={waves}")
    subprocess.run([sys.executable, __file__, "--one"], env=env)
