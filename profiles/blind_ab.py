"""The largest frontier a BATCHED level takes ($TLAMC_BLIND_LOG2; 16 = 65 536 states since round 2): models with many levels of a few hundred
thousand states — the compiled PlusCal models: 37 / 105 levels — pay a host round trip per level beyond it.  One process per setting.
python profiles/blind_ab.py"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT))
    import tla_rust_amd as amd
    G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
    L = os.environ.get("TLAMC_BLIND_LOG2", "16")
    jobs = [("pagecache N=3", (ROOT / "specs" / "pluscal" / "pagecache.tla").read_text(), "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n",
             dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21), (G["distinct"], G["generated"], G["depth"])),
            ("ms_queue_counted N=3 K=3", (ROOT / "specs" / "pluscal" / "ms_queue_counted.tla").read_text(),
             "CONSTANTS N = 3 K = 3 Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n",
             dict(table_capacity=1 << 28, arena_capacity=40 << 20, chunk_states=1 << 21), (35263910, 99861367, 105))]
    for name, src, cfg, kw, want in jobs:
        prog = amd.Program(src, cfg)
        eng = amd.Engine("pcal", prog.params, trace=False, timing=True, jit=True, **kw)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); r = eng.run(); best = min(best, time.perf_counter() - t0)
        eng.close()
        print(json.dumps(dict(workload=name + " (generated code)", blind_log2=L, ok=(r.distinct, r.generated, r.depth) == want, ms=round(1e3 * best, 2), G_states_s=round(r.distinct / best / 1e9, 3))), flush=True)
    g = json.loads((ROOT / "tests" / "golden" / "ssi_levels.json").read_text())
    c = next(c for c in g["cases"] if c["name"] == "ssi_4x3_levels10")
    eng = amd.Engine("ssi", [4, 3, 127, 0], table_capacity=9 << 26, arena_capacity=c["distinct"] + (1 << 20), max_levels=10, chunk_states=(1 << 24) - 256, trace=False, timing=True)
    best = 1e9
    for _ in range(6):
        t0 = time.perf_counter(); r = eng.run(); best = min(best, time.perf_counter() - t0)
    eng.close()
    print(json.dumps(dict(workload="ssi4x3 10 levels", blind_log2=L, ok=list(r.levels) == c["levels"], ms=round(1e3 * best, 2))), flush=True)
    g = json.loads((ROOT / "tests" / "golden" / "raft_levels.json").read_text())
    c = next(c for c in g["cases"] if c["name"] == "raft3_mcr4_t3_m1_k8_complete")
    eng = amd.Engine("raft", [3, 4, 3, 3, 1, 1, 8, 2, 4, 8], table_capacity=40 << 26, arena_capacity=c["distinct"] + (1 << 20), chunk_states=(1 << 24) - 256, trace=False, timing=True)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); r = eng.run(); best = min(best, time.perf_counter() - t0)
    eng.close()
    print(json.dumps(dict(workload="raft t3 complete", blind_log2=L, ok=list(r.levels) == c["levels"], ms=round(1e3 * best, 2))), flush=True)
    sys.exit(0)
for lg in (16, 18, 20, 22, 16, 20):
    subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, TLAMC_BLIND_LOG2=str(lg)))
