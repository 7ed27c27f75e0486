"""Where do the wavefronts of k_expand_pairs<SpecGenT<GenProg>> (generated PlusCal code) spend their time?  The generated unit is built with
-DMC_PHASE_PROF through $TLAMC_JIT_DEFS (set here): shader-clock cycles per phase, exclusive, summed over all wavefronts of one complete search.
python profiles/phase_prof_gen.py  (the product library; the profiling build is the JIT library only)"""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ["TLAMC_JIT_DEFS"] = os.environ.get("PHASE_BASE_DEFS", "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1") + " -DMC_PHASE_PROF"
import tla_rust_amd as amd  # noqa: E402
from tla_rust_amd import binding as B  # noqa: E402

MSQ = "INVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n"
JOBS = [("pagecache N=3", "pagecache.tla", "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n", dict(table_capacity=1 << 27, arena_capacity=22 << 20)),
        ("ms_queue_counted N=3 K=3", "ms_queue_counted.tla", "CONSTANTS N = 3 K = 3 Counted = TRUE\n" + MSQ, dict(table_capacity=1 << 28, arena_capacity=40 << 20))]
names = {0: "row load + unpack", 1: "parent_status", 2: "summarize + guards", 3: "layout: key histogram + scatter", 7: "epilogue",
         8: "pass 1: eval_pair", 9: "pass 1: seen-set probe / insert", 10: "allocation (atomicAdd arena_next)", 11: "pass 2: eval_pair", 12: "pass 2: write_pair"}
L = B.lib()
L.mc_engine_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
for name, f, cfg, kw in JOBS:
    prog = amd.Program((ROOT / "specs" / "pluscal" / f).read_text(), cfg)
    eng = amd.Engine("pcal", prog.params, trace=False, timing=True, jit=True, chunk_states=1 << 21, **kw)
    out = (C.c_uint64 * 48)()
    eng.run()
    B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
    r = eng.run()
    B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
    ks = eng.kernel_stats()
    cyc = [int(out[i]) for i in range(24)]
    tot = sum(cyc)
    waves = int(out[40])
    rows = [dict(phase=names.get(i, f"phase {i}"), cycles_per_wave=round(cyc[i] / max(1, waves)), share=round(cyc[i] / max(1, tot), 4)) for i in range(24) if cyc[i]]
    print(json.dumps(dict(model=name, defs=os.environ["TLAMC_JIT_DEFS"], distinct=r.distinct, generated=r.generated, waves=waves, cycles_per_wave=round(tot / max(1, waves)),
                          expand_ms=ks["expand"]["ms_total"], levels=len(r.levels), largest_level=max(r.levels), seconds=r.seconds, us_per_level=round(1e6 * r.seconds / len(r.levels), 1), pairs_pass1_per_wave=round(int(out[24]) / max(1, waves), 1),
                          pairs_pass2_per_wave=round(int(out[25]) / max(1, waves), 1), phases=rows)), flush=True)
    eng.close()
    prog.close()
