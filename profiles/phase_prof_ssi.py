"""Where do the wavefronts of k_expand_pairs<SpecSsi> spend their time?  Needs a library whose SSI translation unit was built with
-DMC_PHASE_PROF (TLAMC_LIB=tla_rust_amd/_build/libtlamc_ssiprof.so): shader-clock cycles per phase, exclusive, summed over all wavefronts
of one run of BASELINE config 5's model (4 x 3, 10 levels).  python profiles/phase_prof_ssi.py"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
from tla_rust_amd import binding as B

eng = amd.Engine("ssi", [4, 3, 127, 0], table_capacity=9 << 26, arena_capacity=169_200_000, chunk_states=(1 << 24) - 256, max_levels=10, trace=False, timing=True)
L = B.lib()
L.mc_engine_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 48)()
eng.run()
B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
r = eng.run()
B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
ks = eng.kernel_stats()
names = {0: "row load + S::load (tables)", 1: "parent_status (invariants)", 2: "summarize + pair base + guards", 3: "layout: scan + scatter", 7: "epilogue",
         8: "pass 1: eval_pair", 9: "pass 1: seen-set probe / insert", 10: "allocation (atomicAdd arena_next)", 11: "pass 2: eval_pair", 12: "pass 2: write_pair"}
cyc = [int(out[i]) for i in range(24)]
tot = sum(cyc)
waves = int(out[40])
rows = [dict(phase=names.get(i, f"phase {i}"), cycles_per_wave=round(cyc[i] / max(1, waves)), share=round(cyc[i] / tot, 4)) for i in range(24) if cyc[i]]
print(json.dumps(dict(distinct=r.distinct, generated=r.generated, waves=waves, cycles_per_wave=round(tot / max(1, waves)), expand_ms=ks["expand"]["ms_total"],
                      pairs_pass1_per_wave=round(int(out[24]) / max(1, waves), 1), pairs_pass2_per_wave=round(int(out[25]) / max(1, waves), 1), phases=rows), indent=1))
