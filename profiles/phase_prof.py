"""Where do the wavefronts of k_expand_family spend their time?  Needs a library built with TLAMC_PHASE_PROF=1
(python -c "import tla_rust_amd.build as b; b.build(force=True)" under that variable): cycles per phase, exclusive, summed over
all wavefronts of one complete BFS of the bench model.  python profiles/phase_prof.py [msg_keys]"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
from tla_rust_amd import binding as B

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
DEBUG_FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # e.g. 131072 = MC_F_WAVETAIL, 65536 = MC_F_NOINWAVE
params = [3, 4, 2, 3, 1, 1, K, 1, 4, K]
if K == 8:   # the contract workload: MaxTerm 3, MaxMsgKeys 8 (525.8 M states)
    params = [3, 4, 3, 3, 1, 1, 8, 2, 4, 8]
eng = amd.Engine("raft", params, table_capacity=(8 << 26) if K == 10 else (40 << 26) if K == 8 else (26 << 26), arena_capacity=103_000_000 if K == 10 else 527_000_000 if K == 8 else 340_000_000, chunk_states=1 << 23, trace=False, timing=True, debug_flags=DEBUG_FLAGS)
L = B.lib()
L.mc_engine_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 48)()
eng.run()
B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
r = eng.run()
B._check(L.mc_engine_debug_phases(eng._h, out, 1), "mc_engine_debug_phases")
ks = eng.kernel_stats()
FAM = ["F_REQVOTE", "F_APPEND", "F_MISC"]
names = ["load_expand", "dense pairs", "enqueue", "flush_probe", "flush_out / tail: wait at the workgroup barrier", "push fixed", "push messages", "epilogue+drain"] + \
        [f"phaseB {FAM[f] if f < len(FAM) else f}" for f in range(8)] + ["tail: counting sort + allocation", "tail: writes"] + [f"phase {f}" for f in range(18, 24)]
cyc = [int(out[i]) for i in range(24)]
tot = sum(cyc)
waves = int(out[40])
rows = [dict(phase=names[i], cycles_per_wave=round(cyc[i] / max(1, waves)), share=round(cyc[i] / tot, 4),
             pairs_per_wave=round(int(out[24 + i - 8]) / max(1, waves), 1) if 8 <= i < 16 else None) for i in range(24) if cyc[i]]
print(json.dumps(dict(distinct=r.distinct, generated=r.generated, waves=waves, cycles_per_wave=round(tot / max(1, waves)), expand_ms=ks["expand"]["ms_total"],
                      materialise_ms=ks["materialise"]["ms_total"], phases=rows), indent=1))
