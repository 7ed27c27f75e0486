"""How many seen-set probes would a direct-mapped cache of recently probed fingerprints answer?  Host simulation (tests/_shim) of the
by-family kernel's candidate stream in arena order: python profiles/probe_cache_sim.py K [log2 sizes ...]  (raft, 3 servers, MaxTerm 2)"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import helpers

K = int(sys.argv[1]) if len(sys.argv) > 1 else 7
sizes = [int(x) for x in sys.argv[2:]] or [12, 14, 16, 18, 20]
lib = helpers.shim_lib()
d = helpers.spec_desc("raft", [3, 4, 2, 3, 1, 1, K, 1, 4, K])
out = (C.c_uint64 * 8)()
for lg in sizes:
    lib.shim_probe_cache_sim(C.byref(d), C.c_uint64(0), lg, out)
    cands, wave, cache, dups, new = (int(out[i]) for i in range(5))
    print(f"K={K} cache 2^{lg:2d} entries: candidates {cands}, wave filter {wave / cands:.3f}, cache {cache / cands:.3f} "
          f"(= {cache / max(1, cands - wave):.3f} of the probes), table: duplicates {dups / cands:.3f}, new {new / cands:.3f}; states {new}")
