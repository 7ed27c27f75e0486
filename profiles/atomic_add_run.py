"""one complete BFS of atomic_add with N adders (argv[1]) — for rocprofv3 --pmc passes: why does N = 30 probe at half the rate of N = 28?"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
n = int(sys.argv[1])
eng = amd.Engine("atomic_add", [n], table_capacity=1 << (n + 2), arena_capacity=(1 << n) + 4096, chunk_states=1 << 23, trace=False, timing=True)
t = time.perf_counter()
r = eng.run()
dt = time.perf_counter() - t
ks = eng.kernel_stats()
print({"n": n, "ms": round(1e3 * dt, 1), "distinct": r.distinct, "probes_per_s_G": round(ks["cand_cells"] / dt / 1e9, 2), "kernel_ms": {k: round(ks[k]["ms_total"], 1) for k in ("expand", "materialise")}})
eng.close()
