"""Ablation of k_expand_insert on the bench workload: re-expand all resident states of a finished run
(every probe hits a full seen-set) with and without the probe phase."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd

P = [3, 4, 2, 3, 1, 1, 16, 2, 8]
eng = amd.Engine("raft", P, table_capacity=1 << 28, arena_capacity=30_000_000, chunk_states=1 << 20, max_distinct=25_000_000,
                 trace=False, timing=True)
r = eng.run()
ks = eng.kernel_stats()
print("run:", r.distinct, {k: round(ks[k]["ms_total"], 2) for k in ("expand", "materialise")}, "(expands", ks["expand"]["units"], "states)")
for name, fl in (("family kernel, probe(all hit)", 0), ("family kernel, no probe", 16), ("slot kernel, probe(all hit)", 32), ("slot kernel, no probe", 48),
                 ("family kernel, load parents only", 64 + 16), ("slot kernel, load parents only", 64 + 48)):
    ts = [eng.debug_reexpand(fl) for _ in range(3)]
    print(f"re-expand {r.distinct} states, {name}: {min(ts):.2f} ms")
