"""Ablation of k_expand_family on the bench workload (complete graph): re-expand all resident states of the finished run
(every probe finds its fingerprint: bucket reads, no CAS) with and without the probe phase, and the parent loads alone."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
import bench

G0 = bench.golden()
for tl in (27, 28):
    eng = amd.Engine("raft", bench.WORKLOAD["params"], table_capacity=1 << tl, arena_capacity=G0["distinct"] + (1 << 20), chunk_states=1 << 22,
                     trace=False, timing=True)
    r = eng.run()
    ks = eng.kernel_stats()
    print(f"table 2^{tl} run:", r.distinct, r.verdict, {k: round(ks[k]["ms_total"], 2) for k in ("expand", "materialise")}, "(expands", ks["expand"]["units"], "states)")
    for name, fl in (("probe (all hit)", 0), ("no probe", 16), ("load parents only (phase A)", 64 + 16)):
        ts = [eng.debug_reexpand(fl) for _ in range(3)]
        print(f"  re-expand {r.distinct} states, {name}: {min(ts):.2f} ms")
    del eng
