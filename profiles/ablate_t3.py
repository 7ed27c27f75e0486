"""Ablation of k_expand_family on the CONTRACT workload (t3, complete graph, 525.8 M resident states): after the run, re-expand every
resident state (each probe finds its fingerprint: a bucket read, no compare-and-swap, no survivor, no tail) with the probe phase, without
it (flag 16: generation only), and the parent loads alone (flag 64).  Against the run's own expand time this splits a step into
generation / probes / inserts + writes.  python profiles/ablate_t3.py > gpurun_out/ablate_t3.json"""
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import tla_rust_amd as amd
import bench

W = bench.WORKLOADS["t3"]
bench.WORKLOAD = W
G0 = bench.golden()
eng = amd.Engine("raft", W["params"], table_capacity=bench.TABLE_SLOTS["t3"], arena_capacity=G0["distinct"] + (1 << 20), chunk_states=(1 << 24) - 256,
                 trace=False, timing=True)
r = eng.run()
assert r.distinct == G0["distinct"] and r.generated == G0["generated"], (r.distinct, r.generated)
ks = eng.kernel_stats()
out = dict(workload="t3", distinct=r.distinct, generated=r.generated, run_expand_ms=round(ks["expand"]["ms_total"], 2), run_seconds=round(r.seconds * 1e3, 2))
for name, fl in (("reexpand_all_probes_hit_ms", 0), ("reexpand_no_probe_ms", 16), ("reexpand_parent_loads_only_ms", 64 + 16), ("reexpand_no_filter_ms", 8192)):
    out[name] = round(min(eng.debug_reexpand(fl) for _ in range(3)), 2)
print(json.dumps(out))
