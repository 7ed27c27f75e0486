"""Seen-set size against the step (one process per setting; every run golden-gated).  The 32-byte probe mode needs table >= RATIO x arena
($TLAMC_SPARSE_RATIO, default 3); round 3 sized the contract workload's table at 40 << 26 slots (21.5 GB, load 0.2) with the probe loop of that
time — before the rotated slot order.  A smaller table is cleared faster (6.3 TB/s: 3.4 ms for 21.5 GB) and the device's random-read rate is higher
into a smaller region (41.2 G/s into 8 GiB, 38.7 G/s into 24 GiB: profiles/r05a_*), against more second-bucket probes.
python profiles/table_ab.py"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, str(ROOT))
    import tla_rust_amd as amd
    wl, slots = sys.argv[2], int(sys.argv[3])
    if wl == "t3":
        c = next(c for c in json.loads((ROOT / "tests" / "golden" / "raft_levels.json").read_text())["cases"] if c["name"] == "raft3_mcr4_t3_m1_k8_complete")
        eng = amd.Engine("raft", [3, 4, 3, 3, 1, 1, 8, 2, 4, 8], table_capacity=slots, arena_capacity=c["distinct"] + (1 << 20), chunk_states=(1 << 24) - 256, trace=False, timing=True)
        n, want = 6, c["levels"]
    elif wl == "ssi4x3":
        c = next(c for c in json.loads((ROOT / "tests" / "golden" / "ssi_levels.json").read_text())["cases"] if c["name"] == "ssi_4x3_levels10")
        eng = amd.Engine("ssi", [4, 3, 127, 0], table_capacity=slots, arena_capacity=c["distinct"] + (1 << 20), max_levels=10, chunk_states=(1 << 24) - 256, trace=False, timing=True)
        n, want = 8, c["levels"]
    else:
        raise SystemExit("raft5: use bench.py --workload raft5 --table-slots")
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = eng.run(); ts.append(time.perf_counter() - t0)
    ks = eng.kernel_stats()
    eng.close()
    ts = sorted(ts[1:])
    print(json.dumps(dict(workload=wl, slots=slots, slots_26=slots / (1 << 26), table_GB=round(slots * 8 / 1e9, 2), load=round(r.distinct / slots, 3), sparse_ratio=os.environ.get("TLAMC_SPARSE_RATIO", "3"),
                          ok=list(r.levels)[:len(want)] == want, ms_min=round(1e3 * ts[0], 2), ms_median=round(1e3 * ts[len(ts) // 2], 2), expand_ms_last_run=round(ks["expand"]["ms_total"], 2))), flush=True)
    sys.exit(0)
def one(wl, s26, ratio="3"):
    subprocess.run([sys.executable, __file__, "--one", wl, str(s26 << 26)], env=dict(os.environ, TLAMC_SPARSE_RATIO=ratio))
for s26, ratio in ((40, "3"), (32, "3"), (28, "3"), (24, "3"), (20, "2"), (16, "2"), (12, "1.5"), (40, "3"), (24, "3"), (16, "2"), (48, "3")):
    one("t3", s26, ratio)
for s26, ratio in ((9, "3"), (8, "3"), (6, "2"), (5, "1.5"), (12, "3"), (9, "3"), (6, "2")):
    one("ssi4x3", s26, ratio)
