#!/usr/bin/env python
"""bench.py — distinct states/sec of the BFS hot path on examples/raft.tla (3 servers).

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`; one JSON line
from rank 0.  A "step" is one complete pass of the hot path over the workload: a full BFS of the
frozen model configuration from Init to the budget level (every level: expand -> seen-set insert
-> materialise), starting from a cleared seen-set.  Nothing is cached between steps.

Workload (BASELINE.json configs[2]): examples/raft.tla under
specs/MCraft.cfg — Server = {s1,s2,s3}, MaxClientRequests = 4 (=> MaxLogLen 3), MaxTerm = 2,
MaxMsgs = 1, INVARIANT NoTwoLeaders, level-budgeted at the first level whose cumulative distinct
count reaches 25,000,000 (25,752,293 distinct / 23 levels; golden per-level counts from the CPU
oracle in tests/golden/raft_levels.json).
"""
import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# params[6..8] = capacities of the messages / elections / allLogs slot arrays of the packed state:
# the oracle's maxima over this prefix are 14 / 1 / 4 (tests/golden/raft_levels.json max_stat);
# an overflow would raise MC_EOVERFLOW, never drop a state.  W = 18 + 16 + 2*4 + 8 = 50 words.
WORKLOAD = dict(spec="raft", params=[3, 4, 2, 3, 1, 1, 16, 2, 8], max_distinct=25_000_000,
                name="examples/raft.tla Server=3 MaxClientRequests=4 MaxTerm=2 MaxLogLen=3 MaxMsgs=1, budget 25M distinct")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(sample_distinct=1_500_000):
    """The oracle ("port") timed on the host cores: same workload, bounded prefix."""
    exe = ROOT / "oracle" / "_build" / "oracle_mc"
    if not exe.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle")], check=True)
    p = [str(x) for x in WORKLOAD["params"][:6]]
    out = subprocess.run([str(exe), "raft", *p, "--distinct", str(sample_distinct)], capture_output=True, text=True, check=True).stdout
    r = json.loads(out.splitlines()[0])
    return dict(value=r["distinct"] / r["seconds"], unit="distinct states/s", cores=1, kind="port",
                sample=f"same cfg, BFS prefix to {r['distinct']} distinct / {r['generated']} generated states "
                       f"({r['seconds']:.1f} s, single-thread exact-dedup C oracle, in-house CPU BFS — not TLC)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-distinct", type=int, default=WORKLOAD["max_distinct"])
    ap.add_argument("--chunk", type=int, default=1 << 22)
    ap.add_argument("--shard-chunk", type=int, default=1 << 21, help="frontier states per round and rank in the sharded (torchrun) path")
    ap.add_argument("--table-log2", type=int, default=28)
    ap.add_argument("--matrix", action="store_true", help="A/B: unfused candidate-matrix kernels")
    ap.add_argument("--no-family", action="store_true", help="A/B: expand slot by slot instead of by action family")
    a = ap.parse_args()

    import torch
    import tla_rust_amd as amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    import torch.distributed as dist
    use_dist = "RANK" in os.environ  # launched by torch.distributed.run: sharded path, RCCL all-to-all
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    if not use_dist:
        eng = amd.Engine(WORKLOAD["spec"], WORKLOAD["params"], device=local, table_capacity=1 << a.table_log2, matrix=a.matrix, debug_flags=32 if a.no_family else 0,
                         arena_capacity=30_000_000 if a.max_distinct <= 25_000_000 else 2 * a.max_distinct,
                         chunk_states=a.chunk, max_distinct=a.max_distinct, trace=False, timing=True)
        run = eng.run
    else:
        from tla_rust_amd.sharded import ShardedChecker
        # weak scaling: the distinct-state budget grows with the number of GPUs
        # deeper levels than the single-GPU prefix are reached: the message-slot maximum grows by about one per
        # level (13 at level 23), so the sharded run gets 24 message slots (W = 464 B) instead of 16
        sharded_params = WORKLOAD["params"][:6] + [int(os.environ.get("TLAMC_SHARD_CM", "24")), 2, 8]
        chk = ShardedChecker(WORKLOAD["spec"], sharded_params, device=local, max_distinct=a.max_distinct * world,
                             chunk_states=a.shard_chunk, table_capacity=1 << 27,
                             # the last level may overshoot the budget by the growth factor (~1.7x): size for it
                             arena_capacity=64_000_000,
                             fanout_cap=48, new_cap=6)
        run = chk.run

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(a.steps):
        res = run()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    D, G = res.distinct, res.generated
    line = {
        "metric": "distinct states/sec, raft.tla (3 servers)", "value": D * a.steps / dt, "unit": "distinct states/s",
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": WORKLOAD["name"], "distinct": D, "generated": G, "depth": res.depth,
                   "verdict": res.verdict, "generated_per_s": G * a.steps / dt},
    }
    if use_dist:
        line["config"]["parallelism"] = (f"fingerprint-sharded seen-set x{world}, replicated prefix for the small levels, "
                                         f"pipelined two-phase all-to-all over RCCL")
        dist.destroy_process_group()
    else:
        ks = eng.kernel_stats()
        W = ks["state_bytes"]
        # algorithmic bytes per launch: expand reads W per frontier state,
        # insert touches one 8-byte seen-set word per generated candidate, materialise writes W per new state
        alg = {"expand": W * ks["expand"]["units"], "insert": 8 * ks["cand_cells"], "materialise": W * ks["materialise"]["units"]}
        dom = max(("expand", "insert", "materialise"), key=lambda k: ks[k]["ms_total"])
        n_runs = a.steps  # stats are reset by every run(): they describe the last step
        ach = alg[dom] / (ks[dom]["ms_total"] * 1e-3) / 1e9 if ks[dom]["ms_total"] else 0.0
        traffic, traffic_src = None, None
        pmc = sorted((ROOT / "profiles").glob("r*_pmc.json"))
        if pmc:  # HBM bytes per launch from the separate rocprofv3 --pmc passes of this same command
            try:
                d = json.loads(pmc[-1].read_text())
                knames = {"expand": ("k_expand_insert",) if a.no_family else ("k_expand_family", "k_expand_insert"),
                          "insert": ("k_insert",), "materialise": ("k_materialise",)}[dom]
                k = next(v for kn in knames for n, v in d.items() if n.startswith(kn + "<") and "Raft<3>" in n)
                # FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE counts wide coalesced reads at 1/2 (MI355X_MICROARCH.md §HBM)
                traffic = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / k["launches"]
                traffic_src = f"profiles/{pmc[-1].name}: (2*FETCH_SIZE + WRITE_SIZE) per launch, separate --pmc passes"
            except Exception:  # noqa: BLE001
                traffic = None
        kernel_name = {"expand": "k_expand_insert<SpecRaft<3>>" if (a.no_family or a.matrix) else "k_expand_family<SpecRaft<3>>",
                       "insert": "k_insert", "materialise": "k_materialise<SpecRaft<3>>"}[dom]
        line["roofline"] = {"bound": "hbm", "kernel": kernel_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                            "launches": ks[dom]["launches"], "avg_launch_ms": ks[dom]["ms_total"] / max(1, ks[dom]["launches"]),
                            "alg_bytes_per_launch": alg[dom] / max(1, ks[dom]["launches"]),
                            "kernel_ms": {k: ks[k]["ms_total"] for k in ("expand", "insert", "materialise")},
                            "pipeline_GBs": (2 * W * D + 8 * G) / (1e-3 * sum(ks[k]["ms_total"] for k in ("expand", "insert", "materialise"))) / 1e9}
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
