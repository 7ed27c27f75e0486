#!/usr/bin/env python
"""bench.py — distinct states/sec of the BFS hot path on examples/raft.tla (3 servers).

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`; one JSON line
from rank 0.  A "step" is one complete pass of the hot path over the workload: a full BFS of the
frozen model configuration from Init to the budget level (every level: expand -> seen-set insert
-> materialise), starting from a cleared seen-set.  Nothing is cached between steps.

Workload (BASELINE.json configs[2] "single-GPU full BFS"): examples/raft.tla under
specs/MCraft_t3.cfg — Server = {s1,s2,s3}, MaxClientRequests = 4 (=> MaxLogLen 3), MaxTerm = 3,
MaxMsgs = 1, MaxMsgKeys = 8, INVARIANT NoTwoLeaders — the COMPLETE state graph: 525,782,408 distinct /
6,708,500,293 generated / depth 33, verdict "ok" with nothing left on the queue (golden per-level counts
from the exact-dedup CPU oracle in tests/golden/raft_levels.json; the line is refused unless the run
reproduces them).  `--workload k10` is rounds 1-2's line (MaxTerm = 2, MaxMsgKeys = 10: 102,586,254 states).

`python bench.py --gpus N` (N > 1) starts its own N ranks, one per GPU; each builds the RCCL communicator (mc_comm_*) and
runs the library's sharded level loop (mc_shard_run) on the SAME complete graph ("scaling": "strong").
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

# params[6..8] = capacities of the messages / elections / allLogs slot arrays of the packed state = the
# oracle's maxima over the complete graph, 10 / 1 / 4 (tests/golden/raft_levels.json max_stat); an overflow
# would raise MC_EOVERFLOW, never drop a state.  params[9] = MaxMsgKeys.  W = 2 + 2*3 + 10/2 + 1*2 + 1 = 16 words = 128 B
# (round 3 layout: logs as 2 bits per client-request value, 32-bit messages two per word; 288 B in rounds 1-2).
WORKLOAD = dict(spec="raft", params=[3, 4, 2, 3, 1, 1, 10, 1, 4, 10], golden="raft3_mcr4_t2_m1_k10_complete",
                name="examples/raft.tla Server=3 MaxClientRequests=4 MaxTerm=2 MaxLogLen=3 MaxMsgs=1 MaxMsgKeys=10 "
                     "(specs/MCraft.cfg), complete state graph")
# --msg-keys 11: the next complete graph of the same model (336 581 097 states, 165 ms per step; golden made by the same oracle on
# the GPU box's host, tests/golden/raft_levels.json `source`): an optional longer step, NOT the default workload of the contract line
WORKLOAD_K11 = dict(spec="raft", params=[3, 4, 2, 3, 1, 1, 11, 1, 4, 11], golden="raft3_mcr4_t2_m1_k11_complete",
                    name="examples/raft.tla Server=3 MaxClientRequests=4 MaxTerm=2 MaxLogLen=3 MaxMsgs=1 MaxMsgKeys=11 "
                         "(specs/MCraft.cfg with MaxMsgKeys = 11), complete state graph")
# DEFAULT since round 3: the complete graph of the same model with MaxTerm = 3 — two elections, leader
# changes, the conflict-truncate branch of HandleAppendEntriesRequest reachable (with MaxTerm = 2 exactly one term of elections is
# explored and NoTwoLeaders is trivially true) — bounded by MaxMsgKeys = 8: 525 782 408 distinct / 6 708 500 293 generated / depth
# 33, verified by the exact-dedup oracle on the GPU box's host (tests/golden/raft_levels.json `source`); one step = 0.24 s of GPU time.
# Slot capacities 8 / 2 / 4 = the oracle's maxima (two election records now).  W = 2 + 2*3 + 8/2 + 2*2 + 1 = 17 words = 136 B.
WORKLOAD_T3 = dict(spec="raft", params=[3, 4, 3, 3, 1, 1, 8, 2, 4, 8], golden="raft3_mcr4_t3_m1_k8_complete",
                   name="examples/raft.tla Server=3 MaxClientRequests=4 MaxTerm=3 MaxLogLen=3 MaxMsgs=1 MaxMsgKeys=8 "
                        "(specs/MCraft_t3.cfg), complete state graph")
# BASELINE config 4's model (north_star's target sentence: raft.tla, 5 servers, bounded log, fingerprint-sharded over the GPUs of a node):
# Server = {s1..s5}, MaxClientRequests = 6 (log <= 5), MaxTerm = 2, MaxMsgs = 1, slot capacities 18 / 1 / 4 (W = 192 B), LEVEL-BUDGETED as
# SURVEY.md 8d prescribes: 18 BFS levels = 924 041 864 states / 177 GB of states — what ONE 288 GB device holds (level 19 alone has
# 1.26e9), so that N = 1, 2, 4, 8 run the same search ("scaling": "strong").  Gate: all 18 per-level counts equal the exact-dedup
# oracle's (tests/golden/raft_levels.json raft5_mcr6_t2_m1_levels18).
WORKLOAD_RAFT5 = dict(spec="raft", params=[5, 6, 2, 5, 1, 1, 18, 1, 4], golden="raft5_mcr6_t2_m1_levels18", max_levels=18, verdict="budget",
                      metric="distinct states/sec, raft.tla (5 servers, 18 BFS levels)",
                      name="examples/raft.tla Server=5 MaxClientRequests=6 MaxTerm=2 MaxLogLen=5 MaxMsgs=1 (BASELINE config 4), levels 1-18")
# BASELINE config 5: serializableSnapshotIsolation.tla, TxnId = {T1..T4}, Key = {K1,K2,K3}, every invariant of :59-79 on, no SYMMETRY,
# level-budgeted: 10 BFS levels = 168 M states (W = 80 B).  Gate: the oracle's per-level counts (tests/golden/ssi_levels.json).
WORKLOAD_SSI = dict(spec="ssi", params=[4, 3, 127, 0], golden="ssi_4x3_levels10", golden_file="ssi_levels.json", max_levels=10, verdict="budget",
                    packed_fanout=40, imbalance=1.8,   # (the last level is 87 % of the states and stays where it was generated)
                    metric="distinct states/sec, serializableSnapshotIsolation.tla (4 txns x 3 keys, 10 BFS levels)",
                    name="examples/serializableSnapshotIsolation.tla TxnId=4 Key=3 all invariants (BASELINE config 5), levels 1-10")
# --deep (the default of `--gpus N`, N >= 8: VERDICT round 5, next 6a): eight devices hold more than the one-GPU budget of configs 4 and 5, so the
# 8-rank form of these workloads goes one level deeper — config 5 to 11 levels (1 184 049 193 states, golden tests/golden/ssi_levels.json
# ssi_4x3_levels11), config 4 to 19 levels (the oracle's golden ends at 18: the first 18 per-level counts are gated, level 19 is reported)
WORKLOAD_SSI_DEEP = dict(WORKLOAD_SSI, golden="ssi_4x3_levels11", max_levels=11,
                         metric="distinct states/sec, serializableSnapshotIsolation.tla (4 txns x 3 keys, 11 BFS levels)",
                         name="examples/serializableSnapshotIsolation.tla TxnId=4 Key=3 all invariants (BASELINE config 5), levels 1-11")
WORKLOAD_RAFT5_DEEP = dict(WORKLOAD_RAFT5, max_levels=19, golden_prefix=18, expect_distinct=2_250_000_000,
                           metric="distinct states/sec, raft.tla (5 servers, 19 BFS levels)",
                           name="examples/raft.tla Server=5 MaxClientRequests=6 MaxTerm=2 MaxLogLen=5 MaxMsgs=1 (BASELINE config 4), levels 1-19 (levels 1-18 golden-gated)")
WORKLOADS = {"t3": WORKLOAD_T3, "k10": WORKLOAD, "k11": WORKLOAD_K11, "raft5": WORKLOAD_RAFT5, "ssi4x3": WORKLOAD_SSI}
DEEP = {"raft5": WORKLOAD_RAFT5_DEEP, "ssi4x3": WORKLOAD_SSI_DEEP}
DEEP_TABLE_SLOTS = {"raft5": 7 << 30, "ssi4x3": 60 << 26}
# seen-set slots: >= 3 x the arena, so that the table can never be more than a third full and is probed 32 bytes at a time
# (engine.hip seen_insert: random HBM reads cost by the byte); load 0.2 at the end of the run: 21.5 / 4.3 / 14 GB of the 288
TABLE_SLOTS = {"t3": 40 << 26, "k10": 8 << 26, "k11": 26 << 26, "raft5": 3 << 30, "ssi4x3": 9 << 26}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
XGMI_PEAK_GBS = 7 * 153.0  # MI355X_MICROARCH.md: 7 xGMI links x ~153 GB/s per GPU, point to point
# MEASURED on this device (profiles/calib/probe_locality.hip, round 5: profiles/r05a_probe_locality_8GiB.jsonl / _24GiB.jsonl): what the
# memory system serves when every request is a random 32-byte seen-set bucket — 41.2 G requests/s (1.32 TB/s) into an 8 GiB table, 38.7 G
# into 24 GiB, whatever the occupancy (3 .. 8 wavefronts per SIMD) or the probes in flight per lane (1, 2, 4); 55 - 58 G/s when the
# region fits the 256 MiB Infinity Cache, 235 G/s when it fits an XCD's 4 MiB L2; first-time inserts (read + compare-and-swap) 20 G/s.
# Partitioning the candidates by table region first (histogram + scatter + probe) ran at 0.52 - 0.79 x the unpartitioned rate.
RANDOM_REQ_CEILING = 41.2e9   # random 32-byte requests per second, whole-table
RANDOM_INSERT_CEILING = 20.0e9


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the environment
    torch.distributed.run would give them), wait, and take the first failing rank's siblings down with it.
    No rendezvous port: the 128 bytes of the RCCL communicator id travel through a file in a directory only this job knows
    ($TLAMC_BENCH_IDDIR, as `mc -gpus` does, mc_main.cpp) — round 3 picked a "free" port by bind(0) + close and rank 0 bound it
    seconds later, after `import torch`: anything could take it in between (EADDRINUSE on the driver's box)."""
    import shutil
    import tempfile
    iddir = tempfile.mkdtemp(prefix="tlamc_bench_")
    procs = []
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), TLAMC_BENCH_IDDIR=iddir,
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.pop("MASTER_PORT", None)   # (nothing listens anywhere)
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
        worst = 0
        left = set(range(a.gpus))
        while left:
            for r in list(left):
                rc = procs[r].poll()
                if rc is None:
                    continue
                left.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for q in left:   # nobody is left waiting in a collective for a rank that died
                        procs[q].terminate()
            time.sleep(0.05)
        return worst
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(iddir, ignore_errors=True)


def comm_id(rank, world):
    """The 128 bytes of the RCCL communicator id travel from rank 0 to the others
    * through a file when bench.py spawned the ranks itself ($TLAMC_BENCH_IDDIR: written under a temporary name and renamed,
      so a reader sees all of it or nothing; no socket, nothing to collide with);
    * through the launcher's TCP store under torch.distributed.run (the agent hosts it at MASTER_ADDR:MASTER_PORT and is
      listening before any rank starts);
    * ranks started by hand (RANK / WORLD_SIZE / MASTER_PORT set, no agent): rank 0 hosts the store and retries while the
      port is still held by something else."""
    from tla_rust_amd.binding import Comm
    iddir = os.environ.get("TLAMC_BENCH_IDDIR")
    if iddir:
        f = Path(iddir) / "comm_id"
        if rank == 0:
            uid = bytes(Comm.unique_id())
            tmp = Path(iddir) / "comm_id.tmp"
            tmp.write_bytes(uid)
            os.replace(tmp, f)
            return uid, None
        deadline = time.monotonic() + 300
        while not f.exists():
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {rank}: no communicator id in {f} after 300 s (did rank 0 start?)")
            time.sleep(0.02)
        return f.read_bytes(), None
    import datetime
    import torch.distributed as dist
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
    store, err = None, None
    for attempt in range(60):
        try:
            store = dist.TCPStore(addr, port, world_size=None if agent else world, is_master=(rank == 0 and not agent),
                                  timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
            break
        except (RuntimeError, OSError) as e:   # DistNetworkError (EADDRINUSE on the master, refused on a client): try again
            err = e
            time.sleep(0.5)
    if store is None:
        raise err
    key = "tlamc/bench/comm_id"
    if rank == 0:
        uid = Comm.unique_id()
        store.set(key, uid)
    else:
        uid = store.get(key)
    return bytes(uid), store


KERNEL_SOURCES = {"raft": ["engine_kernels.h", "spec_raft.h", "mc_common.h"],
                  "ssi": ["engine_kernels.h", "engine_pairs.h", "spec_ssi.h", "mc_common.h"],
                  "vm": ["engine_kernels.h", "spec_vm.h", "mc_common.h"]}


def kernel_source_hash(spec="raft"):
    """the stamp profiles/summarize_pmc.py puts into a PMC summary: the kernel sources of ONE spec the counters were collected on (the
    engine's kernels + that spec's lowering; the by-pairs kernel has its own file, so that the raft stamp does not move with it)"""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[spec]:
        if (ROOT / "tla_rust_amd" / "csrc" / f).exists():
            h.update((ROOT / "tla_rust_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def pmc_for(spec, stag, knames):
    """the newest profiles/r*_pmc.json that was collected on THIS tree's kernel sources of `spec` and holds a kernel named knames[i]<..stag..>:
    (file name, that kernel's counters) or (reason, None).  Counters of other kernels say nothing about the ones timed here."""
    want = kernel_source_hash(spec)
    seen = []
    for f in sorted((ROOT / "profiles").glob("r*_pmc.json"), reverse=True):
        try:
            d = json.loads(f.read_text())
        except ValueError:
            continue
        src = d.get("__source__", {})
        if src.get("spec", "raft") != spec:
            continue
        if src.get("hash") != want:
            seen.append(f"{f.name} was collected on kernel sources {src.get('hash')}")
            continue
        k = next((v for kn in knames for n, v in d.items() if n.startswith(kn + "<") and stag in n), None)
        if k is not None:
            return f.name, k
    return "none: " + ("; ".join(seen[:2]) if seen else f"no profiles/r*_pmc.json of spec {spec}") + f" (the timed sources are {want})", None


def kernel_roofline(ks, dt, spec, stag, slots, arena_states, generated, expand_kernel, no_family=False, matrix=False, dense_table=False):
    """The `roofline` object of a fused one-GPU run: the dominant kernel by summed HIP-event time, its algorithmic bytes over that time,
    the measured traffic / L2 hit rate / instructions per successor from the separate rocprofv3 --pmc passes of the same command (when a
    summary stamped with this tree's kernel sources exists), and the pipeline-level figure over the wall time of a step."""
    W = ks["state_bytes"]
    inwave = ks.get("inwave_states", 0)   # states the expand wavefronts wrote themselves: their W bytes are that kernel's
    expanded, written = ks["expand"]["units"], ks["materialise"]["units"]   # states read as parents / states in the arena at the end
    alg = {"expand": W * expanded + W * inwave, "insert": 8 * ks["cand_cells"], "materialise": W * (written - inwave)}
    state_only = alg["expand"]
    if not matrix:  # the seen-set insert is FUSED into the expand kernel: its 8 bytes per look-up (SURVEY 8d's G x 8 term, counted
        alg["expand"] += 8 * ks["cand_cells"]   # on the in-model, state-changing successors: the ones that ARE looked up) are that kernel's
    dom = max(("expand", "insert", "materialise"), key=lambda k: ks[k]["ms_total"])
    ach = alg[dom] / (ks[dom]["ms_total"] * 1e-3) / 1e9 if ks[dom]["ms_total"] else 0.0
    knames = {"expand": ("k_expand_insert",) if no_family else ("k_expand_family", "k_expand_pairs", "k_expand_insert"),
              "insert": ("k_insert",), "materialise": ("k_materialise",)}[dom]
    traffic, traffic_lower, l2_hit, l2_miss_rate, valu, salu, wait = None, None, None, None, None, None, None
    traffic_src, k = pmc_for(spec, stag, knames)
    if k is not None:
        # FETCH_SIZE / WRITE_SIZE are in KB.  Calibration on this box (profiles/r02e_calib_*.json*, profiles/calib/calib_fetch.hip):
        # FETCH_SIZE = read requests x 64 B; the arena's 8 B/lane row reads and 16 B/lane streaming reads are 128-B
        # requests (FETCH_SIZE = exactly 1/2 of the known bytes, as MI355X_MICROARCH.md says), a random 64-byte seen-set
        # bucket probe is ONE request (request size not observable); WRITE_SIZE equals the known bytes.  `traffic`
        # applies the guide's x2 to every read request (upper bound), `traffic_lower` counts a probe's request as 64 B.
        traffic = (2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / k["launches"]
        traffic_lower = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / k["launches"]
        l2_hit = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]) if "TCC_HIT_sum" in k else None
        if "TCC_MISS_sum" in k and ks[dom]["ms_total"]:   # (counters of ONE step against the timed step's kernel time: the same launches)
            l2_miss_rate = k["TCC_MISS_sum"] / k["launches"] * ks[dom]["launches"] / (ks[dom]["ms_total"] * 1e-3) / 1e9
        if "SQ_INSTS_VALU" in k and generated:   # wave-instructions per generated successor (the counter passes ran ONE step)
            valu, salu = k["SQ_INSTS_VALU"] / generated, k.get("SQ_INSTS_SALU", 0) / generated
        if "SQ_WAIT_ANY" in k and k.get("SQ_WAVE_CYCLES"):
            wait = k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]
        traffic_src = f"profiles/{traffic_src}: (2*FETCH_SIZE + WRITE_SIZE) per launch, separate --pmc passes of this command"
    kernel_name = {"expand": expand_kernel, "insert": "k_insert", "materialise": f"k_materialise<{stag}>"}[dom]
    # what the step really moves (a level-budgeted run never reads its last level): W x states read as parents + W x states written
    # + 8 x look-ups, over the WALL time of a step; `pipeline_frac_2WD` = SURVEY.md 8d's literal (2 W D + 8 G), which charges a read of
    # the unexpanded last level (87 % of config 5's states) and counts every generated successor as a look-up
    moved = W * expanded + W * written + 8 * ks["cand_cells"]
    return {"bound": "hbm", "kernel": kernel_name, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_lower": traffic_lower, "l2_hit_rate": l2_hit, "traffic_source": traffic_src,
            "launches": ks[dom]["launches"], "avg_launch_ms": ks[dom]["ms_total"] / max(1, ks[dom]["launches"]),
            "alg_bytes_per_launch": alg[dom] / max(1, ks[dom]["launches"]),
            "seen_set_lookups": ks["cand_cells"], "states_expanded": expanded, "states_written": written,
            "inwave_states": inwave,
            "alg_bytes": ("W x states expanded + W x states written in-wave + 8 x seen-set look-ups (in-model, state-changing successors; insert and write are fused into this kernel)" if dom == "expand" and not matrix
                          else "W x units"),
            "frac_state_bytes_only": (state_only / (ks[dom]["ms_total"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dom == "expand" and ks[dom]["ms_total"] else None,
            # the practical ceiling SURVEY 8d asks for: a probe moves a whole bucket (32 bytes in a sparse table, 64 in a full one)
            "probe_bytes": 32 if slots >= 3 * arena_states and not dense_table else 64,
            "kernel_ms": {k_: ks[k_]["ms_total"] for k_ in ("expand", "insert", "materialise")},
            "valu_per_successor": valu, "salu_per_successor": salu, "wait_frac_of_wave_cycles": wait,
            "pipeline_GBs": moved / dt / 1e9, "pipeline_frac": moved / dt / 1e9 / HBM_PEAK_GBS,
            "pipeline_frac_2WD": (2 * W * written + 8 * generated) / dt / 1e9 / HBM_PEAK_GBS,
            "pipeline_bytes": "W x states expanded + W x states written + 8 x seen-set look-ups over the wall time of a step (pipeline_frac_2WD: SURVEY 8d's literal 2 W D + 8 G)",
            "state_bytes": W,
            # the MEASURED random-access ceiling of the device (RANDOM_REQ_CEILING above) and what this kernel asks of the memory
            # system: L2 misses (probes, rows, parent rows the writer comes back for) + memory-side atomics per second of kernel time
            "random_access_ceiling_GBs": 32 * RANDOM_REQ_CEILING / 1e9, "random_access_ceiling_Greq_s": RANDOM_REQ_CEILING / 1e9,
            "l2_miss_Greq_s": l2_miss_rate, "frac_of_request_ceiling": (l2_miss_rate / (RANDOM_REQ_CEILING / 1e9)) if l2_miss_rate else None,
            "lookups_Gs": ks["cand_cells"] / (ks[dom]["ms_total"] * 1e-3) / 1e9 if dom == "expand" and ks[dom]["ms_total"] else None}


def expand_kernel_name(spec, params, no_family=False, matrix=False):
    stag = "SpecSsi" if spec == "ssi" else "SpecRaft<%d>" % params[0]
    if no_family or matrix or spec not in ("raft", "ssi"):
        return stag, f"k_expand_insert<{stag}>"
    return stag, (f"k_expand_family<{stag}>" if spec == "raft" else f"k_expand_pairs<{stag}>")


def dist_roofline(state_bytes, distinct, generated, world, step_s):
    """the N-rank line's HBM figure: SURVEY.md 8d's (2 W + 8 G/D) bytes per distinct state, a rank's share of them over the WALL time
    of a step — the pipeline-level number of the one-GPU line (`pipeline_frac`), not a kernel's (the kernels of a sharded step are
    timed by nobody: route-mode expand, probes, keep on three streams)"""
    alg = (2 * state_bytes * distinct + 8 * generated) / max(1, world)
    return {"bound": "hbm", "kernel": None, "achieved": alg / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / step_s / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "alg_bytes_per_step_per_gpu": alg, "state_bytes": state_bytes,
            "note": "per GPU, whole step: (2 W D + 8 G) / N bytes over the step's wall time; no kernel-level figure and no counters for the sharded step"}


def golden():
    g = json.loads((ROOT / "tests" / "golden" / WORKLOAD.get("golden_file", "raft_levels.json")).read_text())
    c = dict(next(c for c in g["cases"] if c["name"] == WORKLOAD["golden"]))
    if WORKLOAD.get("reduced_levels"):  # --levels L (tests): the first L levels of the golden, all gated; sizes follow
        c["prefix_levels"] = list(c["levels"][:WORKLOAD["reduced_levels"]])
        c["distinct"] = sum(c["prefix_levels"])
    elif WORKLOAD.get("golden_prefix"):   # a budget beyond the oracle's golden: its levels are a PREFIX of the run's (gated), the rest is reported
        c["prefix_levels"] = list(c["levels"])
        c["distinct"] = WORKLOAD["expect_distinct"]   # (sizes the arenas only)
    return c


def usable_cores():
    """cores this process may really use: the affinity mask and the cgroup CPU quota, not just the box's core count (round 3: the
    256-core GPU box gives its container ~16 cores' worth of CPU time — user time of a 192-thread run / its wall time = 15.9)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return n, quota


def cpu_baseline(max_seconds=10.0):
    """The exact-dedup CPU oracle ("port"; in-house CPU BFS, NOT TLC: the box has no JVM) on ALL host cores
    (TLC's run-book: "Number of worker threads: Use the number of cpu cores",
    examples/serializableSnapshotIsolation.tla:52-53): same model, BFS levels until `max_seconds` have passed; plus short
    samples at 1 / 8 / 64 threads, so that the line shows how the baseline itself scales (oracle/bfs_mt.c: per-thread arenas,
    one lock-free CAS table, exact comparison of the state bytes)."""
    exe = ROOT / "oracle" / "_build" / "oracle_mc"
    if not exe.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "oracle")], check=True)
    host_cores, quota = usable_cores()
    # threads: every core the process may use; under a CPU quota four threads per core's worth of quota (measured on the GPU
    # box, quota 16 of 256 cores: 64 threads 4.8 M states/s, 256 threads 2.8 M — throttled threads holding blocks of work)
    cores = host_cores if not quota else max(1, min(host_cores, int(4 * quota + 0.5)))
    wp = WORKLOAD["params"]
    spec = WORKLOAD["spec"]
    # the oracle's constants of the same model (its slot arrays are unbounded: no capacities; MaxMsgKeys is its 8th raft argument)
    p = [str(x) for x in (wp[:6] + ([0, wp[9]] if len(wp) > 9 else []) if spec == "raft" else wp)]

    def sample(threads, seconds):
        t0 = time.perf_counter()
        c0 = os.times()
        out = subprocess.run([str(exe), spec, *p, "--threads", str(threads), "--max-seconds", str(seconds), "--distinct", "200000000"],
                             capture_output=True, text=True, check=True).stdout
        c1 = os.times()
        r = json.loads(out.splitlines()[0])
        # CPU time the sample really got / its wall time = the cores it ran on (a container may be throttled far below nproc)
        r["cores_used"] = (c1.children_user + c1.children_system - c0.children_user - c0.children_system) / max(1e-9, time.perf_counter() - t0)
        return r

    r = sample(cores, max_seconds)
    table = [dict(threads=t, value=(q := sample(t, 3.0))["distinct"] / q["seconds"], levels=q["depth"], cores_used=round(q["cores_used"], 1))
             for t in (1, 8, 64) if t < cores]
    table.append(dict(threads=cores, value=r["distinct"] / r["seconds"], levels=r["depth"], cores_used=round(r["cores_used"], 1)))
    # stock TLC on the same box would be the preferred baseline — probe for it every time and say what was found
    import shutil
    java = shutil.which("java")
    jar = next((str(q) for d in ("/usr/share/java", "/opt", str(Path.home())) if Path(d).is_dir() for q in Path(d).glob("**/tla2tools.jar")), None) if java else None
    tlc = f"java at {java}, tla2tools.jar {'at ' + jar if jar else 'not found'}" if java else "no java on PATH"
    one = next((t["value"] for t in table if t["threads"] == 1), None)
    # how well the baseline itself scales: its rate against (cores it really ran on) x (its own single-thread rate) — VERDICT round 3 weak 13
    eff = (r["distinct"] / r["seconds"]) / (max(1.0, r["cores_used"]) * one) if one else None
    return dict(tlc_probe=tlc, value=r["distinct"] / r["seconds"], unit="distinct states/s", cores=cores, kind="port", scaling=table,
                cores_used=round(r["cores_used"], 1), cgroup_cpu_quota=quota, per_core=r["distinct"] / r["seconds"] / max(1.0, r["cores_used"]),
                parallel_efficiency=eff,
                sample=f"not TLC (no JVM on the box): in-house exact-dedup multi-threaded C BFS, {cores} threads on "
                       f"{r['cores_used']:.1f} cores' worth of CPU time (nproc {host_cores}, cgroup quota {quota}), same cfg, "
                       f"levels 1-{r['depth']} = {r['distinct']} distinct / {r['generated']} generated states in {r['seconds']:.1f} s "
                       f"(a sample stops after the level that exceeds its time; deeper levels have more duplicates per distinct state)")


def atomic_add_series(amd, device, n=28, steps=3):
    """north_star's second workload, in the same line: the synthetic N-process atomic-counter spec (atomic_add.tla:9-21 with N adders +
    the checker; SURVEY.md 8d config 2's throughput series).  One 64-bit word per state: the 32-byte random seen-set probe IS the
    workload.  Gate: the closed form D = 2^N + 1, G = N 2^(N-1) + 3, depth N + 2 (counts are not measured, they are checked)."""
    eng = amd.Engine("atomic_add", [n], device=device, table_capacity=1 << (n + 2), arena_capacity=(1 << n) + 4096, chunk_states=1 << 23,
                     trace=False, timing=True)
    eng.run()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.run()
    dt = (time.perf_counter() - t0) / steps
    ks = eng.kernel_stats()
    eng.close()
    D, G = (1 << n) + 1, n * (1 << (n - 1)) + 3
    if (r.distinct, r.generated, r.depth, r.verdict) != (D, G, n + 2, "ok"):
        print(f"bench.py: atomic_add N={n}: got {(r.distinct, r.generated, r.depth, r.verdict)}, the closed form is {(D, G, n + 2, 'ok')}", file=sys.stderr)
        sys.exit(1)
    alg = 2 * 8 * D + 8 * G   # (2 W + 8 G/D) bytes per distinct state, W = 8
    return {"workload": f"atomic_add.tla, {n} adders + checker (synthetic throughput series), complete graph", "value": D / dt, "unit": "distinct states/s",
            "ms_per_step": 1e3 * dt, "steps": steps, "distinct": D, "generated": G, "depth": n + 2, "generated_per_s": G / dt,
            "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dt / 1e9 / HBM_PEAK_GBS,
                         "alg_bytes": "(2 x 8 + 8 x G/D) bytes per distinct state over the wall time of a step",
                         "probe_GBs_at_32B": 32 * G / dt / 1e9,   # what the probes really move: a 32-byte bucket each
                         # the ceiling that binds this workload: the device's MEASURED random 32-byte request rate (see RANDOM_REQ_CEILING)
                         "random_access_ceiling_GBs": 32 * RANDOM_REQ_CEILING / 1e9, "random_access_ceiling_Gprobes_s": RANDOM_REQ_CEILING / 1e9,
                         "probes_Gs": G / dt / 1e9, "frac_of_random_access_ceiling": G / dt / RANDOM_REQ_CEILING,
                         "ceiling_source": "profiles/r05a_probe_locality_8GiB.jsonl (k_probe<4>, whole table); region-partitioned probing measured at 0.52-0.79 x",
                         "kernel_ms": {k: ks[k]["ms_total"] for k in ("expand", "insert", "materialise")}},
            "golden": "closed form 2^N + 1 / N 2^(N-1) + 3 / depth N + 2 (tests/test_oracle_golden.py: equal to the oracle for N <= 16)"}


def pcal_series(amd, device, steps=3):
    """The compiled-PlusCal path in the driver's line: two models nobody hand-lowered, compiled by the PlusCal
    front-end, translated into straight-line C++ and built for the device when the engine is created (MC_F_JIT: pcal_codegen.cpp, spec_gen.h),
    expanded by the by-pairs kernel with the pairs sorted by label.  Gates: pagecache.tla N = 3 = tests/golden/pcal_channels.json
    (counts and per-level counts); ms_queue_counted.tla N = 3, K = 3 = the same compiled program on the host build of the interpreter
    (profiles/bench_msq_counted.py).  `engine_create_s` is the load-time build (hipcc; cached by the program's hash afterwards)."""
    import io
    G = json.loads((ROOT / "tests" / "golden" / "pcal_channels.json").read_text())["pagecache_n3"]
    jobs = [("specs/pluscal/pagecache.tla N=3 (lock-free page cache, 3 threads)", ROOT / "specs" / "pluscal" / "pagecache.tla",
             "CONSTANTS N = 3 Blind = FALSE\nINVARIANTS Conservation HeadIsAllocated\n", dict(table_capacity=1 << 27, arena_capacity=22 << 20, chunk_states=1 << 21),
             (G["distinct"], G["generated"], G["depth"]), G["levels"], "tests/golden/pcal_channels.json:pagecache_n3"),
            ("specs/pluscal/ms_queue_counted.tla N=3 K=3 (Michael-Scott queue with counted pointers, 3 threads)", ROOT / "specs" / "pluscal" / "ms_queue_counted.tla",
             "CONSTANTS N = 3 K = 3 Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n",
             dict(table_capacity=1 << 28, arena_capacity=40 << 20, chunk_states=1 << 21), (35263910, 99861367, 105), None,
             "the same compiled program on the host build of the interpreter (profiles/bench_msq_counted.py)")]
    out = []
    for name, path, cfg, kw, want, levels, src in jobs:
        prog = amd.Program(path.read_text(), cfg)
        t0 = time.perf_counter()
        eng = amd.Engine("pcal", prog.params, device=device, trace=False, timing=True, jit=True, **kw)
        build_s = time.perf_counter() - t0
        eng.run()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = eng.run()
        dt = (time.perf_counter() - t0) / steps
        ks = eng.kernel_stats()
        eng.close()
        W_public = amd.state_bytes("pcal", prog.params)   # the interpreter's row (32 bits per cell): what leaves the engine
        prog.close()
        if (r.distinct, r.generated, r.depth) != tuple(want) or r.verdict != "ok" or (levels is not None and list(r["levels"]) != levels):
            print(f"bench.py: pcal {name}: got {(r.distinct, r.generated, r.depth, r.verdict)}, want {want}", file=sys.stderr)
            sys.exit(1)
        W = ks["state_bytes"]
        moved = 2 * W * r.distinct + 8 * ks["cand_cells"]
        out.append({"workload": name, "backend": "generated code (MC_F_JIT), by-pairs kernel" if ks.get("inwave_states", 0) else "slot-by-slot kernel (generated code or the interpreter: see stderr)",
                    "value": r.distinct / dt, "unit": "distinct states/s", "ms_per_step": 1e3 * dt, "steps": steps, "distinct": r.distinct, "generated": r.generated,
                    "depth": r.depth, "generated_per_s": r.generated / dt, "engine_create_s": build_s, "state_bytes": W, "state_bytes_interpreter": W_public,
                    "kernel_ms": {k: ks[k]["ms_total"] for k in ("expand", "insert", "materialise")},
                    "pipeline_GBs": moved / dt / 1e9, "pipeline_frac": moved / dt / 1e9 / HBM_PEAK_GBS, "golden": src})
    return out


def other_config(amd, device, key, steps=5):
    """BASELINE.json's configs 4 and 5 name models that are meant for eight GPUs; their graphs up to a level budget fit ONE MI355X, and the
    driver's line carries them as objects of their own (round 5: until then only builder-run `--workload raft5 / ssi4x3` lines existed,
    which nobody else measured).  Same engine, same gate as the main line: the run must reproduce the oracle's golden — counts and every
    per-level count — or nothing is printed.  One untimed run, `steps` timed ones."""
    w = WORKLOADS[key]
    g = json.loads((ROOT / "tests" / "golden" / w.get("golden_file", "raft_levels.json")).read_text())
    G0 = next(c for c in g["cases"] if c["name"] == w["golden"])
    slots = TABLE_SLOTS[key]
    ML = w.get("max_levels", 0)
    eng = amd.Engine(w["spec"], w["params"], device=device, table_capacity=slots, arena_capacity=G0["distinct"] + (1 << 20), max_levels=ML,
                     chunk_states=(1 << 24) - 256, trace=False, timing=True)
    eng.run()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.run()
    dt = (time.perf_counter() - t0) / steps
    ks = eng.kernel_stats()
    eng.close()
    got, want = (r.distinct, r.generated, r.depth, r.verdict), (G0["distinct"], G0["generated"], G0["depth"], w.get("verdict", "ok"))
    if got != want or list(r["levels"]) != G0["levels"]:
        print(f"bench.py: {key}: run does not reproduce the golden state graph: got {got}, want {want}", file=sys.stderr)
        sys.exit(1)
    W, D, G = ks["state_bytes"], r.distinct, r.generated
    stag, ek = expand_kernel_name(w["spec"], w["params"])
    roof = kernel_roofline(ks, dt, w["spec"], stag, slots, G0["distinct"] + (1 << 20), G, ek)
    park = None
    if key == "raft5":
        # the same steps through the PARK instantiation of the by-family kernel (MC_F_PARK, VERDICT round 5 next 5: every state written in-wave,
        # nothing through k_materialise), timed beside the default inside the driver's own run: the A/B DESIGN.md quotes is re-measured every time
        eng = amd.Engine(w["spec"], w["params"], device=device, table_capacity=slots, arena_capacity=G0["distinct"] + (1 << 20), max_levels=ML,
                         chunk_states=(1 << 24) - 256, trace=False, timing=True, debug_flags=32768)
        eng.run()
        t0 = time.perf_counter()
        for _ in range(steps):
            rp = eng.run()
        dtp = (time.perf_counter() - t0) / steps
        ksp = eng.kernel_stats()
        eng.close()
        if (rp.distinct, rp.generated, rp.depth, rp.verdict) != want or list(rp["levels"]) != G0["levels"]:
            print(f"bench.py: {key} with MC_F_PARK: run does not reproduce the golden state graph", file=sys.stderr)
            sys.exit(1)
        park = {"flag": "MC_F_PARK (bench.py --park): the in-wave writers park the overflow of their survivor lists, the workgroup's tail writes it in later rounds",
                "ms_per_step": 1e3 * dtp, "inwave_states": ksp.get("inwave_states", 0),
                "kernel_ms": {k: ksp[k]["ms_total"] for k in ("expand", "insert", "materialise")}, "default": False}
    return {**({"park_ab": park} if park else {}),
            "workload": w["name"] + " — ONE GPU, fused engine (the 8-GPU form of this configuration is `bench.py --gpus 8 --workload " + key + "`)",
            "value": D / dt, "unit": "distinct states/s", "ms_per_step": 1e3 * dt, "steps": steps, "distinct": D, "generated": G, "depth": r.depth,
            "verdict": r.verdict, "state_bytes": W, "seen_set_load": D / float(slots), "inwave_states": ks.get("inwave_states", 0),
            "kernel_ms": {k: ks[k]["ms_total"] for k in ("expand", "insert", "materialise")},
            "roofline": roof, "pipeline_GBs": roof["pipeline_GBs"], "pipeline_frac": roof["pipeline_frac"], "alg_bytes": roof["pipeline_bytes"],
            "golden": f"tests/golden/{w.get('golden_file', 'raft_levels.json')}:{w['golden']} (exact-dedup CPU oracle): counts and per-level counts equal"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)   # 40 complete BFS runs of 50 ms: two seconds of GPU work in the timed region
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-atomic-add", action="store_true", help="skip the synthetic N-process atomic-counter object of the line")
    ap.add_argument("--no-pcal", action="store_true", help="skip the compiled-PlusCal object of the line (two models as generated code: MC_F_JIT)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the one-GPU objects of BASELINE configs 4 and 5 (raft with 5 servers, SSI 4 x 3)")
    ap.add_argument("--max-distinct", type=int, default=0, help="A/B only: stop at a budget (the line is then marked invalid)")
    ap.add_argument("--chunk", type=int, default=(1 << 24) - 256)   # frontier states per launch: the engine's maximum (a column index has 24 bits).  Round 3: 169.4 / 163.6 / 160.5 ms
    # per step at 2^21 / 2^22 / 2^23; round 5 (profiles/r05k_ab.jsonl): 2^24 - 256 against 2^23 = 135.3 against 135.5 - 136.3 ms on t3 (70 launches instead of 102),
    # 91.4 against 92.7 on k11, 181 - 184 against 192 - 197 on raft5
    ap.add_argument("--shard-chunk", type=int, default=0, help="frontier states per round and rank in the sharded (--gpus N) path; 0 = 2^23 at world "
                    "size 1 (one engine launch per round: 168.9 ms per step against 184.1 at 2^21), 2^21 otherwise (several rounds per level, so that "
                    "the exchange of round r+1 overlaps the probes of round r)")
    ap.add_argument("--packed-fanout", type=int, default=0, help="in-model successors per state the fixed-capacity exchange buckets allow for "
                    "(0 = the workload's: 16 for the raft models, 40 for the SI model, whose frontier grows 8 x per level)")
    ap.add_argument("--exchange", choices=["auto", "exact", "measured", "packed"], default="auto",
                    help="--gpus N: how a stay level exchanges its candidates (include/tlamc.h MC_SHARD_*): host-paced rounds with exact sizes, "
                         "pipelined fixed-capacity buckets sized from the previous level's measured fill, or the same buckets from --packed-fanout; "
                         "auto (default) = one untimed trial step in the exact and in the measured form, the faster one (max over ranks) is timed: "
                         "which of the two wins depends on what a host wait + two small all-gathers per round cost over xGMI, and no 8-GPU box "
                         "has told yet")
    ap.add_argument("--force-shard", action="store_true", help="--gpus 1 under a launcher: run the N-rank engine at world size 1 (RCCL, route-mode "
                    "kernels) instead of the fused engine — the sharded path's own overhead (+9 % on t3, profiles/r04q_bench_world1_rccl.json)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: every rank on GPU 0 (needs a librccl stand-in in $TLAMC_RCCL; RCCL refuses it)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="t3", help="t3: MaxTerm 3, MaxMsgKeys 8 (525.8 M states, the contract line); "
                    "k10 / k11: MaxTerm 2 with 10 / 11 message keys (102.6 M / 336.6 M states; rounds 1-2's line was k10); raft5: BASELINE config 4 "
                    "(5 servers, 18 levels = 924 M states); ssi4x3: BASELINE config 5 (10 levels = 168 M states)")
    ap.add_argument("--deep", dest="deep", action="store_true", default=None, help="--workload raft5 / ssi4x3: one BFS level beyond the one-GPU budget (19 / 11 levels); "
                    "the default with --gpus 8 and more")
    ap.add_argument("--no-deep", dest="deep", action="store_false")
    ap.add_argument("--levels", type=int, default=0, help="TEST ONLY: stop after this many BFS levels (a prefix of the workload's golden, every level gated; "
                    "seen-set and arenas sized for it) — the 8-rank command lines of configs 4 and 5 on one shared GPU")
    ap.add_argument("--table-slots", type=int, default=0, help="seen-set slots (any multiple of 64; 0 = the workload's default, load ~0.5 at the end)")
    ap.add_argument("--table-log2", type=int, default=0, help="A/B: a power-of-two seen-set (27: load 0.76, 28: load 0.38)")
    ap.add_argument("--msg-keys", type=int, default=0, choices=[0, 10, 11], help="(rounds 1-2) same as --workload k10 / k11")
    ap.add_argument("--matrix", action="store_true", help="A/B: unfused candidate-matrix kernels")
    ap.add_argument("--no-family", action="store_true", help="A/B: expand slot by slot instead of by action family")
    ap.add_argument("--dense-table", action="store_true", help="A/B: 64-byte (8-slot) seen-set buckets even when the table is sparse enough for 32-byte probes")
    ap.add_argument("--occ3", action="store_true", help="A/B: by-family expand kernel compiled for 3 waves per SIMD (no register spills)")
    ap.add_argument("--no-filter", action="store_true", help="A/B: without the per-wavefront duplicate filter in front of the seen-set")
    ap.add_argument("--sync-probe", action="store_true", help="A/B: resolve every batch of seen-set probes right where it is issued (MC_F_SYNCPROBE)")
    ap.add_argument("--wave-tail", action="store_true", help="A/B: in-wave writes by wavefront (no workgroup barrier) instead of by workgroup")
    ap.add_argument("--park", action="store_true", help="A/B: the in-wave writers PARK the overflow of their survivor lists and the workgroup's own tail writes it in "
                    "later rounds (MC_F_PARK: the PARK instantiation of the by-family kernel) instead of sending it through the new-list to k_materialise")
    ap.add_argument("--no-inwave", action="store_true", help="A/B: every new state through the new-list and k_materialise (rounds 1-3) instead of "
                    "being written by the expand wavefront that found it")
    a = ap.parse_args()
    global WORKLOAD
    if a.msg_keys:
        a.workload = "k%d" % a.msg_keys
    WORKLOAD = WORKLOADS[a.workload]
    if a.deep is None:
        a.deep = a.gpus >= 8
    deep = bool(a.deep) and a.workload in DEEP
    if deep:
        WORKLOAD = DEEP[a.workload]
    if a.levels:   # the same command line at a reduced budget: L levels of the workload's golden, every one gated (tests/test_gpu_sharded.py)
        WORKLOAD = dict(WORKLOAD, max_levels=a.levels, reduced_levels=a.levels, name=WORKLOAD["name"] + f" — REDUCED to {a.levels} levels (--levels: a functional run, not the benchmark)")
        if not a.table_slots:
            a.table_slots = max(1 << 22, (8 * golden()["distinct"]) // 64 * 64)
    if not a.table_slots:
        a.table_slots = DEEP_TABLE_SLOTS[a.workload] if deep else TABLE_SLOTS[a.workload]
    if not a.packed_fanout:
        a.packed_fanout = WORKLOAD.get("packed_fanout", 16)

    if a.dense_table:
        os.environ["TLAMC_DENSE_TABLE"] = "1"
    launched = "RANK" in os.environ
    if a.gpus > 1 and not launched:
        sys.exit(spawn_ranks(a))

    import torch
    import tla_rust_amd as amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    # one rank of N.  At world size 1 — the N = 1 point of a launched scaling run — the fused engine runs, as without a launcher: the
    # N-rank engine at world size 1 is the same search with the exchange machinery idling beside it (route-mode kernels, three streams)
    # and cost 9 % in round 4; a SCALE curve's first point should be the one-GPU line.  --force-shard keeps
    # that path measurable.
    use_dist = launched and (world > 1 or a.force_shard)
    if not a.shard_chunk:
        a.shard_chunk = (1 << 23) if world == 1 else (1 << 21)

    G0 = golden()
    ML = WORKLOAD.get("max_levels", 0)   # level-budgeted workloads (BASELINE configs 4, 5: SURVEY.md 8d); 0 = the complete graph
    slots = (1 << a.table_log2) if a.table_log2 else a.table_slots
    comm = store = None
    if not use_dist:
        eng = amd.Engine(WORKLOAD["spec"], WORKLOAD["params"], device=local, table_capacity=slots, matrix=a.matrix,
                         debug_flags=(32 if a.no_family else 0) | (2048 if a.occ3 else 0) | (8192 if a.no_filter else 0) | (65536 if a.no_inwave else 0) | (131072 if a.wave_tail else 0) | (4096 if a.sync_probe else 0) | (32768 if a.park else 0),
                         arena_capacity=G0["distinct"] + (1 << 20), max_levels=ML,
                         chunk_states=a.chunk, max_distinct=a.max_distinct, trace=False, timing=True)
        run = eng.run
        stats = {}
    else:
        # strong scaling: the same complete graph, its seen-set and frontier sharded over the ranks by fingerprint; the level loop
        # is the library's (mc_shard_run: C++, collectives issued with RCCL directly — no Python inside a step)
        from tla_rust_amd.binding import Comm
        uid, store = comm_id(rank, world)
        comm = Comm(uid, rank, world, local)
        eng = amd.Engine(WORKLOAD["spec"], WORKLOAD["params"], device=local, shard_rank=rank, shard_count=world, trace=False, timing=True,
                         chunk_states=a.shard_chunk, max_distinct=a.max_distinct, table_capacity=(slots * 4 // 3) // world // 64 * 64,
                         # a rank's share of the states (+25 % imbalance allowance) + the replicated prefix
                         arena_capacity=int(G0["distinct"] / world * (WORKLOAD.get("imbalance", 1.25) if world > 1 else 1.0)) + (1 << 22))
        stats = {}

        def run():
            r, st = comm.shard_run(eng, chunk_states=a.shard_chunk, max_distinct=a.max_distinct, max_levels=ML, packed_fanout=a.packed_fanout,
                                    exchange=a.exchange)
            stats.update(st)
            return r

        trial = None
        if a.exchange == "auto" and world == 1:
            a.exchange = "exact"   # (a single rank exchanges nothing)
        if a.exchange == "auto":   # one untimed step in each form; every rank sees the same maxima, so every rank picks the same form
            trial = {}
            for form in ("exact", "measured"):
                a.exchange = form
                comm.all_gather_u64(0)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                try:
                    run()
                    mine = time.perf_counter() - t1
                except amd.McError as e:   # (every rank leaves mc_shard_run with the same agreed code: every rank lands here, or none)
                    print(f"bench.py: rank {rank}: the trial step in the {form} form failed ({e}); not chosen", file=sys.stderr)
                    mine = float("inf")
                torch.cuda.synchronize()
                trial[form] = max(comm.all_gather_f64(mine))
            a.exchange = min(trial, key=trial.get)
            if trial[a.exchange] == float("inf"):
                print("bench.py: both trial steps failed", file=sys.stderr)
                sys.exit(1)
            trial = {k: (v if v != float("inf") else None) for k, v in trial.items()}

    def barrier():
        if use_dist:
            comm.all_gather_u64(0)
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
        dt = max(comm.all_gather_f64(dt))                                   # MAX over ranks
        shares = comm.all_gather_u64(stats.get("distinct_local", 0))
        sent = comm.all_gather_u64(stats.get("sent_bytes", 0))
        routed = comm.all_gather_u64(stats.get("routed_candidates", 0))
        fpans = comm.all_gather_u64(stats.get("fp_answer_bytes", 0))
        # per rank, last step: the GPU time of its three kinds of kernels (HIP events on the engine's own streams) and the host time of
        # its level loop inside engine calls (launches + the waits for its own streams) and inside collectives (the exchange + waiting for
        # the slowest peer) — what a step's wall time is made of, rank by rank
        ksd = eng.kernel_stats()
        per_rank = {k: comm.all_gather_f64(ksd[n]["ms_total"]) for k, n in (("expand_ms", "expand"), ("probe_ms", "insert"), ("keep_ms", "materialise"))}
        per_rank["engine_host_ms"] = [x / 1e6 for x in comm.all_gather_u64(stats.get("engine_ns", 0))]
        per_rank["collective_host_ms"] = [x / 1e6 for x in comm.all_gather_u64(stats.get("collective_ns", 0))]
        per_rank["collectives"] = comm.all_gather_u64(stats.get("collectives", 0))
    if rank != 0:
        eng.close()
        comm.close()
        return
    D, G = res.distinct, res.generated
    if not a.max_distinct:
        # parity gate: the measured run must BE the golden graph (exact counts, per-level where the engine reports them)
        got = (D, G, res.depth, res.verdict)
        want = (G0["distinct"], G0["generated"], G0["depth"], WORKLOAD.get("verdict", "ok"))
        if "prefix_levels" in G0:   # gated on the golden's levels, one more level reported
            ok = "levels" in res and list(res["levels"])[:len(G0["prefix_levels"])] == G0["prefix_levels"] and res.depth == ML and res.verdict == want[3]
            got, want = (list(res["levels"])[:len(G0["prefix_levels"])] if "levels" in res else None, res.depth, res.verdict), (G0["prefix_levels"], ML, want[3])
            if not ok:
                print(f"bench.py: run does not reproduce the golden prefix: got {got}, want {want}", file=sys.stderr)
                sys.exit(1)
        elif got != want or ("levels" in res and list(res["levels"]) != G0["levels"]):
            print(f"bench.py: run does not reproduce the golden state graph: got {got}, want {want}", file=sys.stderr)
            sys.exit(1)
    line = {
        "metric": WORKLOAD.get("metric", "distinct states/sec, raft.tla (3 servers)"), "value": D * a.steps / dt, "unit": "distinct states/s",
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "strong" if a.gpus > 1 else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": WORKLOAD["name"], "distinct": D, "generated": G, "depth": res.depth,
                   "verdict": res.verdict, "queue_left": res.queue_left, "generated_per_s": G * a.steps / dt,
                   "seen_set_load": D / float(slots) if not use_dist else None,
                   "golden": f"tests/golden/{WORKLOAD.get('golden_file', 'raft_levels.json')}:{WORKLOAD['golden']} (exact-dedup CPU oracle, its multi-threaded BFS on the GPU box's host): counts and per-level counts equal"
                             if not a.max_distinct else "A/B run with a budget: NOT the benchmark"},
    }
    if use_dist:
        step_s = dt / a.steps
        how = {"exact": "host-paced rounds with exact sizes: all_to_all_v of 8 bytes per routed candidate out and 1 byte back, the next round's expand in flight meanwhile",
               "measured": "pipelined fixed-capacity rounds, buckets sized from the previous level's measured fill: the exchange of round r+1 overlaps the probes of round r and the next expand",
               "packed": "pipelined fixed-capacity rounds, buckets sized from the fan-out allowance: the exchange of round r+1 overlaps the probes of round r and the next expand"}[a.exchange]
        line["config"]["parallelism"] = (f"fingerprint-sharded seen-set x{world} (owner = fingerprint high bits), replicated prefix for the small "
                                         f"levels, two-phase fingerprint-first exchange over RCCL (mc_shard_run: level loop in C++; large levels: {how})")
        line["config"]["shares"] = shares
        line["config"]["exchange"] = a.exchange
        if trial is not None:
            line["config"]["exchange_trial_ms"] = {k: (1e3 * v if v is not None else None) for k, v in trial.items()}   # (--exchange auto: one untimed step in each form; null = it failed)
        line["per_rank"] = dict(per_rank, rounds=stats.get("rounds"), host_ms_per_round=[(e + c) / max(1, stats.get("rounds", 0))
                                for e, c in zip(per_rank["engine_host_ms"], per_rank["collective_host_ms"])],
                                note="last step; *_ms of kernels = summed HIP-event times on the engine's streams (they overlap each other); "
                                     "engine_host / collective_host = wall time of this rank's level loop inside engine calls / inside collectives")
        line["config"]["levels"] = {k: stats.get(k) for k in ("replicated_levels", "stay_levels", "move_levels", "rounds")}
        line["config"]["frontier_imbalance"] = stats.get("max_frontier", 0) / max(1, stats.get("mean_frontier", 1))
        # xGMI roofline: bytes a rank hands to the all-to-alls for OTHER ranks per step, against 7 links x 153 GB/s
        per_rank = max(sent) if sent else 0
        line["xgmi"] = {"bound": "xgmi", "sent_bytes_per_step_per_gpu": per_rank, "achieved": per_rank / step_s / 1e9, "peak": XGMI_PEAK_GBS,
                        "unit": "GB/s", "frac": per_rank / step_s / 1e9 / XGMI_PEAK_GBS,
                        # what the exchange NEEDS: 9 bytes (fingerprint out, answer back) per candidate a rank sent to another rank's seen-set
                        # slice — counted by the loop (mc_shard_stats.routed_candidates), not estimated — against what it MOVED for them
                        "exchange": a.exchange, "routed_candidates_per_step": sum(routed), "model_bytes_per_step": 9 * sum(routed),
                        "fp_answer_bytes_per_step": sum(fpans), "sent_over_model": sum(fpans) / max(1, 9 * sum(routed)),
                        "restarts": stats.get("restarts", 0), "measured_levels": stats.get("measured_levels", 0),
                        "note": "packed / measured: the buckets of a stay round are moved whole (sent bytes include their unused tails); "
                                "exact: all_to_all_v moves what the counts say"}
        try:   # (never worth the line: a failure here leaves the object out)
            line["roofline"] = dist_roofline(amd.state_bytes(WORKLOAD["spec"], WORKLOAD["params"]), D, G, world, step_s)
        except Exception as e:  # noqa: BLE001
            line["roofline"] = None
            line["config"]["roofline_error"] = str(e)
        line["cpu_baseline"] = None   # (timed beside the one-GPU line only: rank 0 at N = 1)
        if a.share_gpu:
            line["config"]["NOT_A_MEASUREMENT"] = "all ranks share ONE GPU through a librccl stand-in ($TLAMC_RCCL): a functional run of the N-rank path, not a scaling number"
        eng.close()
        comm.close()
    else:
        ks = eng.kernel_stats()
        stag, ek = expand_kernel_name(WORKLOAD["spec"], WORKLOAD["params"], a.no_family, a.matrix)
        line["roofline"] = kernel_roofline(ks, dt / a.steps, WORKLOAD["spec"], stag, slots, G0["distinct"] + (1 << 20), G, ek,
                                           no_family=a.no_family, matrix=a.matrix, dense_table=a.dense_table)
        if not a.no_atomic_add and not a.max_distinct and a.workload == "t3":   # (the contract line carries both of north_star's workloads)
            eng.close()
            line["atomic_add"] = atomic_add_series(amd, local)
        if not a.no_other_configs and not a.max_distinct and a.workload == "t3" and not a.matrix and not a.no_family:
            # ... and BASELINE configs 4 and 5 as far as one GPU holds them (a few seconds: two engines of 200 GB and 30 GB come and go)
            eng.close()
            line["config4_model_one_gpu"] = other_config(amd, local, "raft5")
            line["config5_model_one_gpu"] = other_config(amd, local, "ssi4x3")
        if not a.no_pcal and not a.max_distinct and a.workload == "t3" and not a.matrix and not a.no_family:
            eng.close()
            line["pcal"] = pcal_series(amd, local)
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
