"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches).
Usage: [PMC_SPEC=raft|ssi|vm] python profiles/summarize_pmc.py out.json file.csv [file.csv ...]
The summary is stamped with the hash of the kernel sources of ONE spec (the one the counters were collected on): the engine's
kernels + that spec's lowering.  bench.py recomputes the stamp for the workload it times and refuses a summary of other sources."""
import collections
import csv
import hashlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SPEC = os.environ.get("PMC_SPEC", "raft")
# (raft keeps rounds 3-5's list, so that a summary of unchanged kernels keeps its stamp)
SPEC_SOURCES = {"raft": ["engine_kernels.h", "spec_raft.h", "mc_common.h"],
                "ssi": ["engine_kernels.h", "engine_pairs.h", "spec_ssi.h", "mc_common.h"],
                "vm": ["engine_kernels.h", "spec_vm.h", "mc_common.h"]}
KERNEL_SOURCES = ["tla_rust_amd/csrc/" + f for f in SPEC_SOURCES[SPEC]]


def kernel_source_hash():
    """identifies the kernels a counter pass was taken on: bench.py refuses traffic numbers whose stamp is not the timed library's"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        if (ROOT / f).exists():
            h.update((ROOT / f).read_bytes())
    return h.hexdigest()[:16]


out = collections.defaultdict(dict)
for path in sys.argv[2:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void mc::", "").replace("mc::", "")[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r["Dispatch_Id"])
    for k, d in agg.items():
        out[k].update(d)
        out[k]["launches"] = len(launches[k])
out["__source__"] = {"hash": kernel_source_hash(), "files": KERNEL_SOURCES, "spec": SPEC}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    if "k_" in k:
        print(k, {a: (f"{b:.4g}" if isinstance(b, float) else b) for a, b in d.items()})
