"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches).
Usage: python profiles/summarize_pmc.py out.json file.csv [file.csv ...]"""
import collections
import csv
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SOURCES = ["tla_rust_amd/csrc/engine_kernels.h", "tla_rust_amd/csrc/spec_raft.h", "tla_rust_amd/csrc/mc_common.h"]


def kernel_source_hash():
    """identifies the kernels a counter pass was taken on: bench.py refuses traffic numbers whose stamp is not the timed library's"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
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
out["__source__"] = {"hash": kernel_source_hash(), "files": KERNEL_SOURCES}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    if "k_" in k:
        print(k, {a: (f"{b:.4g}" if isinstance(b, float) else b) for a, b in d.items()})
