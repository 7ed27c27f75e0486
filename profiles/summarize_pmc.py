"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches).
Usage: python profiles/summarize_pmc.py out.json file.csv [file.csv ...]"""
import collections
import csv
import json
import sys

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
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, d in out.items():
    if "k_" in k:
        print(k, {a: (f"{b:.4g}" if isinstance(b, float) else b) for a, b in d.items()})
