"""Summarise rocprofv3 --pmc counter_collection CSVs (FETCH_SIZE / WRITE_SIZE, KB) per kernel.
Usage: python profiles/summarize_pmc.py out.json NAME=path.csv [NAME=path.csv ...]"""
import collections
import csv
import json
import sys

out = {}
for arg in sys.argv[2:]:
    name, path = arg.split("=", 1)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0][:90]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    out[name] = {k: dict(launches=n, total_KB=v, per_launch_KB=v / n) for k, (n, v) in agg.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
