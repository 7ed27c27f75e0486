"""Where does a step's wall time go that is not inside k_expand_family?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --steps 1`,
takes the LAST complete BFS (the timed step) and prints: busy time of each kernel, the union of all kernels' busy time, the time the
expand stream is idle while a materialise runs (expand waiting for a new-list to be free) and while nothing runs at all."""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mc::", "")[:40]))
rows.sort()
# the last run = after the last k_init_cand
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_init_cand")]
rows = rows[starts[-1]:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
ex = [(s, e) for s, e, n in rows if n.startswith("k_expand")]
ma = [(s, e) for s, e, n in rows if n.startswith("k_materialise")]
print("step wall ms", (t1 - t0) / 1e6, "kernels", len(rows))
print("expand busy ms", union(ex) / 1e6, "launches", len(ex), " materialise busy ms", union(ma) / 1e6)
print("any kernel busy ms", union([(s, e) for s, e, n in rows]) / 1e6)
print("expand OR materialise busy ms", union(ex + ma) / 1e6)
# gaps between consecutive expand launches, split by whether a materialise covers the gap
gaps = []
exs = sorted(ex)
for (s0, e0), (s1, e1) in zip(exs, exs[1:]):
    if s1 > e0:
        cov = sum(max(0, min(e, s1) - max(s, e0)) for s, e in ma)
        gaps.append((s1 - e0, cov))
print("gaps between expands: total ms", sum(g for g, _ in gaps) / 1e6, "of which a materialise is running", sum(c for _, c in gaps) / 1e6,
      " count > 50 us:", sum(1 for g, _ in gaps if g > 50000))
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (ms, covered by materialise ms):", [(round(g / 1e6, 2), round(c / 1e6, 2)) for g, c in big])
