"""Per BFS level of the LAST step in a rocprofv3 --kernel-trace CSV of bench.py: the level's wall time (first expand start to the next
level's first expand start), the time its expand kernels are busy (union), the idle time before the first expand of the next level
(after the last expand ended), and which kernels ran in that window.  A level boundary = a gap between expands that holds a
k_commit / k_end_level / memcpy and no expand."""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void mc::", "").split("<")[0]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_init_cand")]
fills = [i for i, r in enumerate(rows) if "fillBuffer" in r[2] and r[1] - r[0] > 1_000_000]
first = fills[-1] if fills and fills[-1] < starts[-1] else starts[-1]
step = rows[first:]
t0, t1 = step[0][0], max(r[1] for r in step)
print("step wall ms", round((t1 - t0) / 1e6, 3), " table clear ms", round((step[0][1] - step[0][0]) / 1e6, 3) if "fillBuffer" in step[0][2] else None)
ex = [(s, e) for s, e, n in step if n.startswith("k_expand")]
big = [(s, e) for s, e in ex if e - s > 200_000]
print("expand launches", len(ex), "of which > 0.2 ms", len(big), " their busy ms", round(sum(e - s for s, e in big) / 1e6, 2))
print("time before the first > 0.2 ms expand (clear + Init + the small levels) ms", round((big[0][0] - t0) / 1e6, 3))
print("time after the last > 0.2 ms expand ms", round((t1 - big[-1][1]) / 1e6, 3))
# gaps between consecutive big expands
tot = 0
hist = {}
for (s0, e0), (s1, e1) in zip(big, big[1:]):
    g = s1 - e0
    if g <= 0:
        continue
    tot += g
    inside = sorted(set(n for s, e, n in step if s < s1 and e > e0 and not n.startswith("k_expand")))
    key = "+".join(inside) or "nothing"
    h = hist.setdefault(key, [0, 0])
    h[0] += 1
    h[1] += g
print("gaps between > 0.2 ms expands: total ms", round(tot / 1e6, 3))
for k, (n, g) in sorted(hist.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:3d} gaps, {g / 1e6:7.3f} ms, avg {g / n / 1e3:7.1f} us : {k}")
