#!/bin/bash
# profiles/pcsample2.sh TAG [LIB] — run on the GPU box (via gpurun): rocprofv3 PC sampling (stochastic = hardware sampling with issue /
# stall information on gfx950; host_trap as the fallback) of two steps of the contract workload.  What the raw CSV (hundreds of MB)
# is reduced to: its header + first rows, a histogram of every low-cardinality column (stall reason, instruction type, ...), and the
# most-sampled instructions.  Counters / tracing are NOT combined with it.
TAG=${1:-rXX}; LIB=${2:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export TLAMC_LIB=$LIB
for M in "stochastic cycles 262144" "host_trap time 100"; do
  set -- $M
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $2 --pc-sampling-method $1 \
     --pc-sampling-interval $3 --output-format csv -d $OUT/pcs_$1 -- python $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-atomic-add > $OUT/pcs_$1.log 2>&1
  echo "$1: rc $?" >> $OUT/status.txt
  for f in $(ls $OUT/pcs_$1/*/*pc_sampling*.csv 2>/dev/null); do
     python3 - "$f" "$OUT/pcs_$1_$(basename $f .csv)" <<'PY'
import csv, sys, collections
src, stem = sys.argv[1], sys.argv[2]
hist = {}
top = collections.Counter()
n = 0
with open(src, newline='') as f:
    r = csv.DictReader(f)
    cols = r.fieldnames or []
    head = []
    for row in r:
        n += 1
        if n <= 5: head.append(row)
        for c in cols:
            h = hist.setdefault(c, collections.Counter())
            if len(h) < 400: h[row[c]] += 1
        top[tuple(row.get(c, '') for c in cols if c.lower() in ('instruction', 'instruction_comment', 'code_object_offset', 'kernel_name'))] += 1
with open(stem + '.summary.txt', 'w') as o:
    o.write(f"{n} samples; columns: {cols}\n")
    for hrow in head: o.write(str(hrow) + "\n")
    for c, h in hist.items():
        if len(h) < 60:
            o.write(f"\n== {c}\n")
            for k, v in h.most_common(): o.write(f"{v:10d} {100.0 * v / max(1, n):6.2f}%  {k}\n")
    o.write("\n== most sampled instructions\n")
    for k, v in top.most_common(400): o.write(f"{v:10d} {100.0 * v / max(1, n):6.2f}%  {k}\n")
PY
  done
  rm -rf $OUT/pcs_$1
  tail -c 1500 $OUT/pcs_$1.log > $OUT/pcs_$1.tail; rm -f $OUT/pcs_$1.log
  ls $OUT/*.summary.txt >/dev/null 2>&1 && break
done
ls -la $OUT
