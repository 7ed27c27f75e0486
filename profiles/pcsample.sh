#!/bin/bash
# profiles/pcsample.sh TAG — run on the GPU box (via gpurun): rocprofv3 PC sampling of bench.py's kernels (where do the
# wavefronts of k_expand_family spend their issue slots / stalls?).  Build with TLAMC_LINE_TABLES=1 for source lines.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/avail_full.txt 2>&1
grep -i -B2 -A12 "pc.sampl" $OUT/avail_full.txt | head -80 > $OUT/avail.txt
for IV in 1000 100 10; do
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap \
     --pc-sampling-interval $IV --output-format csv -d $OUT/pcs_$IV -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pcs_$IV.log 2>&1
  f=$(ls $OUT/pcs_$IV/*/*pc_sampling*.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
     wc -l $f > $OUT/pcs_$IV.count
     # histogram by (instruction text, comment): the raw file can be hundreds of MB
     python3 - "$f" "$OUT/pcs_$IV.hist.csv" <<'PY'
import csv, sys, collections
c = collections.Counter()
with open(sys.argv[1], newline='') as f:
    r = csv.DictReader(f)
    cols = r.fieldnames
    for row in r:
        c[(row.get('Instruction', ''), row.get('Instruction_Comment', ''))] += 1
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['count', 'instruction', 'comment', 'columns=' + '|'.join(cols or [])])
    for (i, cm), n in c.most_common(6000):
        w.writerow([n, i, cm])
PY
  fi
  rm -rf $OUT/pcs_$IV
  tail -c 600 $OUT/pcs_$IV.log > $OUT/pcs_$IV.tail; rm -f $OUT/pcs_$IV.log
done
rm -f $OUT/avail_full.txt.bak
ls -la $OUT
