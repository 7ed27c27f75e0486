# round 4, GPU call v: counters for the synthetic spec — N = 28 (8.6 GB seen-set) against N = 30 (34 GB): address translation (UTCL1 / UTCL2) and L2
cd /tmp && export TMPDIR=/tmp; R=/root/repo; D=$R/gpurun_out/r04v; mkdir -p $D
rocprofv3 --list-avail 2>/dev/null | grep -i -o "UTCL[0-9A-Za-z_]*\|TCP_TCC_[A-Z_]*REQ[a-z_]*\|TCP_PENDING[A-Za-z_]*\|TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_TAG_STALL[A-Za-z_]*" | sort -u | head -60 > $D/avail.txt; cat $D/avail.txt | tr '\n' ' '
for n in 28 30; do
  python $R/profiles/atomic_add_run.py $n 2>/dev/null | tail -1 | tee $D/plain_$n.txt
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $set --output-format csv -d $D/pmc_${n}_$name -- python $R/profiles/atomic_add_run.py $n > $D/pmc_${n}_$name.log 2>&1
    cp $D/pmc_${n}_$name/*/*_counter_collection.csv $D/pmc_${n}_$name.csv 2>/dev/null; rm -rf $D/pmc_${n}_$name
  done
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('/root/repo/gpurun_out/r04v/pmc_*.csv')):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_expand_insert' in r['Kernel_Name']: agg[r['Counter_Name']] += float(r['Counter_Value'])
    print(f.split('/')[-1], {k: f'{v:.4g}' for k, v in agg.items()})
PY
