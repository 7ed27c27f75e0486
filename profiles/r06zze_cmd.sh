# round 6, GPU call zze: rocprofv3 kernel stats + counters of the by-pairs kernel on GENERATED code (packed rows, final tree): pagecache N = 3 and ms_queue_counted K = 3,
# four searches each per process (profiles/pcal_pack_ab.py, packed form only); counters in their own passes
cd /root/repo; D=$PWD/gpurun_out/r06zze; mkdir -p $D
export PACK_AB_ONLY=1 PACK_AB_JOBS=2
python profiles/pcal_pack_ab.py > $D/warm.jsonl 2>/dev/null   # (fills the compiler cache: the profiled processes load the libraries)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python /root/repo/profiles/pcal_pack_ab.py > $D/trace.log 2>&1
cp $D/trace/*/*_kernel_stats.csv $D/pcal_kernel_stats.csv 2>/dev/null; rm -rf $D/trace
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --output-format csv -d $D/pmc_$name -- python /root/repo/profiles/pcal_pack_ab.py > $D/pmc_$name.log 2>&1
  cat $D/pmc_$name/*/*_counter_collection.csv > $D/pmc_$name.csv 2>/dev/null; rm -rf $D/pmc_$name
done
python - <<'PY'
import csv, glob, json
D = '/root/repo/gpurun_out/r06zze'
tot = {}
for f in glob.glob(D + '/pmc_*.csv'):
    for row in csv.DictReader(open(f)):
        if row.get('Kernel_Name', '').find('k_expand_pairs') >= 0 and row['Counter_Name'] != 'Counter_Name':
            tot[row['Counter_Name']] = tot.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
succ = 4 * (47629297 + 99861367)   # generated successors of the eight searches of one process
out = {'kernel': 'k_expand_pairs<SpecGenT<GenProg>> (both programs, 4 searches each)', 'counters': tot, 'generated_successors': succ,
       'valu_per_successor': tot.get('SQ_INSTS_VALU', 0) / succ, 'salu_per_successor': tot.get('SQ_INSTS_SALU', 0) / succ,
       'wait_any_of_wave_cycles': tot.get('SQ_WAIT_ANY', 0) / max(1.0, tot.get('SQ_WAVE_CYCLES', 0))}
json.dump(out, open(D + '/pcal_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out)[:900])
PY
head -4 $D/pcal_kernel_stats.csv | cut -c1-60,200-330
