# round 6, GPU call zzb: config 5's model with the survivors ALWAYS compacted before pass 2 (MC_PAIR_COMPACT_Q = 4: 81 % of its pairs survive; the shipped threshold, 3 / 4, leaves
# the SI model as it was) against the product library, alternating, ten steps each; VALU / SALU instructions of both by one PMC pass each
cd /root/repo; D=$PWD/gpurun_out/r06zzb; mkdir -p $D
for rep in 1 2 3; do
for lib in product ssiab; do
  L=""; [ $lib = ssiab ] && L=$PWD/tla_rust_amd/_build/libtlamc_ssiab.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload ssi4x3 --steps 10 --no-cpu-baseline 2>>$D/err.txt | grep -v amdgpu.ids | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(json.dumps({'lib': '$lib', 'rep': $rep, 'ms_per_step': round(d['ms_per_step'], 3), 'expand_ms': round(r['kernel_ms']['expand'], 3) if isinstance(r.get('kernel_ms'), dict) else r.get('kernel_ms')}))" | tee -a $D/ssi_compact_q_ab.jsonl
done; done
cd /tmp && export TMPDIR=/tmp
for lib in product ssiab; do
  L=""; [ $lib = ssiab ] && L=/root/repo/tla_rust_amd/_build/libtlamc_ssiab.so
  TLAMC_LIB=$L rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $D/pmc_$lib -- python /root/repo/bench.py --workload ssi4x3 --steps 1 --warmup 0 --no-cpu-baseline > $D/pmc_$lib.log 2>&1
  python - <<PY
import csv, glob
tot = {}
for f in glob.glob('$D/pmc_$lib/*/*_counter_collection.csv'):
    for row in csv.DictReader(open(f)):
        if 'k_expand_pairs' in row['Kernel_Name']:
            tot[row['Counter_Name']] = tot.get(row['Counter_Name'], 0.0) + float(row['Counter_Value'])
print('$lib', {k: round(v / 200276725, 3) for k, v in tot.items() if k != 'SQ_WAVES'}, 'per generated successor (one step)')
PY
  rm -rf $D/pmc_$lib
done
grep -v amdgpu.ids $D/err.txt | tail -3
