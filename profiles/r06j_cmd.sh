# round 6, GPU call j: rocprofv3 kernel stats of `bench.py --workload ssi4x3` for the product (step status + commit family), the full-status
# build and the three-family build: the check_frontier and expand kernels' own durations (the wall-clock A/B of call i was noisy)
cd /root/repo; D=$PWD/gpurun_out/r06j; mkdir -p $D; B=$PWD/tla_rust_amd/_build
cd /tmp && export TMPDIR=/tmp
for v in new fullstatus fam3 new fullstatus fam3; do
  L=$B/libtlamc.so; [ $v != new ] && L=$B/libtlamc_$v.so
  TLAMC_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace_$v -- python /root/repo/bench.py --workload ssi4x3 --steps 20 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs > $D/trace_$v.log 2>&1
  python - <<PY
import csv, glob
f = glob.glob('$D/trace_$v/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void mc::','')[:40]
    if 'expand_pairs' in n or 'check_frontier' in n: print('$v', n, r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us avg', round(float(r['TotalDurationNs'])/1e6/22,3), 'ms per step')
PY
  rm -rf $D/trace_$v
done
