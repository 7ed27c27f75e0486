# round 2, GPU call o: stand-alone kernel times (TLAMC_SERIAL=1: no expand kernel beside a materialise kernel) vs overlapped
cd /root/repo; mkdir -p gpurun_out/r02o
for v in "" "--table-log2 28"; do
  for ser in 0 1; do
    echo "== bench $v serial=$ser" >> gpurun_out/r02o/bench_ab.log
    if [ $ser = 1 ]; then export TLAMC_SERIAL=1; else unset TLAMC_SERIAL; fi
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02o/bench_ab.log 2>&1
  done
done
grep -E "^==|ms_per_step|golden" gpurun_out/r02o/bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
