# round 2, GPU call n: k_materialise = copy the row while reading it once, then patch (apply_copy_patch)
cd /root/repo; mkdir -p gpurun_out/r02n
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r02n/pytest_gpu_parity.log 2>&1; tail -3 gpurun_out/r02n/pytest_gpu_parity.log
for v in "" "--no-dense" "--table-log2 28"; do
  echo "== bench $v" >> gpurun_out/r02n/bench_ab.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02n/bench_ab.log 2>&1
done
grep -E "^==|ms_per_step|golden" gpurun_out/r02n/bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
