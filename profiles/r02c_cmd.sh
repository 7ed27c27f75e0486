# round 2, GPU call c: self-loop shortcut + NB arena blocks per wavefront in the by-family kernel; last-level invariant check (SSI)
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02c_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02c_pytest_gpu.log
for v in "--fam-blocks 4" "--fam-blocks 2" "--fam-blocks 1" "--direct" "--fam-blocks 4 --chunk 2097152" "--fam-blocks 4 --chunk 8388608" "--fam-blocks 4 --table-log2 28"; do
  echo "== bench $v" >> gpurun_out/r02c_bench_ab.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02c_bench_ab.log 2>&1
done
grep -E "^==|ms_per_step" gpurun_out/r02c_bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
