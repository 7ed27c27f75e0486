# round 2, GPU call b: the one-kernel expand (k_expand_direct) on the complete-graph bench model: parity suite + A/B bench lines
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02b_pytest_gpu.log
for v in "" "--occ3" "--no-direct"; do
  echo "== bench $v" >> gpurun_out/r02b_bench_ab.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02b_bench_ab.log 2>&1
done
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r02b_bench_line.json 2> gpurun_out/r02b_bench_line.err
nproc > gpurun_out/r02b_host.txt; free -g >> gpurun_out/r02b_host.txt; (java -version 2>&1 | head -1) >> gpurun_out/r02b_host.txt
cut -c1-600 gpurun_out/r02b_bench_ab.log
