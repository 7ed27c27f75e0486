# round 2, GPU call g: dense slots inline + 5-wave variant: parity suite (raft tests) and A/B bench lines
cd /root/repo; mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02g/pytest_gpu.log
for v in "" "--no-dense" "--occ3" "--occ3 --no-dense" "--fam-blocks 2"; do
  echo "== bench $v" >> gpurun_out/r02g/bench_ab.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02g/bench_ab.log 2>&1
done
grep -E "^==|ms_per_step" gpurun_out/r02g/bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
