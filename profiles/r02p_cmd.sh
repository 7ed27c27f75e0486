# round 2, GPU call p: per-wavefront duplicate filter (LDS, 256 entries) in front of the seen-set, RequestVote from a per-parent digest, deadlock bits by ballot: parity + A/B
cd /root/repo; mkdir -p gpurun_out/r02p
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r02p/pytest_gpu_parity.log 2>&1; tail -3 gpurun_out/r02p/pytest_gpu_parity.log
for v in "" "--no-filter" "--table-log2 28"; do
  for ser in 0 1; do
    echo "== bench $v serial=$ser" >> gpurun_out/r02p/bench_ab.log
    if [ $ser = 1 ]; then export TLAMC_SERIAL=1; else unset TLAMC_SERIAL; fi
    timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02p/bench_ab.log 2>&1
  done
done
unset TLAMC_SERIAL
grep -E "^==|ms_per_step|golden" gpurun_out/r02p/bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
python profiles/ablate2.py > gpurun_out/r02p/ablate2.log 2>&1; grep -v amdgpu.ids gpurun_out/r02p/ablate2.log
