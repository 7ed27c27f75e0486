# round 2, GPU call d: 8-slot bucket probes, fingerprints handed to k_materialise, NB=1 default; parity suite, bench, rocprofv3 stats + PMC
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02d_pytest_gpu.log 2>&1; tail -3 gpurun_out/r02d_pytest_gpu.log
for v in "" "--table-log2 28" "--fam-blocks 2" "--chunk 8388608"; do
  echo "== bench $v" >> gpurun_out/r02d_bench_ab.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $v >> gpurun_out/r02d_bench_ab.log 2>&1
done
grep -E "^==|ms_per_step" gpurun_out/r02d_bench_ab.log | sed -E 's/.*"ms_per_step": ([0-9.]+).*"kernel_ms": (\{[^}]*\}).*/\1 \2/'
bash profiles/collect.sh r02d > gpurun_out/r02d_collect.log 2>&1
python profiles/summarize_pmc.py gpurun_out/r02d/pmc.json gpurun_out/r02d/pmc_*.csv > gpurun_out/r02d/pmc_summary.txt 2>&1
head -30 gpurun_out/r02d/kernel_stats.csv | cut -c1-200
