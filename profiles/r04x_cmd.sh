# round 4, GPU call x: the whole GPU suite on the final tree (incl. the random-configuration and random-algorithm tests), smoke,
# counters of configs 4 and 5 (raft5, ssi4x3: their dominant kernels), the driver's command
cd /root/repo; D=/root/repo/gpurun_out/r04x; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $D/pytest_gpu.log; tail -3 $D/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; tail -2 $D/smoke.log
for w in raft5 ssi4x3; do
  BENCH_ARGS="--workload $w" timeout 600 bash profiles/collect.sh r04x/$w > /dev/null 2>&1
  python profiles/summarize_pmc.py $D/${w}_counters.json $D/$w/pmc_*.csv > $D/${w}_counters_summary.txt 2>&1
  cp $D/$w/kernel_stats.csv $D/${w}_kernel_stats.csv; cp $D/$w/bench_line.json $D/${w}_bench_line.json; rm -rf $D/$w
  cut -c1-200 $D/${w}_bench_line.json
done
cd /root/repo; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $D/bench_default_line.json; cut -c1-260 $D/bench_default_line.json
