# round 4, GPU call y: after MC_EROUTE / restarts (engine.hip's host code changed: its three error returns) — the sharded GPU tests,
# the counters re-collected on the final kernel sources (stamp), the driver's command
cd /root/repo; D=/root/repo/gpurun_out/r04y; mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_checkpoint.py -x -q 2>&1 | tail -8 > $D/pytest_gpu_sharded.log; tail -2 $D/pytest_gpu_sharded.log
timeout 600 bash profiles/collect.sh r04y/c > /dev/null 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/c/pmc_*.csv > $D/pmc_summary.txt 2>&1
cp $D/c/kernel_stats.csv $D/kernel_stats.csv; cp $D/c/bench_line.json $D/bench_line.json; rm -rf $D/c
cp $D/pmc.json profiles/r04y_pmc.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' > $D/bench_default_line.json; cut -c1-220 $D/bench_default_line.json
python -c "
import json; d=json.loads(open('$D/bench_default_line.json').read()); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['traffic_source'][:50])"
