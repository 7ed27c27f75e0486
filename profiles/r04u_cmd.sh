# round 4, GPU call u (last): kernel sources final (engine.hip's header comment rewritten, atomic_add filter 256) -> the PMC summary re-stamped:
# parity subset, smoke, rocprofv3 kernel stats + separate PMC passes of the bench command, the contract line with that summary
cd /root/repo; D=gpurun_out/r04u; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pcal.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_gpu_subset.log 2>&1; tail -n 2 $D/pytest_gpu_subset.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
BENCH_ARGS="--no-atomic-add" timeout 1200 bash profiles/collect.sh r04u > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1; cp $D/pmc.json profiles/r04u_pmc.json
timeout 900 python bench.py 2>$D/bench.err | grep metric > $D/bench_default_line.json; cut -c1-330 $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print({k: r[k] for k in ('frac','traffic','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','avg_launch_ms','traffic_source')}); print(json.dumps(d['atomic_add'])[:300])"
head -3 $D/kernel_stats.csv | cut -c1-60,330-420
