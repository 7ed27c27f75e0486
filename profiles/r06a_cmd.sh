# round 6, GPU call a: the per-kernel evidence BASELINE configs 4 and 5 never had on the round-5 kernels —
# rocprofv3 kernel stats + the separate PMC passes of `bench.py --workload ssi4x3` and `--workload raft5` on HEAD's kernels (before
# any kernel work of this round), each summary stamped with its own spec's sources
cd /root/repo; D=$PWD/gpurun_out/r06a; mkdir -p $D
for wl in ssi4x3 raft5; do
  BENCH_ARGS="--workload $wl --no-atomic-add --no-other-configs" timeout 1500 bash profiles/collect.sh r06a_$wl > $D/collect_$wl.log 2>&1
  S=$PWD/gpurun_out/r06a_$wl
  spec=raft; [ $wl = ssi4x3 ] && spec=ssi
  PMC_SPEC=$spec python profiles/summarize_pmc.py $D/${wl}_pmc.json $S/pmc_*.csv > $D/${wl}_pmc_summary.txt 2>&1
  cp $S/kernel_stats.csv $D/${wl}_kernel_stats.csv; cp $S/bench_line.json $D/${wl}_bench_line_under_rocprof.json
  rm -rf $S
  timeout 600 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>$D/bench_$wl.err | grep '"metric"' > $D/${wl}_bench_line.json
  python -c "
import json; d=json.load(open('$D/${wl}_bench_line.json')); r=d['roofline']; print('$wl', round(d['ms_per_step'],2), r['kernel'], round(r['frac'],4), r['kernel_ms'])"
  head -5 $D/${wl}_kernel_stats.csv | cut -c1-200
done
