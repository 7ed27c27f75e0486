# round 6, GPU call zi (the tree the round ends with: MC_F_PARK opt-in, the default kernels are those of call z again): the whole GPU suite with its slowest tests named, smoke(), rocprofv3 kernel stats + the
# separate PMC passes of the bench command for the THREE lowered workloads (t3 = the contract line, raft5 = BASELINE config 4's model, ssi4x3 =
# config 5's), each summary stamped with its spec's kernel sources and copied into profiles/ BEFORE the driver's command runs, so that the
# line's three roofline objects carry measured traffic; then the driver's command itself
cd /root/repo; D=$PWD/gpurun_out/r06zi; mkdir -p $D
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -12
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
for wl in t3 raft5 ssi4x3; do
  BENCH_ARGS="--workload $wl --no-atomic-add --no-other-configs --no-pcal" timeout 1500 bash profiles/collect.sh r06zi_$wl > $D/collect_$wl.log 2>&1
  S=$PWD/gpurun_out/r06zi_$wl
  spec=raft; [ $wl = ssi4x3 ] && spec=ssi
  PMC_SPEC=$spec python profiles/summarize_pmc.py $D/${wl}_pmc.json $S/pmc_*.csv > $D/${wl}_pmc_summary.txt 2>&1
  cp $S/kernel_stats.csv $D/${wl}_kernel_stats.csv; cp $S/bench_line.json $D/${wl}_bench_line_under_rocprof.json; rm -rf $S
  cp $D/${wl}_pmc.json profiles/r06zi_${wl}_pmc.json
  head -3 $D/${wl}_kernel_stats.csv | cut -c1-70,300-420
done
timeout 1200 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r06zi/bench_default_line.json')); r = d['roofline']
print(round(d['ms_per_step'], 2), round(d['value'] / 1e9, 3), {k: r[k] for k in ('kernel', 'frac', 'traffic', 'traffic_lower', 'l2_hit_rate', 'pipeline_frac', 'pipeline_frac_2WD', 'kernel_ms', 'valu_per_successor', 'salu_per_successor', 'traffic_source')})
for k in ('config4_model_one_gpu', 'config5_model_one_gpu'):
    o = d[k]; r = o['roofline']
    print(k, round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), {a: r[a] for a in ('kernel', 'frac', 'traffic', 'pipeline_frac', 'pipeline_frac_2WD', 'kernel_ms', 'valu_per_successor', 'salu_per_successor', 'inwave_states')})
print('atomic_add', round(d['atomic_add']['ms_per_step'], 2))
for o in d.get('pcal', []): print('pcal', o['workload'][:40], round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), o['backend'], round(o['engine_create_s'], 1))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['tlc_probe'])
PY
tail -n 3 $D/bench.err
