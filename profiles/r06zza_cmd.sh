# round 6, GPU call zza (the tree the round ends with: packed rows of generated code, pairs sorted by (instance, label), survivors compacted before pass 2 of the by-pairs
# kernel): the whole GPU suite, smoke(), rocprofv3 kernel stats + the separate PMC passes for config 5's model (engine_pairs.h changed: the SSI stamp moved; the raft kernels'
# sources and stamps stand), copied into profiles/ BEFORE the driver's command runs, then the driver's command
cd /root/repo; D=$PWD/gpurun_out/r06zza; mkdir -p $D
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_full.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
for wl in ssi4x3; do
  BENCH_ARGS="--workload $wl --no-atomic-add --no-other-configs --no-pcal" timeout 1500 bash profiles/collect.sh r06zza_$wl > $D/collect_$wl.log 2>&1
  S=$PWD/gpurun_out/r06zza_$wl
  PMC_SPEC=ssi python profiles/summarize_pmc.py $D/${wl}_pmc.json $S/pmc_*.csv > $D/${wl}_pmc_summary.txt 2>&1
  cp $S/kernel_stats.csv $D/${wl}_kernel_stats.csv; cp $S/bench_line.json $D/${wl}_bench_line_under_rocprof.json; rm -rf $S
  cp $D/${wl}_pmc.json profiles/r06zza_${wl}_pmc.json
  head -3 $D/${wl}_kernel_stats.csv | cut -c1-70,300-420
done
( time timeout 1200 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r06zza/bench_default_line.json')); r = d['roofline']
print(round(d['ms_per_step'], 2), round(d['value'] / 1e9, 3), {k: r.get(k) for k in ('kernel', 'frac', 'traffic', 'traffic_source', 'kernel_ms')})
for k in ('config4_model_one_gpu', 'config5_model_one_gpu'):
    o = d[k]; r = o['roofline']
    print(k, round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), {a: r.get(a) for a in ('kernel', 'frac', 'traffic', 'traffic_source', 'valu_per_successor', 'salu_per_successor')})
print('atomic_add', round(d['atomic_add']['ms_per_step'], 2))
for o in d.get('pcal', []): print('pcal', o['workload'][:40], round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), o['state_bytes'], o['state_bytes_interpreter'], round(o['engine_create_s'], 1))
PY
tail -n 3 $D/bench.err
