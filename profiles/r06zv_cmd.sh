# round 6, GPU call zv: the tree with packed rows of generated code: the whole GPU suite (incl. mc -checkpoint / -recover across the two layouts), smoke, the driver's command
cd /root/repo; D=$PWD/gpurun_out/r06zv; mkdir -p $D
timeout 3000 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_full.log | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; echo smoke rc $?; tail -n 2 $D/smoke.log
( time timeout 1200 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r06zv/bench_default_line.json')); r = d['roofline']
print(round(d['ms_per_step'], 2), round(d['value'] / 1e9, 3), round(r['frac'], 4), r['traffic_source'][:40], r['kernel_ms'])
o = d['config4_model_one_gpu']; print('config4', round(o['ms_per_step'], 2), o['inwave_states'])
o = d['config5_model_one_gpu']; print('config5', round(o['ms_per_step'], 2), o['roofline'].get('traffic_source', '')[:50]); print('atomic_add', round(d['atomic_add']['ms_per_step'], 2))
for o in d.get('pcal', []): print('pcal', o['workload'][:40], round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), o['state_bytes'], o['state_bytes_interpreter'])
PY
tail -n 2 $D/bench.err
