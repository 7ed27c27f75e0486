# round 6, GPU call zk: the driver's command on the final tree — config 4's object now times the MC_F_PARK instantiation beside the default (park_ab)
cd /root/repo; D=$PWD/gpurun_out/r06zk; mkdir -p $D
( time timeout 1200 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r06zk/bench_default_line.json')); r = d['roofline']
print(round(d['ms_per_step'], 2), round(d['value'] / 1e9, 3), r['frac'], r['traffic_source'][:60])
o = d['config4_model_one_gpu']; print('config4', round(o['ms_per_step'], 2), o['inwave_states'], 'park:', round(o['park_ab']['ms_per_step'], 2), o['park_ab']['inwave_states'], o['park_ab']['kernel_ms'])
o = d['config5_model_one_gpu']; print('config5', round(o['ms_per_step'], 2))
PY
tail -n 2 $D/bench.err
