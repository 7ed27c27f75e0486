# round 5, GPU call s: the driver's command with the one-GPU objects of BASELINE configs 4 and 5 in its line (golden-gated), timed
cd /root/repo; D=$PWD/gpurun_out/r05s; mkdir -p $D
T0=$(date +%s.%N); timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; echo "wall seconds of the default command: $(echo "$(date +%s.%N) - $T0" | bc)"
python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), round(r['frac'],4), round(r['pipeline_frac'],4))
for k in ('atomic_add','config4_model_one_gpu','config5_model_one_gpu'):
    o=d.get(k); print(k, None if o is None else {x: (round(o[x],3) if isinstance(o[x],float) else o[x]) for x in ('ms_per_step','value','distinct','generated','depth','verdict','pipeline_frac','kernel_ms','inwave_states') if x in o})
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores_used'])"
