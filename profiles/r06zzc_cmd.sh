# round 6, GPU call zzc: the last tree (one host-side line after call zza: pcal_codegen.cpp): the JIT / PlusCal GPU files, then the driver's command once more
cd /root/repo; D=$PWD/gpurun_out/r06zzc; mkdir -p $D
timeout 2400 python -m pytest tests/test_gpu_zz_jit.py tests/test_gpu_pcal.py -m gpu -x -q > $D/pytest_pcal.log 2>&1; tail -n 2 $D/pytest_pcal.log
( time timeout 1200 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json ) 2>&1 | grep real
python - <<'PY'
import json
d = json.load(open('/root/repo/gpurun_out/r06zzc/bench_default_line.json')); r = d['roofline']
print(round(d['ms_per_step'], 2), round(d['value'] / 1e9, 3), round(r['frac'], 4), r['traffic_source'][:30])
for k in ('config4_model_one_gpu', 'config5_model_one_gpu'):
    o = d[k]; print(k, round(o['ms_per_step'], 2), o['roofline'].get('traffic_source', '')[:34])
for o in d.get('pcal', []): print('pcal', o['workload'][:40], round(o['ms_per_step'], 2), round(o['value'] / 1e9, 3), o['state_bytes'])
PY
