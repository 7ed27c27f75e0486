for w in t3 k10; do
TLAMC_SERIAL=1 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r03n_${w}_serial.json
python - <<PY
import json
for l in open('gpurun_out/r03n_${w}_serial.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$w', d['ms_per_step'], d['roofline']['kernel_ms'])
PY
done
