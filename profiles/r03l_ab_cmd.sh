# A/B on the t3 workload: frontier states per launch (chunk)
set -x
for c in 21 22 23; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --chunk $((1<<c)) > gpurun_out/ab_chunk$c.json 2> gpurun_out/ab.err
done
tail -n 3 gpurun_out/ab.err
python - <<'PY'
import json
for f in ['ab_chunk21','ab_chunk22','ab_chunk23']:
    for l in open('gpurun_out/'+f+'.json'):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'], d['config'].get('verdict'))
PY
