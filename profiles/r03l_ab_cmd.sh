# A/B on the t3 and k10 workloads: XCD-aware tile order of the by-family expand kernel
set -x
for w in t3 k10; do
python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${w}_xcd.json 2> gpurun_out/ab.err
python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-xcd > gpurun_out/ab_${w}_noxcd.json 2>> gpurun_out/ab.err
done
tail -n 3 gpurun_out/ab.err
python - <<'PY'
import json
for f in ['ab_t3_xcd','ab_t3_noxcd','ab_k10_xcd','ab_k10_noxcd']:
    for l in open('gpurun_out/'+f+'.json'):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('verdict'), d['config'].get('seen_set_load'))
PY
