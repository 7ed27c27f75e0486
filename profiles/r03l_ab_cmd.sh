# A/B on the t3 and K = 10 workloads: pipelined seen-set probes in the by-family kernel (second library built with -DMC_PROBE_PIPELINE=0)
set -x
for w in t3 k10; do python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${w}_pipe.json 2> gpurun_out/ab.err; done
cp tla_rust_amd/_build/libtlamc.so /tmp/keep.so
cp tla_rust_amd/_build/libtlamc_np.so tla_rust_amd/_build/libtlamc.so
for w in t3 k10; do python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${w}_nopipe.json 2>> gpurun_out/ab.err; done
cp /tmp/keep.so tla_rust_amd/_build/libtlamc.so
tail -n 3 gpurun_out/ab.err
python - <<'PY'
import json
for f in ['ab_t3_pipe','ab_t3_nopipe','ab_k10_pipe','ab_k10_nopipe']:
    for l in open('gpurun_out/'+f+'.json'):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('verdict'))
PY
