# A/B on the t3 workload: seen-set bucket width (64 / 32 / 16 bytes per probe) and table sparsity
set -x
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_s4_24.json 2> gpurun_out/ab.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --table-slots $((40<<26)) > gpurun_out/ab_s4_40.json 2>> gpurun_out/ab.err
cp tla_rust_amd/_build/libtlamc.so /tmp/keep.so
cp tla_rust_amd/_build/libtlamc_s2.so tla_rust_amd/_build/libtlamc.so
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --table-slots $((40<<26)) > gpurun_out/ab_s2_40.json 2>> gpurun_out/ab.err
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --table-slots $((64<<26)) > gpurun_out/ab_s2_64.json 2>> gpurun_out/ab.err
cp /tmp/keep.so tla_rust_amd/_build/libtlamc.so
python - <<'PY'
import json
for f in ['ab_s4_24','ab_s4_40','ab_s2_40','ab_s2_64']:
    for l in open('gpurun_out/'+f+'.json'):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config'].get('verdict'), d['config'].get('seen_set_load'))
PY
