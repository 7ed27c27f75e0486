# per-phase cycle profile of k_expand_family (library built with TLAMC_PHASE_PROF=1) on the K=10 graph, sparse table
cp tla_rust_amd/_build/libtlamc.so /tmp/keep.so
cp tla_rust_amd/_build/libtlamc_pp.so tla_rust_amd/_build/libtlamc.so
python profiles/phase_prof.py 10 > gpurun_out/r03l_phase_profile_k10.json 2> gpurun_out/phase.err
cp /tmp/keep.so tla_rust_amd/_build/libtlamc.so
tail -n 3 gpurun_out/phase.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03l_phase_profile_k10.json'))
print({k:v for k,v in d.items() if k!='phases'})
for r in d['phases']: print(r)
PY
