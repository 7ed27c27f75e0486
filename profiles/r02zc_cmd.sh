# round 2, GPU call zc: compiler scheduling strategies for the raft-3 kernels (same source, engine.hip -DMC_TU=3 recompiled with
# -mllvm <flag>), A/B on the bench; each variant must still reproduce the golden graph (bench.py refuses otherwise)
cd /root/repo; mkdir -p gpurun_out/r02zc
cp tla_rust_amd/_build/libtlamc.so /tmp/libtlamc_default.so
for v in default ilp clause prio relax default; do
  [ $v = default ] && cp /tmp/libtlamc_default.so tla_rust_amd/_build/libtlamc.so || cp tla_rust_amd/_build/variants/libtlamc_$v.so tla_rust_amd/_build/libtlamc.so
  echo "== $v" >> gpurun_out/r02zc/bench_ab.log
  for s in 0 1; do TLAMC_SERIAL=$s timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric >> gpurun_out/r02zc/bench_ab.log; done
done
cp /tmp/libtlamc_default.so tla_rust_amd/_build/libtlamc.so
python - <<'PY'
import json
for l in open('gpurun_out/r02zc/bench_ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    d=json.loads(l); print('  ', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['roofline']['kernel_ms'].items()})
PY
