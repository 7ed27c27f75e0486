# round 4, GPU call g: the lean in-wave writer (row copy + action + patch, starting from the parent Summary in LDS): parity, bench A/B, phase profile
cd /root/repo; D=gpurun_out/r04g; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 2 $D/pytest_gpu_parity.log
for f in "" "--wave-tail"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add $f 2>$D/bench$f.err | grep -v amdgpu.ids > $D/bench$f.json; cut -c1-260 $D/bench$f.json
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload k10 $f 2>>$D/bench$f.err | grep -v amdgpu.ids > $D/bench_k10$f.json; cut -c1-200 $D/bench_k10$f.json
done
for fl in 0 131072; do
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 $fl > $D/phase_profile_t3_$fl.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_t3_$fl.json'))
print($fl, d['cycles_per_wave'], round(d['expand_ms'],1), [(r['phase'][:22], r['cycles_per_wave']) for r in d['phases']])"
done
