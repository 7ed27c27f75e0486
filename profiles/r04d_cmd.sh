# round 4, GPU call d: parent rows prefetched in one round trip by the writers (apply_copy_patch) — parity subset, bench lines, phase profile
cd /root/repo; D=gpurun_out/r04d; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 3 $D/pytest_gpu_parity.log
for f in "" "--no-inwave"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $f 2>$D/bench$f.err | grep -v amdgpu.ids > $D/bench$f.json; cut -c1-300 $D/bench$f.json
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload k10 $f 2>>$D/bench$f.err | grep -v amdgpu.ids > $D/bench_k10$f.json; cut -c1-200 $D/bench_k10$f.json
done
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 > $D/phase_profile_t3.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_t3.json'))
print(d['cycles_per_wave'], d['expand_ms'], [(r['phase'], r['share']) for r in d['phases']])"
export TLAMC_RCCL=$(python -c "import sys; sys.path.insert(0,'tests'); import helpers; print(helpers.build_fakerccl())")
timeout 900 python bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --workload ssi4x3 2>$D/bench_share2_ssi4x3.err | grep -v amdgpu.ids > $D/bench_share2_ssi4x3.json; cut -c1-300 $D/bench_share2_ssi4x3.json; tail -n 3 $D/bench_share2_ssi4x3.err
