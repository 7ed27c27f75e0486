# round 4, GPU call p: tail with two barriers (class counts taken before the first, in the shadow of the wait): parity subset, bench lines twice, phase profile
cd /root/repo; D=gpurun_out/r04p; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 2 $D/pytest_gpu_parity.log
for rep in 1 2; do for w in t3 k10; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $w 2>$D/bench_$w.err | grep metric > $D/bench_${w}_$rep.json
  python -c "
import json; d=json.load(open('$D/bench_${w}_$rep.json')); r=d['roofline']; print('$w', round(d['ms_per_step'],1), {k:round(v,1) for k,v in r['kernel_ms'].items()}, r['inwave_states'])"
done; done
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_t3.json'))
print(d['cycles_per_wave'], [(r['phase'][:24], r['cycles_per_wave']) for r in d['phases']])"
