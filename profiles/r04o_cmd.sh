# round 4, GPU call o: generated-only shortcuts (RequestVote of a candidate / DuplicateMessage when the bag already holds MaxMsgs copies: counted, not
# evaluated) — the whole GPU suite (every parity test compares `generated`), bench lines, phase profile
cd /root/repo; D=gpurun_out/r04o; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $D/pytest_gpu.log 2>&1; tail -n 9 $D/pytest_gpu.log
for w in t3 k10 raft5; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $w 2>$D/bench_$w.err | grep metric > $D/bench_$w.json
  python -c "
import json; d=json.load(open('$D/bench_$w.json')); r=d['roofline']; print('$w', round(d['ms_per_step'],1), {k:round(v,1) for k,v in r['kernel_ms'].items()}, r['inwave_states'], round(r['frac'],4), round(r['pipeline_frac'],4))"
done
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_t3.json'))
print(d['cycles_per_wave'], [(r['phase'][:24], r['cycles_per_wave']) for r in d['phases']])"
