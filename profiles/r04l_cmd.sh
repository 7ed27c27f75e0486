# round 4, GPU call l: in-wave writes chosen level by level (a level that grew by more than TLAMC_INWAVE_GROWTH x goes through k_materialise): parity + A/B of the limit
cd /root/repo; D=gpurun_out/r04l; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 2 $D/pytest_gpu_parity.log
for g in 1.7 1.3 2.2 100; do
  for w in t3 raft5 k10; do
    TLAMC_INWAVE_GROWTH=$g timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $w 2>$D/bench_${w}_$g.err | grep -v amdgpu.ids > $D/bench_${w}_$g.json
    python -c "
import json; d=json.load(open('$D/bench_${w}_$g.json')); r=d['roofline']; print('$g', '$w', round(d['ms_per_step'],1), {k:round(v,1) for k,v in r['kernel_ms'].items()}, r['inwave_states'])"
  done
done
