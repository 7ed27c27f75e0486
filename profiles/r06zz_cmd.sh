# round 6, GPU call zz: survivors compacted before pass 2 of the by-pairs kernel (engine_pairs.h): A/B on generated code through $TLAMC_JIT_DEFS, the SSI + JIT + PlusCal GPU tests,
# config 5's line (bench.py --workload ssi4x3) on the new kernel
cd /root/repo; D=$PWD/gpurun_out/r06zz; mkdir -p $D
for rep in 1 2; do
for defs in "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1 -DMC_PAIR_COMPACT=0" "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1"; do
  TLAMC_JIT_DEFS="$defs" PACK_AB_ONLY=1 timeout 600 python profiles/pcal_pack_ab.py 2>>$D/ab.err | tee -a $D/compact_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['model'], d.get('ms'), d.get('states_per_s_G'), d.get('defs', '')[-22:], d.get('distinct'), d.get('error', ''))"
done; done
timeout 2400 python -m pytest tests/test_gpu_zz_jit.py tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_pairs.log 2>&1; tail -n 3 $D/pytest_pairs.log
timeout 600 python bench.py --workload ssi4x3 --steps 10 2>$D/ssi.err | grep -v amdgpu.ids > $D/ssi4x3_line.json; python -c "
import json; d = json.load(open('$D/ssi4x3_line.json')); print('ssi4x3', round(d['ms_per_step'], 2), d['roofline'].get('kernel_ms'), d['roofline'].get('traffic'))"
grep -v amdgpu.ids $D/ab.err | tail -3
