# round 6, GPU call zw: the fingerprint of packed rows as a SUM of one hmum term per stored word + one fmix64 (MC_GEN_FP_SUM) against the chain of fmix64 rounds:
# A/B through $TLAMC_JIT_DEFS in one call, then the JIT GPU tests (state sets against the interpreter) with the summed form
cd /root/repo; D=$PWD/gpurun_out/r06zw; mkdir -p $D
for rep in 1 2; do
for defs in "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1" "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1 -DMC_GEN_FP_SUM=1"; do
  TLAMC_JIT_DEFS="$defs" PACK_AB_ONLY=1 timeout 600 python profiles/pcal_pack_ab.py 2>>$D/ab.err | tee -a $D/fp_sum_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['model'], d.get('ms'), d.get('states_per_s_G'), d.get('defs', '')[-20:], d.get('distinct'), d.get('error', ''))"
done; done
TLAMC_JIT_DEFS="-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1 -DMC_GEN_FP_SUM=1" timeout 1500 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q > $D/pytest_jit_fp_sum.log 2>&1; tail -n 3 $D/pytest_jit_fp_sum.log
grep -v amdgpu.ids $D/ab.err | tail -3
