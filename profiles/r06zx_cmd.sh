# round 6, GPU call zx: the by-pairs kernel's sort key for generated code = (instance, label) instead of the label alone (a label function is a template over the
# instance: a batch of one label and N instances ran N copies of the code): A/B through $TLAMC_JIT_DEFS in one call, then the JIT GPU tests on the new default
cd /root/repo; D=$PWD/gpurun_out/r06zx; mkdir -p $D
for rep in 1 2; do
for defs in "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1 -DMC_GEN_KEY_BY_INST=0" "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1"; do
  TLAMC_JIT_DEFS="$defs" PACK_AB_ONLY=1 timeout 600 python profiles/pcal_pack_ab.py 2>>$D/ab.err | tee -a $D/key_ab.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['model'], d.get('ms'), d.get('states_per_s_G'), d.get('defs', '')[-24:], d.get('distinct'), d.get('error', ''))"
done; done
timeout 1500 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q > $D/pytest_jit.log 2>&1; tail -n 3 $D/pytest_jit.log
grep -v amdgpu.ids $D/ab.err | tail -3
