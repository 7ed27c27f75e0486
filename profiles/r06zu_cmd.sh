# round 6, GPU call zu: packed rows of generated PlusCal code after the compiler work-around (-fno-slp-vectorize: clang's SLP vectorizer crashed on two programs
# in call zt): the JIT GPU tests, the A/B against the interpreter's rows, what the range check costs, the by-pairs kernel's shape now that the kernels need 90 - 150 VGPRs
cd /root/repo; D=$PWD/gpurun_out/r06zu; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q --durations=4 > $D/pytest_jit.log 2>&1; grep -E 'passed|failed|error|s call|Error|assert' $D/pytest_jit.log | tail -8
timeout 900 python profiles/pcal_pack_ab.py 2>$D/ab.err | tee $D/pack_ab.jsonl
for defs in "-DMC_PAIR_MINW=2 -DMC_PAIR_WAVES=1 -DMC_GEN_NO_RANGE_CHECK" "-DMC_PAIR_MINW=4 -DMC_PAIR_WAVES=1" "-DMC_PAIR_MINW=3 -DMC_PAIR_WAVES=1" "-DMC_PAIR_MINW=4 -DMC_PAIR_WAVES=2" "-DMC_PAIR_MINW=3 -DMC_PAIR_WAVES=2"; do
  TLAMC_JIT_DEFS="$defs" PACK_AB_ONLY=1 PACK_AB_JOBS=3 timeout 600 python profiles/pcal_pack_ab.py 2>>$D/ab.err | tee -a $D/pack_shapes.jsonl | cut -c1-230
done
grep -v amdgpu.ids $D/ab.err | tail -5
