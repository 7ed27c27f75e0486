# round 6, GPU call zt: generated PlusCal code with rows packed to the cells' inferred ranges — first device run: the JIT GPU tests (per-level state sets,
# traces, checkpoint, the shard gate), then the A/B against the interpreter's rows
cd /root/repo; D=$PWD/gpurun_out/r06zt; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q --durations=4 > $D/pytest_jit.log 2>&1; grep -E 'passed|failed|error|s call|Error|assert' $D/pytest_jit.log | tail -12
timeout 900 python profiles/pcal_pack_ab.py 2>$D/ab.err | tee $D/pack_ab.jsonl
grep -v amdgpu.ids $D/ab.err | tail -5
