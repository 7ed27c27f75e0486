# round 5, GPU call w: the PlusCal CHANNELS (arrays of sequences, sequences of records; spec_vm.h VM_SEQSEL / VM_SEQLEN, every sequence
# instruction with a third operand): the new GPU cases, every other GPU test of the compiled-program path (the interpreter changed), and
# the 2.85 M-state two-phase commit
cd /root/repo; D=$PWD/gpurun_out/r05w; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_zz_channels.py tests/test_gpu_pcal.py tests/test_gpu_zz_ms_queue.py -m gpu -q --durations=6 > $D/pytest_gpu_pcal.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_pcal.log | tail -10; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_pcal.log | head
timeout 300 python profiles/bench_channels.py 2>$D/chan.err | tee $D/bench_channels.jsonl; tail -c 300 $D/chan.err
