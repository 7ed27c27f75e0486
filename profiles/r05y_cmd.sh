# round 5, GPU call y: SETS of records (the message soup; spec_vm.h VM_RSADD / VM_RSDEL / VM_RSHAS) beside the channels: the new GPU cases,
# every GPU test of the compiled-program path (the interpreter changed again), the three larger models
cd /root/repo; D=$PWD/gpurun_out/r05y; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_zz_channels.py tests/test_gpu_pcal.py tests/test_gpu_zz_ms_queue.py -m gpu -q --durations=6 > $D/pytest_gpu_pcal.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_pcal.log | tail -10; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_pcal.log | head
timeout 300 python profiles/bench_channels.py 2>$D/chan.err | tee $D/bench_channels.jsonl
