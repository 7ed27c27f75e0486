# round 5, GPU call za: the compiled-program path after the host-side changes of the round's last hours (record parameters, `with v = r`, small
# constant intervals unrolled, CHECK_DEADLOCK, field types in dependency order) and the Paxos model
cd /root/repo; D=$PWD/gpurun_out/r05za; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_zz_channels.py tests/test_gpu_pcal.py tests/test_gpu_zz_ms_queue.py -m gpu -q --durations=6 > $D/pytest_gpu_pcal.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_pcal.log | tail -10; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_pcal.log | head
