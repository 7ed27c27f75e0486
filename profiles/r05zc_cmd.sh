# round 5, GPU call zc (the last GPU seconds of the round): tests/test_gpu_zz_channels.py on the last tree — the channel / soup / Paxos cases
# again after `with` copies only the fields it reads, and the two specs added after r05zb (epoch_gc, io_buffer)
cd /root/repo; D=$PWD/gpurun_out/r05zc; mkdir -p $D
timeout 120 python -m pytest tests/test_gpu_zz_channels.py -m gpu -q --durations=4 > $D/pytest_gpu_zz_channels.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_zz_channels.log | tail -6; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_zz_channels.log | head
