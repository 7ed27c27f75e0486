# round 5, GPU call zd: tests/test_gpu_zz_channels.py again with the radix tree added (3.4 M states)
cd /root/repo; D=$PWD/gpurun_out/r05zd; mkdir -p $D
timeout 120 python -m pytest tests/test_gpu_zz_channels.py -m gpu -q --durations=4 > $D/pytest_gpu_zz_channels.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_zz_channels.log | tail -6; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_zz_channels.log | head
