# round 5, GPU call ze: tests/test_gpu_zz_channels.py with the pagecache added (20 M states)
cd /root/repo; D=$PWD/gpurun_out/r05ze; mkdir -p $D
timeout 150 python -m pytest tests/test_gpu_zz_channels.py -m gpu -q --durations=4 > $D/pytest_gpu_zz_channels.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_zz_channels.log | tail -6; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_zz_channels.log | head
