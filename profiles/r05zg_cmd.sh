# round 5, GPU call zg: tests/test_gpu_zz_channels.py once more — the front-end now compiles CASE / DOMAIN / set filters / CHOOSE over sets /
# LET, and Paxos counts its quorums through a set filter (host-side changes only; the interpreter is r05y's)
cd /root/repo; D=$PWD/gpurun_out/r05zg; mkdir -p $D
timeout 120 python -m pytest tests/test_gpu_zz_channels.py -m gpu -q --durations=3 > $D/pytest_gpu_zz_channels.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_zz_channels.log | tail -5; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_zz_channels.log | head
