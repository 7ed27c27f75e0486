# round 2, GPU call w: the Paxos family on the GPU (tests/test_gpu_paxos.py), then the whole GPU suite
cd /root/repo; mkdir -p gpurun_out/r02w
timeout 900 python -m pytest tests/test_gpu_paxos.py -x -q > gpurun_out/r02w/pytest_gpu_paxos.log 2>&1; tail -5 gpurun_out/r02w/pytest_gpu_paxos.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02w/pytest_gpu.log 2>&1; tail -4 gpurun_out/r02w/pytest_gpu.log
