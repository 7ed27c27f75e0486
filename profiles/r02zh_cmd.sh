# round 2, GPU call zh (last): bench.py with its defaults (40 steps + 3 warm-up, cpu_baseline) and the front-end GPU tests on the final library
cd /root/repo; mkdir -p gpurun_out/r02zh
timeout 200 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02zh/bench_default_line.json; cut -c1-330 gpurun_out/r02zh/bench_default_line.json
timeout 200 python -m pytest tests/test_frontend.py tests/test_gpu_paxos.py -m gpu -x -q > gpurun_out/r02zh/pytest_gpu_frontend.log 2>&1; tail -2 gpurun_out/r02zh/pytest_gpu_frontend.log
