# round 2, GPU call zb: mc_engine_step (incremental search in place) + checkpoint suite
cd /root/repo; mkdir -p gpurun_out/r02zb
timeout 900 python -m pytest tests/test_gpu_checkpoint.py -x -q > gpurun_out/r02zb/pytest_gpu_checkpoint.log 2>&1; tail -12 gpurun_out/r02zb/pytest_gpu_checkpoint.log
