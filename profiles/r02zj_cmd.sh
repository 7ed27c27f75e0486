# round 2, GPU call zj: the device node exists on the GPU box (binding.lib() then loads torch first) and the torch-using sharded tests pass
cd /root/repo; mkdir -p gpurun_out/r02zj
ls -la /dev/kfd > gpurun_out/r02zj/devnode.txt 2>&1; cat gpurun_out/r02zj/devnode.txt
timeout 70 python -m pytest tests/test_gpu_sharded.py -x -q > gpurun_out/r02zj/pytest_gpu_sharded.log 2>&1; tail -2 gpurun_out/r02zj/pytest_gpu_sharded.log
