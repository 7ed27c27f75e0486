# round 2, GPU call v: the whole GPU suite at the current head + smoke()
cd /root/repo; mkdir -p gpurun_out/r02v
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r02v/pytest_gpu.log 2>&1; tail -4 gpurun_out/r02v/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
