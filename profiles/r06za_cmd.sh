# round 6, GPU call za: the whole GPU suite as the driver runs it, after the launcher tests learnt to retry on EADDRINUSE (call z stopped at the
# first port collision, 366 tests in)
cd /root/repo; D=$PWD/gpurun_out/r06za; mkdir -p $D
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -12
