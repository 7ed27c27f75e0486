# round 6, GPU call zc: the whole GPU suite on the tree with the parked in-wave overflow, the oracle-made PlusCal goldens, the 8-rank deep command
# lines at a reduced budget (tests/test_gpu_sharded.py) and the recursion stack reported as MC_EOVERFLOW
cd /root/repo; D=$PWD/gpurun_out/r06zc; mkdir -p $D
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -12
