# round 2, GPU call zg: the MaxMsgKeys = 11 complete graph (3.4e8 states, golden made by the oracle on this box's host in r02zf) as a GPU test
cd /root/repo; mkdir -p gpurun_out/r02zg
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -k "next_complete or bench_workload or seen_set" > gpurun_out/r02zg/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02zg/pytest_gpu.log
