# round 2, GPU call s: sharded GPU tests (cross-rank counterexamples, packed rounds, front door)
cd /root/repo; mkdir -p gpurun_out/r02s
timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q -k "invariant_counterexample or eight or replicated or front_door or gpus_option" > gpurun_out/r02s/pytest_gpu_sharded2.log 2>&1; tail -5 gpurun_out/r02s/pytest_gpu_sharded2.log
