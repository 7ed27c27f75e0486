# round 4, GPU call r: the new tests on the GPU — PlusCal modules with procedures (compiled programs vs the evaluator, mc on them), the driver's torchrun launch of bench.py
cd /root/repo; D=gpurun_out/r04r; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_pcal.py tests/test_gpu_sharded.py -m gpu -x -q -k "proc_ or treiber_procs or drivers_launcher or module_with_procedures or contract_invocation" --durations=5 > $D/pytest_new.log 2>&1; tail -n 12 $D/pytest_new.log
