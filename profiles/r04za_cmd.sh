# round 4, GPU call za (the round's last 6 GPU-minutes): the 48 seeds of the random model x engine-settings sweep that the suite runs
cd /root/repo; mkdir -p gpurun_out/r04za
timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "random_model_and_engine_settings" > gpurun_out/r04za/pytest_sweep_48.log 2>&1; echo rc=$? >> gpurun_out/r04za/pytest_sweep_48.log; tail -15 gpurun_out/r04za/pytest_sweep_48.log | cut -c1-600
