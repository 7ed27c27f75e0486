# round 4, GPU call z: the random model x engine-settings sweep alone (new test), before it joins the suite
cd /root/repo; mkdir -p gpurun_out/r04z
timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "random_model_and_engine_settings" 2>&1 | tail -40 > gpurun_out/r04z/pytest_sweep.log; grep -v "^$" gpurun_out/r04z/pytest_sweep.log | tail -30 | cut -c1-400
