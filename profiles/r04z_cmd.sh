# round 4, GPU call z: the random model x engine-settings sweep (new test) — 48 seeds alone first, then 400 seeds once; the random
# PlusCal algorithms and the random lowering configurations with more seeds than the suite runs
cd /root/repo; mkdir -p gpurun_out/r04z
TLAMC_SWEEP=400 timeout 500 python -m pytest tests/test_gpu_parity.py -q -k "random_model_and_engine_settings" 2>&1 | tail -40 > gpurun_out/r04z/pytest_sweep_400.log; grep -v "^$" gpurun_out/r04z/pytest_sweep_400.log | tail -30 | cut -c1-600
