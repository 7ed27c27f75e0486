# round 4, GPU call w: the PlusCal GPU tests after records (new CASES: treiber_records, ring_buffer; mc on both)
cd /root/repo; mkdir -p gpurun_out/r04w
timeout 900 python -m pytest tests/test_gpu_pcal.py -x -q 2>&1 | tail -15 > gpurun_out/r04w/pytest_gpu_pcal.log; tail -5 gpurun_out/r04w/pytest_gpu_pcal.log
