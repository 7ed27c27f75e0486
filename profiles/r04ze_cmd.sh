# round 4, GPU call ze (what is left of the GPU budget): the tests of the exchange forms again on the final loop (segmented probes)
cd /root/repo; D=gpurun_out/r04ze; mkdir -p $D
timeout 70 python -m pytest tests/test_gpu_sharded.py -q -k "three_forms or exchange_forms or (bench_contract and 2) or stay_mode" > $D/pytest_exchange_final.log 2>&1; echo rc=$? >> $D/pytest_exchange_final.log; grep -E "^E  |^FAILED|passed|failed|rc=" $D/pytest_exchange_final.log | cut -c1-400 | tail -8
