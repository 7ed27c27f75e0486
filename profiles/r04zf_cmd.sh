# round 4, GPU call zf (the last seconds): 8 engines on one GPU through the torch door and the native loop with 8 ranks, on the final loop
cd /root/repo; D=gpurun_out/r04zf; mkdir -p $D
timeout 42 python -m pytest tests/test_gpu_sharded.py -q -k "eight_engines or (native_rccl_loop_with and 8)" > $D/pytest_8_ranks.log 2>&1; echo rc=$? >> $D/pytest_8_ranks.log; grep -E "^E  |^FAILED|passed|failed|rc=" $D/pytest_8_ranks.log | cut -c1-400 | tail -8
