# round 4, GPU call zg: per-rank checkpoint / recover with 2 ranks and the failing-rank test on the final loop
cd /root/repo; D=gpurun_out/r04zg; mkdir -p $D
timeout 20 python -m pytest tests/test_gpu_sharded.py -q -k "(checkpoint_per_rank and 2) or native_loop_survives" > $D/pytest_ckpt.log 2>&1; echo rc=$? >> $D/pytest_ckpt.log; grep -E "^E  |^FAILED|passed|failed|rc=" $D/pytest_ckpt.log | cut -c1-400 | tail -6
