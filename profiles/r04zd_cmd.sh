# round 4, GPU call zd (the last 1.8 GPU-minutes): the 8-rank contract invocation after the exact stay rounds probe their sources in
# rotating segments (r04zc: the lower ranks won the ties, 17.3 M states on rank 0 against 11.1 M, 9 stay levels)
cd /root/repo; D=gpurun_out/r04zd; mkdir -p $D
timeout 95 python -m pytest tests/test_gpu_sharded.py -q -k "bench_contract and 8" > $D/pytest_bench_contract_8.log 2>&1; echo rc=$? >> $D/pytest_bench_contract_8.log; grep -E "^E  |passed|failed|rc=" $D/pytest_bench_contract_8.log | cut -c1-400 | tail -8
