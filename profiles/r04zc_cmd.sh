# round 4, GPU call zc: the whole sharded GPU file on the final tree (default stay form = exact sizes; AbiOps maps the pending-list
# condition of mc_shard_expand_finish to MC_EROUTE)
cd /root/repo; D=gpurun_out/r04zc; mkdir -p $D
timeout 245 python -m pytest tests/test_gpu_sharded.py -q > $D/pytest_gpu_sharded.log 2>&1; echo rc=$? >> $D/pytest_gpu_sharded.log; grep -E "^FAILED|passed|failed|rc=" $D/pytest_gpu_sharded.log | cut -c1-300 | tail -12
