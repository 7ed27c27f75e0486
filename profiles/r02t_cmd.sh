# round 2, GPU call t: hip-rccl back-end behind the C ABI (mc_comm_* / mc_shard_run; `mc X.tla -gpus 1` native, world 1 over RCCL) + sharded tests
cd /root/repo; mkdir -p gpurun_out/r02t
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_abi_symbols.py -x -q -k "gpus_option or abi or front_door or stay_mode" > gpurun_out/r02t/pytest.log 2>&1; tail -5 gpurun_out/r02t/pytest.log
( time ./tla_rust_amd/_build/mc specs/MCraft.tla -config specs/MCraft.cfg -gpus 1 -tablelog2 27 -arena 104000000 -chunk 2097152 ) > gpurun_out/r02t/mc_gpus1_native_bench_model.log 2>&1; tail -12 gpurun_out/r02t/mc_gpus1_native_bench_model.log
