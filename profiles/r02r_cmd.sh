# round 2, GPU call r: cross-rank counterexamples (Python side), packed rounds with the tighter capacity: sharded GPU tests + RCCL world-1 bench
cd /root/repo; mkdir -p gpurun_out/r02r
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_abi_symbols.py -x -q > gpurun_out/r02r/pytest_gpu_sharded.log 2>&1; tail -5 gpurun_out/r02r/pytest_gpu_sharded.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02r/bench_fused.json 2>&1; tail -1 gpurun_out/r02r/bench_fused.json | cut -c1-200
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r02r/bench_rccl_world1.json 2>&1; tail -1 gpurun_out/r02r/bench_rccl_world1.json | cut -c1-300
