# round 2, GPU call y (final state of the round): whole GPU suite + smoke, the contract bench line (with cpu_baseline), chunk-size
# A/B, the torchrun world-1 path, rocprofv3 kernel stats + PMC passes of the same command (profiles/collect.sh), every lowered workload
cd /root/repo; mkdir -p gpurun_out/r02y
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02y/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02y/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02y/smoke.log 2>&1; tail -2 gpurun_out/r02y/smoke.log
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02y/bench_line.json; cut -c1-400 gpurun_out/r02y/bench_line.json
for o in "--chunk 2097152" "--chunk 8388608"; do
  echo "== bench $o" >> gpurun_out/r02y/bench_ab.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $o 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02y/bench_ab.log
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 2>&1 | grep "\"metric\"" > gpurun_out/r02y/bench_rccl_world1.json; cut -c1-300 gpurun_out/r02y/bench_rccl_world1.json
timeout 900 bash profiles/collect.sh r02y > gpurun_out/r02y/collect.log 2>&1
timeout 600 python profiles/bench_all.py > gpurun_out/r02y/bench_all_workloads.jsonl 2>&1; cut -c1-220 gpurun_out/r02y/bench_all_workloads.jsonl
