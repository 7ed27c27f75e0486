# round 2, GPU call ze (the state the round ends in): whole GPU suite + smoke, the contract bench line with cpu_baseline,
# rocprofv3 kernel stats + PMC passes of the same command, every lowered workload
cd /root/repo; mkdir -p gpurun_out/r02ze
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02ze/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02ze/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02ze/smoke.log 2>&1; tail -2 gpurun_out/r02ze/smoke.log
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02ze/bench_default_line.json; cut -c1-300 gpurun_out/r02ze/bench_default_line.json
timeout 900 bash profiles/collect.sh r02ze > gpurun_out/r02ze/collect.log 2>&1
timeout 600 python profiles/bench_all.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02ze/bench_all_workloads.jsonl; cut -c1-200 gpurun_out/r02ze/bench_all_workloads.jsonl
