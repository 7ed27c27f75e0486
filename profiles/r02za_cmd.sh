# round 2, GPU call za: Paxos invariants checked per stored state (CHECK_ON_EXPAND), a stored state's own violation ordered before
# the violations found while generating successors on the same level (viol_key): whole GPU suite, Paxos timing, bench sanity
cd /root/repo; mkdir -p gpurun_out/r02za
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02za/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02za/pytest_gpu.log
timeout 300 python profiles/bench_all.py "Paxos" 2>&1 | grep -v amdgpu.ids > gpurun_out/r02za/bench_paxos.jsonl; cut -c1-330 gpurun_out/r02za/bench_paxos.jsonl
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric > gpurun_out/r02za/bench_line.json; cut -c1-260 gpurun_out/r02za/bench_line.json
