# round 4, GPU call h: checkpoint of the round — whole GPU suite, smoke, the contract line (raft + the atomic_add object + cpu_baseline), every lowered workload
cd /root/repo; D=gpurun_out/r04h; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu.log 2>&1; tail -n 12 $D/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; cut -c1-400 $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); print(json.dumps(d.get('atomic_add'))[:900]); print(json.dumps(d.get('cpu_baseline'))[:300])"
timeout 900 python profiles/bench_all.py 2>&1 | grep -v amdgpu.ids > $D/bench_all_workloads.jsonl; cut -c1-220 $D/bench_all_workloads.jsonl
