# round 2, GPU call z: sparse seen-set mode (32-byte probes when capacity >= 3 x the arena): parity suites, then atomic_add / SSI /
# Paxos / raft-5 workloads with and without it (TLAMC_DENSE_TABLE=1 forces 64-byte probes), bench.py unchanged (raft: dense mode)
cd /root/repo; mkdir -p gpurun_out/r02z
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paxos.py tests/test_gpu_pcal.py tests/test_gpu_checkpoint.py -x -q > gpurun_out/r02z/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02z/pytest_gpu.log
for w in "atomic_add" "SSI 2x3" "SSI 3x2" "Paxos 3 acc" "raft 5"; do
  timeout 300 python profiles/bench_all.py "$w" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02z/bench_sparse.jsonl
  TLAMC_DENSE_TABLE=1 timeout 300 python profiles/bench_all.py "$w" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02z/bench_dense.jsonl
done
python - <<'PY'
import json
for f in ("sparse","dense"):
    print("==",f)
    for l in open(f"gpurun_out/r02z/bench_{f}.jsonl"):
        d=json.loads(l)
        print(d['workload'][:60], d.get('ms'), d.get('distinct_per_s'), d.get('kernel_ms'), d.get('error'))
PY
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep metric | cut -c1-260
