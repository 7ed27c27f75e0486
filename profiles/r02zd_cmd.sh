# round 2, GPU call zd: slot-sliced launches of the generic expand kernel (small frontiers of specs with many slots per state):
# whole GPU suite, then Paxos / compiled PlusCal / atomic_add timings with and without (TLAMC_NOSLICE=1)
cd /root/repo; mkdir -p gpurun_out/r02zd
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02zd/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02zd/pytest_gpu.log
for e in "" "TLAMC_NOSLICE=1"; do
  echo "== ${e:-slices}" >> gpurun_out/r02zd/bench_ab.log
  env $e timeout 300 python profiles/bench_all.py "Paxos" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02zd/bench_ab.log
  env $e timeout 300 python profiles/bench_all.py "atomic_add N=24" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02zd/bench_ab.log
  env $e timeout 300 python profiles/bench_pcal.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02zd/bench_ab.log
done
python - <<'PY'
import json
for l in open('gpurun_out/r02zd/bench_ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l[:160].rstrip()); continue
    print('  ', (d.get('workload') or d.get('name') or '')[:60], d.get('ms'), d.get('distinct'), d.get('error'))
PY
