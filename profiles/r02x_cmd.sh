# round 2, GPU call x: seen-set of any size (multiply-shift home bucket) — parity tests, then the bench at three table sizes
cd /root/repo; mkdir -p gpurun_out/r02x
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_paxos.py tests/test_gpu_sharded.py -x -q > gpurun_out/r02x/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02x/pytest_gpu.log
for o in "" "--table-log2 27" "--table-log2 28" "--table-slots 167772160"; do
  echo "== bench $o" >> gpurun_out/r02x/bench_ab.log
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline $o 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02x/bench_ab.log
done
python - <<'PY'
import json
for l in open('gpurun_out/r02x/bench_ab.log'):
    if l.startswith('=='): print(l.strip()); continue
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(round(d['ms_per_step'],2), round(d['config']['seen_set_load'],3), d['roofline']['kernel_ms'], round(d['roofline']['frac'],4))
PY
