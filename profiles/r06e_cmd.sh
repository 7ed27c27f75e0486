# round 6, GPU call e: the seen-set's ROTATED slot order (every kernel) and the blind first compare-and-swap (k_expand_pairs<SpecSsi>):
# whole-table parity (raft / SSI / atomic_add / pcal parity cases), then A/B in ONE call — ssi4x3: blind against read-first (both rotated);
# t3: rotated against rounds 2-5's first-empty-slot order
cd /root/repo; D=$PWD/gpurun_out/r06e; mkdir -p $D; B=$PWD/tla_rust_amd/_build
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py tests/test_gpu_sharded.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 3 $D/pytest_gpu_parity.log
for v in new noblind new noblind; do
  L=$B/libtlamc.so; [ $v != new ] && L=$B/libtlamc_$v.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload ssi4x3 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='ssi4x3 $v'; print(json.dumps(d))" >> $D/ab.jsonl
done
for v in new norot new norot; do
  L=$B/libtlamc.so; [ $v != new ] && L=$B/libtlamc_$v.so
  TLAMC_LIB=$L timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='t3 $v'; print(json.dumps(d))" >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06e/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['variant'], round(d['ms_per_step'], 2), r['kernel_ms'])
PY
