# round 6, GPU call i: SSI — invariants of a stored state from what its last step can have changed (parent_status_step) and Commit as a pair
# family of its own, against (a) every invariant on every state (-DMC_SSI_STEP_STATUS=0) and (b) Commit riding with Begin / Abort
# (-DMC_SSI_COMMIT_FAMILY=0); 3 x 30 steps alternating in ONE call; the SSI / SI parity cases (violating models included) on the product
cd /root/repo; D=$PWD/gpurun_out/r06i; mkdir -p $D; B=$PWD/tla_rust_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q -k "ssi or SSI or textbook or si_" > $D/pytest_gpu_ssi.log 2>&1; tail -n 2 $D/pytest_gpu_ssi.log
for v in new fullstatus fam3 new fullstatus fam3 new fullstatus fam3; do
  L=$B/libtlamc.so; [ $v != new ] && L=$B/libtlamc_$v.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload ssi4x3 --steps 30 --warmup 3 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='ssi4x3 $v'; print(json.dumps(d))" >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06i/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['variant'], round(d['ms_per_step'], 2), r['kernel_ms'])
PY
