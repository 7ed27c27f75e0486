# round 6, GPU call d: k_expand_pairs<SpecSsi> with the cheaper fingerprint (hmum, pair base: three terms per successor instead of six
# fmix64), events packed in one word — against call b's kernel (libtlamc_r06b.so), alternating, in ONE call; the SSI parity cases on the
# new hash; the phase profile of the new kernel
cd /root/repo; D=$PWD/gpurun_out/r06d; mkdir -p $D; B=$PWD/tla_rust_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q -k "ssi or SSI or textbook or si_" > $D/pytest_gpu_ssi.log 2>&1; tail -n 2 $D/pytest_gpu_ssi.log
for v in new r06b new r06b; do
  L=$B/libtlamc.so; [ $v = r06b ] && L=$B/libtlamc_r06b.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload ssi4x3 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='$v'; print(json.dumps(d))" >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06d/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['variant'], round(d['ms_per_step'], 2), r['kernel_ms'])
PY
TLAMC_LIB=$B/libtlamc_ssiprof.so timeout 600 python profiles/phase_prof_ssi.py > $D/phase_profile_ssi4x3.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_ssi4x3.json')); print({k:v for k,v in d.items() if k!='phases'}); [print(p) for p in d['phases']]"
