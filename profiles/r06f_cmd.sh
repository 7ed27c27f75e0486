# round 6, GPU call f: why did the rotated slot order of the seen-set cost the SSI kernel 2 ms when it gained the raft kernel 6?  ssi4x3 with
# the rotated order against the first-empty-slot order, both read-first, alternating in ONE call, + the phase profile of each
cd /root/repo; D=$PWD/gpurun_out/r06f; mkdir -p $D; B=$PWD/tla_rust_amd/_build
for v in noblind norot5 noblind norot5 noblind norot5; do
  TLAMC_LIB=$B/libtlamc_$v.so timeout 600 python bench.py --workload ssi4x3 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='ssi4x3 $v'; print(json.dumps(d))" >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06f/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['variant'], round(d['ms_per_step'], 2), r['kernel_ms'])
PY
for v in rotprof norotprof; do
TLAMC_LIB=$B/libtlamc_$v.so timeout 600 python profiles/phase_prof_ssi.py > $D/phase_profile_$v.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_$v.json')); print('$v', {k:v for k,v in d.items() if k!='phases'}); [print(p) for p in d['phases']]"
done
