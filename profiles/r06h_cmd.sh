# round 6, GPU call h: (1) k_expand_pairs<SpecSsi> with the THREE-STATION probe pipeline (bucket reads and compare-and-swaps both left in flight
# across an evaluation) against probing where evaluated (-DMC_PAIR_PIPE=0), 3 x 30 steps alternating; SSI parity; phase profile.
# (2) raft t3: what the in-wave writer's time is made of — one level (76.6 M parents) timed with ablation bits (profiles/tail_ablate.py)
cd /root/repo; D=$PWD/gpurun_out/r06h; mkdir -p $D; B=$PWD/tla_rust_amd/_build
timeout 900 python -m pytest tests -m gpu -x -q -k "ssi or SSI or textbook or si_" > $D/pytest_gpu_ssi.log 2>&1; tail -n 2 $D/pytest_gpu_ssi.log
for v in new nopipe new nopipe new nopipe; do
  L=$B/libtlamc.so; [ $v != new ] && L=$B/libtlamc_$v.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload ssi4x3 --steps 30 --warmup 3 --no-cpu-baseline --no-atomic-add --no-other-configs 2>>$D/bench.err | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); d['variant']='ssi4x3 $v'; print(json.dumps(d))" >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06h/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(d['variant'], round(d['ms_per_step'], 2), r['kernel_ms'])
PY
TLAMC_LIB=$B/libtlamc_pipeprof.so timeout 600 python profiles/phase_prof_ssi.py > $D/phase_profile_pipe2.json 2>$D/phase.err; python -c "
import json; d=json.load(open('$D/phase_profile_pipe2.json')); print({k:v for k,v in d.items() if k!='phases'}); [print(p) for p in d['phases']]"
TLAMC_LIB=$B/libtlamc_tailabl.so timeout 900 python profiles/tail_ablate.py 21 3 > $D/tail_ablate_t3.jsonl 2>$D/tail.err; cat $D/tail_ablate_t3.jsonl | cut -c1-200; tail -3 $D/tail.err
