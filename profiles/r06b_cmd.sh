# round 6, GPU call b: the by-pairs kernel (engine_pairs.h: k_expand_pairs<SpecSsi>) for the first time on a device — the SSI / SI parity
# cases, smoke(), and BASELINE config 5's model against the slot-by-slot kernel + k_materialise (--no-family) in ONE call
cd /root/repo; D=$PWD/gpurun_out/r06b; mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q -k "ssi or SSI or textbook or si_" --durations=5 > $D/pytest_gpu_ssi.log 2>&1; tail -n 12 $D/pytest_gpu_ssi.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
for v in "" "--no-family" "" "--no-family"; do
  timeout 600 python bench.py --workload ssi4x3 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs $v 2>>$D/bench.err | grep '"metric"' >> $D/ab.jsonl
done
python - <<'PY'
import json
for l in open('/root/repo/gpurun_out/r06b/ab.jsonl'):
    d = json.loads(l); r = d['roofline']
    print(round(d['ms_per_step'], 2), r['kernel'], r['kernel_ms'], 'inwave', r['inwave_states'])
PY
tail -n 5 $D/bench.err
