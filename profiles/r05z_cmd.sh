# round 5, GPU call z (the tree the round ends with: channels and sets of records in the PlusCal front-end, the interpreter with five more
# instructions; the raft / SSI / Paxos kernels are those of r05u — same kernel-source stamp): the whole GPU suite as the driver runs it,
# smoke(), and the driver's bench command
cd /root/repo; D=$PWD/gpurun_out/r05z; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_full.log | tail -10; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_full.log | head
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), {k: r[k] for k in ('frac','pipeline_frac','traffic_source','launches')})"
