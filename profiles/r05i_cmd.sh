# round 5, GPU call i (the tree the round ends with): the whole GPU suite with its slowest tests named, smoke(), the phase profile of the
# final kernel, config 4's model with every level written in-wave (TLAMC_INWAVE_GROWTH=10) against the default (fast-growing levels
# through k_materialise), the N = 1 point of a LAUNCHED run (torch.distributed.run: must be the fused line), the driver's command
cd /root/repo; D=$PWD/gpurun_out/r05i; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=12 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -16
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err; tail -c 200 $D/phase.err
for g in 2.3 10 2.3 10; do
  TLAMC_INWAVE_GROWTH=$g timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add --workload raft5 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(inwave_growth=$g, workload='raft5', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, inwave=r['inwave_states'])))" | tee -a $D/raft5_inwave_growth.jsonl
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add 2>/dev/null | grep '"metric"' > $D/bench_launched_world1.json; python -c "
import json; d=json.load(open('$D/bench_launched_world1.json')); print('launched N=1:', round(d['ms_per_step'],2), d['roofline']['kernel'], 'xgmi' in d)"
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), {k: r[k] for k in ('frac','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','frac_of_request_ceiling','traffic_source')}); print(json.dumps(d.get('atomic_add'))[:260])"
