# round 5, GPU call j: after the last host-side changes (in-wave writes on every level by default; bench.py's --exchange auto survives a
# failing trial form): the parity file, the bench / launcher tests of the sharded file, config 4's model and the contract workload
cd /root/repo; D=$PWD/gpurun_out/r05j; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -2 | tee $D/pytest_parity_ckpt.log
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "bench or launcher" 2>&1 | grep -E 'passed|failed|error' | tail -2 | tee $D/pytest_sharded_bench.log
for w in raft5 k10 t3; do timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $w 2>/dev/null | grep '"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(workload='$w', ms_per_step=round(d['ms_per_step'],2), Gstates_s=round(d['value']/1e9,3), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, inwave=r['inwave_states'])))" | tee -a $D/bench_workloads.jsonl; done
