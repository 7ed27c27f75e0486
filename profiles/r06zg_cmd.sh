# round 6, GPU call zg: the later rounds of the tail (tail_more_rounds) through address-space-3 pointers (ds_ instead of flat accesses to LDS) — call zf
# had the out-of-line rounds through generic pointers: t3 back at the level of call z (product 126.5 / 130.5 / 126.9 against pre 127.5 / 126.3 / 130.3,
# the loop form 137.0 / 133.0 / 137.0), but config 4's model no faster than --list-overflow any more (163.9 against 163.7).  Parity (product + MC_OCAP = 128
# stress build), then raft5 parked against --list-overflow x 3, t3 product against pre x 2
cd /root/repo; D=$PWD/gpurun_out/r06zg; mkdir -p $D
B=$PWD/tla_rust_amd/_build
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_product.log 2>&1; grep -E "passed|failed" $D/pytest_product.log | tail -n 1
TLAMC_LIB=$B/libtlamc_o128.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "raft or config or trace or chunk or table or step or checkpoint" > $D/pytest_o128.log 2>&1; grep -E "passed|failed" $D/pytest_o128.log | tail -n 1
for v in "" "--list-overflow" "" "--list-overflow" "" "--list-overflow"; do
  timeout 600 python bench.py --workload raft5 --steps 5 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline $v 2>>$D/bench.err | grep -v amdgpu.ids | V="$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'raft5','variant':os.environ['V'] or 'parked','ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done
for rep in 1 2; do for v in pre product; do
  L=$B/libtlamc_$v.so; [ $v = product ] && L=$B/libtlamc.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload t3 --steps 10 --warmup 2 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':os.environ['V'],'ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done; done
TLAMC_LIB=$B/libtlamc_o128.so timeout 600 python bench.py --workload t3 --steps 3 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':'MC_OCAP=128 stress build (golden-gated)','ms_per_step':round(d['ms_per_step'],2),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
tail -n 2 $D/bench.err
