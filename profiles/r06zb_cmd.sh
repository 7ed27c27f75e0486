# round 6, GPU call zb: the in-wave writers' overflow parked in the new-list's memory and written by the workgroup's own tail (several rounds)
# instead of through k_materialise: raft parity on the GPU first (the product library, then a stress build whose survivor lists hold 128
# entries instead of 448 — libtlamc_o128.so, TU 3 with -DMC_OCAP=128: nearly every workgroup of a 3-server model parks chunks and runs several
# rounds), then A/B of config 4's model (raft5) and the contract line (t3), each against --list-overflow (= rounds 4-5's form) inside this one call
cd /root/repo; D=$PWD/gpurun_out/r06zb; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_product.log 2>&1; tail -n 2 $D/pytest_product.log
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_o128.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "raft or config or trace or chunk or table or step or checkpoint" > $D/pytest_o128.log 2>&1; tail -n 2 $D/pytest_o128.log
for wl in raft5 t3; do for v in "" "--list-overflow" ""  "--list-overflow"; do
  timeout 600 python bench.py --workload $wl --steps 5 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline $v 2>>$D/bench.err | grep -v amdgpu.ids | V="$v" WL=$wl python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':os.environ['WL'],'variant':os.environ['V'] or 'parked','ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states'),'frac':r.get('frac')}))" | tee -a $D/ab.jsonl
done; done
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_o128.so timeout 600 python bench.py --workload t3 --steps 3 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','variant':'parked, MC_OCAP=128 (stress build, golden-gated like every line)','ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
tail -n 3 $D/bench.err
