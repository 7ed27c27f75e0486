# round 6, GPU call zh: PARK as an INSTANTIATION of the by-family kernel (k_expand_family<.., PARK = true>) that the host's level loop switches to when a
# level sent more than 1 % of its new states through the new-list; the kernel of a model whose lists hardly ever fill up (t3) is the kernel of call z
# again, byte for byte in its resources (128 VGPRs, 0 spilled, scratch 20); the later rounds run at the kernel's very end.  Parity with parking FORCED
# (TLAMC_PARK=1) on the product library and on the MC_OCAP = 128 stress build, then as shipped (adaptive); raft5: never / adaptive / from the first level;
# t3: product against the library of call z (pre)
cd /root/repo; D=$PWD/gpurun_out/r06zh; mkdir -p $D
B=$PWD/tla_rust_amd/_build
TLAMC_PARK=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_product_park1.log 2>&1; grep -E "passed|failed" $D/pytest_product_park1.log | tail -n 1
TLAMC_PARK=1 TLAMC_LIB=$B/libtlamc_o128.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "raft or config or trace or chunk or table or step or checkpoint" > $D/pytest_o128_park1.log 2>&1; grep -E "passed|failed" $D/pytest_o128_park1.log | tail -n 1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_product.log 2>&1; grep -E "passed|failed" $D/pytest_product.log | tail -n 1
for rep in 1 2; do for v in 0 adaptive 1; do
  TLAMC_PARK=$([ $v = adaptive ] && echo "" || echo $v) timeout 600 python bench.py --workload raft5 --steps 5 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | V="$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'raft5','TLAMC_PARK':os.environ['V'],'ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done; done
for rep in 1 2 3; do for v in pre product; do
  L=$B/libtlamc_$v.so; [ $v = product ] && L=$B/libtlamc.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload t3 --steps 10 --warmup 2 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':os.environ['V'],'ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done; done
TLAMC_PARK=1 TLAMC_LIB=$B/libtlamc_o128.so timeout 600 python bench.py --workload t3 --steps 3 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':'MC_OCAP=128 stress build, TLAMC_PARK=1 (golden-gated)','ms_per_step':round(d['ms_per_step'],2),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
tail -n 2 $D/bench.err
