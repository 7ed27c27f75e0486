# round 6, GPU call zf: call ze's driver line read 137.0 ms for t3 (126.1 in call z): the parked-overflow tail as a LOOP around the round had put 8
# spilled VGPRs into the writer's batch loop of the 3-server kernel.  The tail is now: first round inline (as before the change), later rounds out
# of line (tail_more_rounds) — no scratch access on the hot path again.  Three libraries in ONE call, alternating, t3 at 10 steps:
#   pre  = TU 3 from the sources of call z (commit 9c9a5dc: overflow through k_materialise)     libtlamc_pre.so
#   loop = TU 3 from the sources of call ze (the loop around the round)                          libtlamc_loop.so
#   (none) = the product library (first round inline, later rounds out of line)
# then raft parity on the product library + the MC_OCAP = 128 stress build of it, and config 4's model
cd /root/repo; D=$PWD/gpurun_out/r06zf; mkdir -p $D
B=$PWD/tla_rust_amd/_build
for rep in 1 2 3; do for v in pre product loop; do
  L=$B/libtlamc_$v.so; [ $v = product ] && L=$B/libtlamc.so
  TLAMC_LIB=$L timeout 600 python bench.py --workload t3 --steps 10 --warmup 2 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':os.environ['V'],'ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_product.log 2>&1; grep -E "passed|failed" $D/pytest_product.log | tail -n 1
TLAMC_LIB=$B/libtlamc_o128.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "raft or config or trace or chunk or table or step or checkpoint" > $D/pytest_o128.log 2>&1; grep -E "passed|failed" $D/pytest_o128.log | tail -n 1
TLAMC_LIB=$B/libtlamc_o128.so timeout 600 python bench.py --workload t3 --steps 3 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'t3','library':'MC_OCAP=128 stress build of the product sources (golden-gated)','ms_per_step':round(d['ms_per_step'],2),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
for v in "" "--list-overflow" "" "--list-overflow"; do
  timeout 600 python bench.py --workload raft5 --steps 5 --warmup 1 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline $v 2>>$D/bench.err | grep -v amdgpu.ids | V="$v" python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'raft5','variant':os.environ['V'] or 'parked','ms_per_step':round(d['ms_per_step'],2),'kernel_ms':r.get('kernel_ms'),'inwave_states':r.get('inwave_states')}))" | tee -a $D/ab.jsonl
done
tail -n 2 $D/bench.err
