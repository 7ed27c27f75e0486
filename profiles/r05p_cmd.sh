# round 5, GPU call p: the early seen-set (a fresh run starts on a 512 MB table while the 21.5 GB one is cleared on its own stream): its GPU
# tests, the raft / engine-op GPU cases, then A/B against TLAMC_EARLY_TABLE=0 on t3 / k10 / raft5, 20 steps each, alternating
cd /root/repo; D=$PWD/gpurun_out/r05p; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_early_table.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 > $D/pytest_gpu_early.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_early.log | tail -8
run() { # tag workload steps
  local out; out=$(timeout 300 python bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-atomic-add --workload $2 2>$D/err_$1_$2.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"variant\": \"$1\", \"workload\": \"$2\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$2.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(variant='$1', workload='$2', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for rep in 1 2 3; do
  run early t3 20; TLAMC_EARLY_TABLE=0 run noearly t3 20
done
for rep in 1 2; do
  run early k10 30; TLAMC_EARLY_TABLE=0 run noearly k10 30
  run early raft5 8; TLAMC_EARLY_TABLE=0 run noearly raft5 8
done
