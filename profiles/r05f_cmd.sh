# round 5, GPU call f: the chunks of a level alternating between two streams ($TLAMC_EXPAND_STREAMS=2: chunk c+1 fills the CUs the last
# workgroups of chunk c leave idle) against the stream order, on t3 / k10 / raft5 (golden-gated); the request mix of the fused kernel
# (L2 reads / writes / atomics and what leaves the L2, by size); then the whole GPU suite on this tree
cd /root/repo; D=gpurun_out/r05f; mkdir -p $D
run() { # tag workload env
  local out; out=$(env $3 timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $2 2>$D/err_$1_$2.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"variant\": \"$1\", \"workload\": \"$2\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$2.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(variant='$1', workload='$2', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for w in t3 k10 raft5; do
  run one $w TLAMC_EXPAND_STREAMS=1
  run two $w TLAMC_EXPAND_STREAMS=2
  run one2 $w TLAMC_EXPAND_STREAMS=1
  run two2 $w TLAMC_EXPAND_STREAMS=2
done
cd /tmp && export TMPDIR=/tmp
for set in "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $D/mix_$name -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-atomic-add > $D/mix_$name.log 2>&1
  cp $D/mix_$name/*/*_counter_collection.csv $D/mix_$name.csv 2>/dev/null; rm -rf $D/mix_$name; tail -c 300 $D/mix_$name.log > $D/mix_$name.tail; rm -f $D/mix_$name.log
done
cd /root/repo
python profiles/summarize_pmc.py $D/request_mix.json $D/mix_*.csv > /dev/null 2>&1; rm -f $D/mix_*.csv
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error' $D/pytest_gpu_full.log | tail -3
