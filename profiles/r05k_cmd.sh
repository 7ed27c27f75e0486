# round 5, GPU call k: (1) the GPU cases of the nested-record Michael-Scott queue (tests/test_gpu_zz_ms_queue.py) and the record cases;
# (2) A/B of the frontier states per launch: 2^23 (the cap until now: 78 large launches per step on t3) against 2^24 - 256 (what a 24-bit
# column takes: 46), alternating, golden-gated; (3) the compiled-PlusCal path on a 35 M / 125 M-state lock-free model
cd /root/repo; D=$PWD/gpurun_out/r05k; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_zz_ms_queue.py tests/test_gpu_pcal.py -m gpu -x -q --durations=8 -k "ms_queue or counted or records or procedures" > $D/pytest_gpu_msq.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_msq.log | tail -12
run() { # tag chunk workload
  local out; out=$(timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $3 --chunk $2 2>$D/err_$1_$3.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"chunk\": \"$1\", \"workload\": \"$3\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$3.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(chunk='$1', workload='$3', ms_per_step=round(d['ms_per_step'],2), launches=r['launches'], kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for w in t3 k11 raft5; do
  run 2p23 8388608 $w; run 2p24 16776960 $w; run 2p23 8388608 $w; run 2p24 16776960 $w
done
timeout 600 python profiles/bench_msq_counted.py 2>$D/msq.err | tee $D/bench_msq_counted.jsonl; tail -c 300 $D/msq.err
