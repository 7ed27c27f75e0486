# round 5, GPU call q: the in-wave tail asks for its parent rows again (LDS-DMA into the dead probe ring, nobody waits) before the
# counting sort and the barrier: parity (raft GPU cases), then A/B against --no-tail-prefetch, 20 steps, alternating
cd /root/repo; D=$PWD/gpurun_out/r05q; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "raft" > $D/pytest_gpu_raft.log 2>&1; grep -E 'passed|failed|error' $D/pytest_gpu_raft.log | tail -3
run() { # tag workload steps extra...
  local tag=$1 w=$2 st=$3; shift 3
  local out; out=$(timeout 300 python bench.py --steps $st --warmup 2 --no-cpu-baseline --no-atomic-add --workload $w "$@" 2>$D/err_${tag}_$w.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"variant\": \"$tag\", \"workload\": \"$w\", \"FAILED\": \"$(tail -c 300 $D/err_${tag}_$w.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(variant='$tag', workload='$w', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for rep in 1 2 3; do
  run prefetch t3 20; run noprefetch t3 20 --no-tail-prefetch
done
for rep in 1 2; do
  run prefetch k10 30; run noprefetch k10 30 --no-tail-prefetch
  run prefetch raft5 8; run noprefetch raft5 8 --no-tail-prefetch
done
