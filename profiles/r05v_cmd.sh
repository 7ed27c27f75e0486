# round 5, GPU call v: the seen-set size of the contract workload, precisely (3 x 20 steps each, alternating): 24 / 28 / 32 / 40 x 2^26 slots
# (load 0.33 ... 0.20; clearing 12.9 ... 21.5 GB per step)
cd /root/repo; D=$PWD/gpurun_out/r05v; mkdir -p $D
run() { # tag slots
  local out; out=$(timeout 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs --table-slots $(($2 << 26)) 2>$D/err_$1.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"slots\": \"$1\", \"FAILED\": \"$(tail -c 300 $D/err_$1.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(slots='$2 x 2^26', ms_per_step=round(d['ms_per_step'],2), expand_ms=round(r['kernel_ms']['expand'],1), load=round(d['config']['seen_set_load'],3))))" | tee -a $D/ab.jsonl; fi
}
for rep in 1 2 3; do
  run s40 40; run s24 24; run s28 28; run s32 32
done
