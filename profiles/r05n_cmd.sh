# round 5, GPU call n: flag-only A/Bs on the new kernel (survivor list 384, chunk 2^24 - 256), golden-gated, alternating: without the
# per-wavefront duplicate filter; the tail by wavefront; seen-set sizes 24 / 32 / 40 / 48 x 2^26 slots; odd chunks on a second stream
cd /root/repo; D=$PWD/gpurun_out/r05n; mkdir -p $D
run() { # tag workload extra-args...
  local tag=$1 w=$2; shift 2
  local out; out=$(timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $w "$@" 2>$D/err_${tag}_$w.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"variant\": \"$tag\", \"workload\": \"$w\", \"FAILED\": \"$(tail -c 300 $D/err_${tag}_$w.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(variant='$tag', workload='$w', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4), load=round(d['config']['seen_set_load'],3))))" | tee -a $D/ab.jsonl; fi
}
for rep in 1 2; do
  run base t3
  run nofilter t3 --no-filter
  run wavetail t3 --wave-tail
  run slots24 t3 --table-slots $((24 << 26))
  run slots32 t3 --table-slots $((32 << 26))
  run slots48 t3 --table-slots $((48 << 26))
  TLAMC_EXPAND_STREAMS=2 run streams2 t3
done
run base k10; run nofilter k10 --no-filter; run base k10; run nofilter k10 --no-filter
