# round 3, GPU call g: three queued families + inline message actions + bucket prefetch — parity, A/B
D=gpurun_out/r03g; mkdir -p $D
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raft or bench or complete" 2>&1 | tail -4 > $D/parity.log; cat $D/parity.log
B="python bench.py --workload k10 --no-cpu-baseline"
$B --steps 10 --warmup 2 > $D/k10.json 2>/dev/null
TLAMC_SERIAL=1 $B --steps 5 --warmup 1 > $D/k10_serial.json 2>/dev/null
$B --no-prefetch --steps 10 --warmup 2 > $D/k10_noprefetch.json 2>/dev/null
TLAMC_SERIAL=1 $B --no-prefetch --steps 5 --warmup 1 > $D/k10_noprefetch_serial.json 2>/dev/null
$B --occ3 --steps 10 --warmup 2 > $D/k10_occ3.json 2>/dev/null
$B --no-filter --steps 10 --warmup 2 > $D/k10_nofilter.json 2>/dev/null
python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $D/t3.json 2>$D/t3.err
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-prefetch > $D/t3_noprefetch.json 2>/dev/null
for f in $D/*.json; do echo $f; python - $f <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().splitlines()[-1])
    print(round(l["ms_per_step"], 2), l["roofline"]["kernel_ms"], round(l["roofline"]["frac"], 4), round(l["value"] / 1e9, 3))
except Exception as e:
    print("FAILED", e)
PY
done
tail -n 3 $D/t3.err
