# round 3, GPU call f: compact layout + message actions inline — fused step, kernels stand-alone (TLAMC_SERIAL), A/B against the
# family queues for the message actions (--no-inline), the 3-waves-per-SIMD register budget, chunk sizes; the new default workload (MaxTerm = 3) with the CPU baseline's scaling table
D=gpurun_out/r03f; mkdir -p $D
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raft or bench or complete" 2>&1 | tail -4 > $D/parity.log; cat $D/parity.log
B="python bench.py --workload k10 --no-cpu-baseline"
$B --steps 10 --warmup 2 > $D/k10.json 2>/dev/null
TLAMC_SERIAL=1 $B --steps 5 --warmup 1 > $D/k10_serial.json 2>/dev/null
$B --no-inline --steps 10 --warmup 2 > $D/k10_noinline.json 2>/dev/null
TLAMC_SERIAL=1 $B --no-inline --steps 5 --warmup 1 > $D/k10_noinline_serial.json 2>/dev/null
$B --occ3 --steps 10 --warmup 2 > $D/k10_occ3.json 2>/dev/null
TLAMC_SERIAL=1 $B --occ3 --steps 5 --warmup 1 > $D/k10_occ3_serial.json 2>/dev/null
$B --chunk 1048576 --steps 10 --warmup 2 > $D/k10_chunk1m.json 2>/dev/null
$B --chunk 524288 --steps 10 --warmup 2 > $D/k10_chunk512k.json 2>/dev/null
python bench.py --steps 5 --warmup 1 > $D/t3.json 2>$D/t3.err
for f in $D/k10*.json; do echo $f; python - $f <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().splitlines()[-1])
print(round(l["ms_per_step"], 2), l["roofline"]["kernel_ms"], round(l["roofline"]["frac"], 4))
PY
done
cut -c1-3000 $D/t3.json; tail -n 3 $D/t3.err
grep -H . /sys/fs/cgroup/cpu.max 2>/dev/null
