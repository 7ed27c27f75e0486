# round 3, GPU call i: 32-bit messages (W 168 -> 136 / 128 B) — parity, bench, the 10^9-state run of config 4's model on one GPU
D=gpurun_out/r03i; mkdir -p $D
python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -x -q -m gpu --durations=5 2>&1 | tail -12 > $D/parity.log; cat $D/parity.log
python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "native or stay or front_door" 2>&1 | tail -4 > $D/sharded.log; cat $D/sharded.log
B="python bench.py --no-cpu-baseline"
$B --workload k10 --steps 10 --warmup 2 > $D/k10.json 2>/dev/null
TLAMC_SERIAL=1 $B --workload k10 --steps 5 --warmup 1 > $D/k10_serial.json 2>/dev/null
$B --steps 10 --warmup 2 > $D/t3.json 2>$D/t3.err
TLAMC_SERIAL=1 $B --steps 5 --warmup 1 > $D/t3_serial.json 2>/dev/null
python profiles/bench_all.py "raft 5 servers" 2>&1 | grep -v amdgpu.ids > $D/raft5.jsonl; cut -c1-600 $D/raft5.jsonl
for f in $D/k10*.json $D/t3*.json; do echo $f; python - $f <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().splitlines()[-1])
    print(round(l["ms_per_step"], 2), l["roofline"]["kernel_ms"], round(l["roofline"]["frac"], 4), round(l["value"] / 1e9, 3), l["roofline"]["state_bytes"])
except Exception as e:
    print("FAILED", e)
PY
done
tail -n 3 $D/t3.err
