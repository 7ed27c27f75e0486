# round 3, GPU call j: guard mask of 128 fixed slots (5 servers) — config 4's model to 10^9 states; sparse (4-slot, 32-byte) seen-set A/B
D=gpurun_out/r03j; mkdir -p $D
python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=5 2>&1 | tail -8 > $D/parity.log; cat $D/parity.log
python profiles/bench_all.py "raft 5 servers" 2>&1 | grep -v amdgpu.ids > $D/raft5.jsonl; cut -c1-600 $D/raft5.jsonl
B="python bench.py --no-cpu-baseline"
$B --workload k10 --steps 10 --warmup 2 > $D/k10.json 2>/dev/null
$B --workload k10 --table-slots 335544320 --steps 10 --warmup 2 > $D/k10_sparse.json 2>/dev/null
TLAMC_SERIAL=1 $B --workload k10 --table-slots 335544320 --steps 5 --warmup 1 > $D/k10_sparse_serial.json 2>/dev/null
$B --workload k10 --table-slots 268435456 --steps 10 --warmup 2 > $D/k10_dense_2p28.json 2>/dev/null
$B --steps 5 --warmup 1 > $D/t3.json 2>/dev/null
$B --table-slots 1610612736 --steps 5 --warmup 1 > $D/t3_sparse.json 2>/dev/null
for f in $D/k10*.json $D/t3*.json; do echo $f; python - $f <<'PY'
import json, sys
try:
    l = json.loads(open(sys.argv[1]).read().splitlines()[-1])
    print(round(l["ms_per_step"], 2), l["roofline"]["kernel_ms"], round(l["roofline"]["frac"], 4), round(l["value"] / 1e9, 3), l["config"]["seen_set_load"])
except Exception as e:
    print("FAILED", e)
PY
done
