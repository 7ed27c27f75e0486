# round 5, GPU call m: the survivor list of a by-family wavefront with 384 entries (a ring that is no power of two; the kilobyte comes
# from the duplicate filter, 256 -> 128 entries; the tail's sort order lies over the dead family queues + filter) against 256 / 256
# (`o256`): nearly nothing goes through the new-list and k_materialise any more.  Parity first (the raft GPU cases), then golden-gated A/B
cd /root/repo; D=$PWD/gpurun_out/r05m; mkdir -p $D
B=$PWD/tla_rust_amd/_build
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "raft" > $D/pytest_gpu_raft.log 2>&1; grep -E 'passed|failed|error' $D/pytest_gpu_raft.log | tail -3
run() { # tag lib workload
  local out; out=$(TLAMC_LIB=$2 timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $3 2>$D/err_$1_$3.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"lib\": \"$1\", \"workload\": \"$3\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$3.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(lib='$1', workload='$3', ms_per_step=round(d['ms_per_step'],2), launches=r['launches'], inwave=r['inwave_states'], kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for w in t3 k10 k11; do
  run o384 $B/libtlamc.so $w; run o256 $B/libtlamc_o256.so $w; run o384 $B/libtlamc.so $w; run o256 $B/libtlamc_o256.so $w
done
run o384 $B/libtlamc.so raft5; run o384 $B/libtlamc.so raft5
