# round 5, GPU call d: the product library now launches the by-family kernel in workgroups of TWO wavefronts (a tail per pair: call c,
# 148.1 -> 140.4 ms).  A/B around it on t3 and k10 (golden-gated): w4 (four wavefronts, round 4), w1 (one: no barrier at all), w2a1 / w2a2
# (split-phase probes on top); then rocprofv3 PC sampling of the product kernel (line tables) — where do its wavefronts stall? — and
# the parity file on the product library
cd /root/repo; D=gpurun_out/r05d; mkdir -p $D
B=$PWD/tla_rust_amd/_build
run() { # tag lib workload extra
  local out; out=$(TLAMC_LIB=$2 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $3 $4 2>$D/err_$1_$3.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"lib\": \"$1\", \"workload\": \"$3\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$3.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(lib='$1', workload='$3', extra='$4', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4), inwave=r['inwave_states'])))" | tee -a $D/ab.jsonl; fi
}
for w in t3 k10; do
  run base $B/libtlamc.so $w
  for v in w4 w1 w2a1 w2a2; do run $v $B/libtlamc_$v.so $w; done
  run base2 $B/libtlamc.so $w
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -3 | tee $D/pytest_parity_base.log
timeout 600 bash profiles/pcsample2.sh r05d $B/libtlamc_pcs.so > $D/pcsample.log 2>&1; cat $D/status.txt; ls $D | head -30
