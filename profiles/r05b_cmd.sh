# round 5, GPU call b: A/B of the by-family expand kernel's variants on the contract workload (t3) and on k10, every run gated by the
# golden per-level counts inside bench.py: product library (folded fixed-slot loop, has_succ in registers), `old` (round 4's loop
# structure), a1 / a2 (split-phase probes: loads / loads + compare-and-swaps), w2 (workgroups of two wavefronts: a tail per pair),
# a2w2; a2 with MC_F_SYNCPROBE; then the parity file on the product library
cd /root/repo; D=gpurun_out/r05b; mkdir -p $D
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
  for v in old a1 a2 w2 a2w2; do run $v $B/libtlamc_$v.so $w; done
  run a2sync $B/libtlamc_a2.so $w --sync-probe
  run base2 $B/libtlamc.so $w
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > $D/pytest_parity_base.log; tail -3 $D/pytest_parity_base.log
TLAMC_LIB=$B/libtlamc_a2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 > $D/pytest_parity_a2.log; tail -3 $D/pytest_parity_a2.log
