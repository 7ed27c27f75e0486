# round 5, GPU call c: cache-policy A/B of the by-family expand kernel on t3 and k10 (every run gated by the golden per-level counts):
# base (round 4's loop structure again), ntp (seen-set probes read `nt`: past the L2), nts (state rows stored `nt`), ntps (both),
# w2 (workgroups of two wavefronts), a1 (split-phase probe loads), a1ntps, w2ntps.  Then the parity file on the product library.
cd /root/repo; D=gpurun_out/r05c; mkdir -p $D
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
  for v in ntp nts ntps w2 a1 a1ntps w2ntps; do run $v $B/libtlamc_$v.so $w; done
  run base2 $B/libtlamc.so $w
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E 'passed|failed|error' | tail -3 | tee $D/pytest_parity_base.log
