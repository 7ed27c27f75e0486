# round 5, GPU call t: survivor list of 448 entries (LDS 20.3 KB per workgroup: eight of them fill the CU's 160 KB to the byte) against 384:
# t3 / k10 with the 3-server kernels (`o448`), raft5 with the 5-server kernels (`o448r5`); alternating, golden-gated
cd /root/repo; D=$PWD/gpurun_out/r05t; mkdir -p $D
B=$PWD/tla_rust_amd/_build
run() { # tag lib workload steps
  local out; out=$(TLAMC_LIB=$2 timeout 300 python bench.py --steps $4 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs --workload $3 2>$D/err_$1_$3.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"lib\": \"$1\", \"workload\": \"$3\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$3.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(lib='$1', workload='$3', ms_per_step=round(d['ms_per_step'],2), inwave=r['inwave_states'], kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for rep in 1 2 3; do
  run o384 $B/libtlamc.so t3 20; run o448 $B/libtlamc_o448.so t3 20
done
for rep in 1 2 3; do
  run o384 $B/libtlamc.so raft5 8; run o448 $B/libtlamc_o448r5.so raft5 8
done
for rep in 1 2; do
  run o384 $B/libtlamc.so k10 30; run o448 $B/libtlamc_o448.so k10 30
done
