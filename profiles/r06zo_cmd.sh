# round 6, GPU call zo: seen-set size A/B (zero kernel change; $TLAMC_SPARSE_RATIO lets the 32-byte probe mode run on a fuller table): t3 and ssi4x3,
# then config 4's model through bench.py at 3 << 30 (default) and 5 << 29 slots
cd /root/repo; D=$PWD/gpurun_out/r06zo; mkdir -p $D
timeout 1500 python profiles/table_ab.py 2>$D/err.log | tee $D/table_ab.jsonl
tail -n 3 $D/err.log
for S in 3221225472 2684354560 3221225472 2684354560; do
  TLAMC_SPARSE_RATIO=2.5 timeout 300 python bench.py --workload raft5 --no-cpu-baseline --no-atomic-add --no-pcal --no-other-configs --steps 3 --warmup 1 --table-slots $S 2>>$D/err5.log | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(dict(workload='raft5', slots=$S, ms=round(d['ms_per_step'],2), kernel_ms=d['roofline'].get('kernel_ms'))))" | tee -a $D/table_ab_raft5.jsonl
done
tail -n 3 $D/err5.log
