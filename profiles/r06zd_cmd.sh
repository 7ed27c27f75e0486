# round 6, GPU call zd: the budget's last level checked BESIDE its own generation (k_check_range on stream2, chunk by chunk) instead of by
# k_check_frontier after the level — SSI parity tests first (violations of the last level included), then config 5's model A/B
# (A/B knob: TLAMC_CHECK_AFTER=1, read by the engine, keeps the form of rounds 3-5);
cd /root/repo; D=$PWD/gpurun_out/r06zd; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q -k "ssi or si or symmetry or textbook or step or checkpoint" > $D/pytest_ssi.log 2>&1; grep -E "passed|failed" $D/pytest_ssi.log | tail -n 1
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "deep_command" > $D/pytest_deep.log 2>&1; grep -E "passed|failed" $D/pytest_deep.log | tail -n 1
for v in beside after beside after; do
  TLAMC_CHECK_AFTER=$([ $v = after ] && echo 1 || echo "") timeout 600 python bench.py --workload ssi4x3 --steps 20 --warmup 2 --no-atomic-add --no-other-configs --no-pcal --no-cpu-baseline 2>>$D/bench.err | grep -v amdgpu.ids | V=$v python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'workload':'ssi4x3','last_level_check':os.environ['V'],'ms_per_step':round(d['ms_per_step'],3),'kernel_ms':r.get('kernel_ms')}))" | tee -a $D/ab.jsonl
done
tail -n 3 $D/bench.err
# RESULT (profiles/r06zd_ab.jsonl): beside 17.13 / 17.15 ms, after 16.73 / 16.73 ms per step — the expand kernels of the last level take 2.7 ms longer with
# the check kernel beside them (15.7 against 12.9 ms of HIP-event time) and the check itself is 2.9 ms: the overlap buys nothing, both kernels wait on the
# same memory system.  NOT adopted: the engine code of this call (k_snap_arena / k_check_range, the shrunken chunks of the last level) was removed again.
