# round 6, GPU call zp: (1) the new tests: mc_engine_request_stop, `mc` moving a long interpreter run to generated code on its own; (2) the seen-set
# size of the contract workload once more, precisely (call zo: 40 << 26 slots 128.7 / 130.6 ms on a drifting box, 28 << 26 125.8): 3 x 20 steps each, alternating
cd /root/repo; D=$PWD/gpurun_out/r06zp; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_zz_jit.py tests/test_frontend.py -m gpu -x -q --durations=4 > $D/pytest.txt 2>&1; tail -n 12 $D/pytest.txt
for rep in 1 2 3; do for S26 in 40 28 32 48 24; do
  timeout 300 python bench.py --no-cpu-baseline --no-atomic-add --no-pcal --no-other-configs --steps 20 --warmup 2 --table-slots $((S26 << 26)) 2>>$D/err.log | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(dict(workload='t3', slots_26=$S26, rep=$rep, ms=round(d['ms_per_step'],2), expand_ms=round(d['roofline']['kernel_ms']['expand'],2), frac=round(d['roofline']['frac'],4))))" | tee -a $D/table_ab_t3.jsonl
done; done
tail -n 3 $D/err.log
