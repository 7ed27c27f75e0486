# round 4, GPU call i: why did the world-1 sharded run of the 5-server model take 100 s in the suite (r04h)?  + atomic_add without the filter
cd /root/repo; D=gpurun_out/r04i; mkdir -p $D
timeout 300 python profiles/sharded_w1_probe.py t3 23 2>&1 | grep -v amdgpu.ids | tee $D/sharded_w1_t3.jsonl | cut -c1-600
timeout 600 python profiles/sharded_w1_probe.py raft5 21 2>&1 | grep -v amdgpu.ids | tee $D/sharded_w1_raft5.jsonl | cut -c1-600
timeout 300 python -c "
import sys, json; sys.path.insert(0, '.')
import bench, tla_rust_amd as amd
print(json.dumps(bench.atomic_add_series(amd, 0))[:700])" 2>&1 | grep -v amdgpu.ids | tee $D/atomic_add.json
