# round 4, GPU call j: arena indices by atomicAdd only where wavefronts write in-wave (atomic_add / SSI / the no-inwave A/B back to stream-order
# appends); world-1 sharded bench as in round 3 (r03n: 166.5 ms)
cd /root/repo; D=gpurun_out/r04j; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py -m gpu -x -q > $D/pytest_gpu_parity.log 2>&1; tail -n 2 $D/pytest_gpu_parity.log
for f in "" "--no-inwave"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add $f 2>$D/bench$f.err | grep -v amdgpu.ids > $D/bench$f.json; cut -c1-260 $D/bench$f.json
done
timeout 300 python -c "
import sys, json; sys.path.insert(0, '.')
import bench, tla_rust_amd as amd
print(json.dumps(bench.atomic_add_series(amd, 0))[:900])" 2>&1 | grep -v amdgpu.ids | tee $D/atomic_add.json
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload ssi4x3 2>$D/bench_ssi.err | grep -v amdgpu.ids > $D/bench_ssi4x3.json; cut -c1-260 $D/bench_ssi4x3.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 2>$D/w1.err | grep -v amdgpu.ids > $D/bench_world1_rccl.json; cut -c1-300 $D/bench_world1_rccl.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --workload raft5 2>>$D/w1.err | grep -v amdgpu.ids > $D/bench_world1_rccl_raft5.json; cut -c1-300 $D/bench_world1_rccl_raft5.json
