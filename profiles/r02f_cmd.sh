# round 2, GPU call f: local-owner shortcut of the sharded engine: GPU sharded tests (2/4/8 engines on one GPU) + RCCL world-1 bench vs fused
cd /root/repo; mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q > gpurun_out/r02f/pytest_sharded.log 2>&1; tail -5 gpurun_out/r02f/pytest_sharded.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02f/bench_fused.json 2> gpurun_out/r02f/bench_fused.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r02f/bench_rccl_world1.json 2> gpurun_out/r02f/bench_rccl_world1.err
TLAMC_PROFILE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 1 --warmup 1 > gpurun_out/r02f/bench_rccl_world1_prof.log 2>&1
for f in bench_fused bench_rccl_world1; do python -c "
import json,sys
for l in open('gpurun_out/r02f/$f.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', round(d['ms_per_step'],2), d['config'].get('verdict'), d['config'].get('distinct'))
"; done
tail -3 gpurun_out/r02f/bench_rccl_world1.err; grep phases gpurun_out/r02f/bench_rccl_world1_prof.log | tail -2
