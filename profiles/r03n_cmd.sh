# round 3, last GPU call: the whole GPU suite + smoke + the contract bench line (with the stamped PMC summary of r03m found by bench.py)
cd /root/repo; D=gpurun_out/r03n; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=5 > $D/pytest_gpu.log 2>&1; tail -n 8 $D/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; cut -c1-300 $D/bench_default_line.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 2>/dev/null | grep -v amdgpu.ids > $D/bench_world1_rccl.json; cut -c1-200 $D/bench_world1_rccl.json
