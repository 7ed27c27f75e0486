# round 2, GPU call zi: bench.py --msg-keys 11 (the 3.4e8-state complete graph as an optional longer step)
cd /root/repo; mkdir -p gpurun_out/r02zi
timeout 100 python bench.py --msg-keys 11 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids > gpurun_out/r02zi/bench_k11_line.json; cut -c1-700 gpurun_out/r02zi/bench_k11_line.json
