# round 4, GPU call b: the workgroup tail (survivors of 4 wavefronts pooled, sorted by action class, one atomicAdd) — parity (fused-engine suites), A/B;
# in parallel on the host cores: oracle goldens for config 4 (18 levels) and config 5 (10 levels)
cd /root/repo; D=gpurun_out/r04b; mkdir -p $D
make -s -C oracle >/dev/null 2>&1
( time oracle/_build/oracle_mc raft 5 6 2 5 1 1 --threads 48 --levels 18 --levels-out > $D/oracle_raft5_l18.txt ) 2> $D/oracle_raft5_l18.time &
OP=$!
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_checkpoint.py tests/test_frontend.py -m gpu -x -q --durations=5 > $D/pytest_gpu.log 2>&1; tail -n 8 $D/pytest_gpu.log
for f in "" "--no-inwave"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $f 2>$D/bench$f.err | grep -v amdgpu.ids > $D/bench$f.json; cut -c1-300 $D/bench$f.json
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload k10 $f 2>>$D/bench$f.err | grep -v amdgpu.ids > $D/bench_k10$f.json; cut -c1-200 $D/bench_k10$f.json
done
wait $OP; cat $D/oracle_raft5_l18.txt | cut -c1-700; cat $D/oracle_raft5_l18.time
( time oracle/_build/oracle_mc ssi 4 3 127 0 --threads 64 --levels 10 --levels-out > $D/oracle_ssi4x3_l10.txt ) 2> $D/oracle_ssi4x3_l10.time; cat $D/oracle_ssi4x3_l10.txt | cut -c1-500; cat $D/oracle_ssi4x3_l10.time
free -g | head -2
