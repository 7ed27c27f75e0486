# round 4, GPU call c: workgroup tail (pooled survivors, sorted by action class) — the whole GPU suite, A/B bench lines (nothing else on the host),
# the new bench workloads (config 4 / config 5) fused and through the N-rank path on one GPU (stand-in librccl: functional, not a measurement)
cd /root/repo; D=gpurun_out/r04c; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $D/pytest_gpu.log 2>&1; tail -n 8 $D/pytest_gpu.log
for f in "" "--no-inwave"; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline $f 2>$D/bench$f.err | grep -v amdgpu.ids > $D/bench$f.json; cut -c1-300 $D/bench$f.json
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload k10 $f 2>>$D/bench$f.err | grep -v amdgpu.ids > $D/bench_k10$f.json; cut -c1-200 $D/bench_k10$f.json
done
for w in raft5 ssi4x3; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>$D/bench_$w.err | grep -v amdgpu.ids > $D/bench_$w.json; cut -c1-300 $D/bench_$w.json; tail -n 2 $D/bench_$w.err
done
export TLAMC_RCCL=$(python -c "import sys; sys.path.insert(0,'tests'); import helpers; print(helpers.build_fakerccl())")
for w in ssi4x3 raft5; do
  timeout 900 python bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --workload $w 2>$D/bench_share2_$w.err | grep -v amdgpu.ids > $D/bench_share2_$w.json; cut -c1-300 $D/bench_share2_$w.json; tail -n 3 $D/bench_share2_$w.err
done
