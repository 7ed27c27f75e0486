# round 4, GPU call k: config 4's model fused, in-wave writes against k_materialise (its 18 levels all GROW by 2.4 x: ~150 survivors per wavefront)
cd /root/repo; D=gpurun_out/r04k; mkdir -p $D
for f in "" "--no-inwave"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload raft5 $f 2>$D/bench_raft5$f.err | grep -v amdgpu.ids > $D/bench_raft5$f.json; cut -c1-300 $D/bench_raft5$f.json
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload k10 $f 2>$D/bench_k10$f.err | grep -v amdgpu.ids > $D/bench_k10$f.json; cut -c1-200 $D/bench_k10$f.json
done
