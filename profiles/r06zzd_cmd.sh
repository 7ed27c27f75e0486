# round 6, GPU call zzd: what a BFS level of the deep PlusCal graphs costs outside the expand kernel (HIP events on / off, the blind grid's size)
cd /root/repo; D=$PWD/gpurun_out/r06zzd; mkdir -p $D
timeout 900 python profiles/pcal_level_overhead.py 2>$D/err.txt | tee $D/level_overhead.jsonl
grep -v amdgpu.ids $D/err.txt | tail -3
