# round 2, GPU call zf: the next complete 3-server raft graph (MaxMsgKeys = 11, 3.4e8 states) verified by the exact-dedup CPU oracle
# on the GPU box's host (the build container's 62 GB cannot hold it), next to the engine's run of the same model
cd /root/repo; mkdir -p gpurun_out/r02zf
free -g | head -2 > gpurun_out/r02zf/mem.txt; nproc >> gpurun_out/r02zf/mem.txt; cat gpurun_out/r02zf/mem.txt
avail=$(free -g | awk '/Mem:/{print $7}')
python profiles/explore_complete.py '[3,4,2,3,1,1,11,2,6,11]' 2>&1 | grep -v amdgpu.ids > gpurun_out/r02zf/gpu_k11.jsonl; cut -c1-300 gpurun_out/r02zf/gpu_k11.jsonl
if [ "$avail" -gt 300 ]; then
  ( ulimit -v 400000000; timeout 420 oracle/_build/oracle_mc raft 3 4 2 3 1 1 0 11 --threads 192 --levels-out > gpurun_out/r02zf/oracle_k11.txt 2> gpurun_out/r02zf/oracle_k11.err ); echo "oracle rc=$?"; cut -c1-400 gpurun_out/r02zf/oracle_k11.txt | head -3
else echo "not enough host memory: $avail GB"; fi
