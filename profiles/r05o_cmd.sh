# round 5, GPU call o: ablation of the expand kernel on the contract workload (profiles/ablate_t3.py): generation only / + probes
# (all hit) / parent loads only, against the run's own expand time
cd /root/repo; D=$PWD/gpurun_out/r05o; mkdir -p $D
timeout 600 python profiles/ablate_t3.py 2>$D/err.log | tee $D/ablate_t3.json; tail -c 400 $D/err.log
