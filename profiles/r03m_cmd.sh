# round 3, GPU call k (third checkpoint: 32-byte seen-set probes, per-rank checkpoints, host evaluator): whole GPU suite + smoke, the contract bench line with cpu_baseline,
# rocprofv3 kernel stats + PMC passes of the same command, every lowered workload, world-1 RCCL against fused
cd /root/repo; D=gpurun_out/r03m; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $D/pytest_gpu.log 2>&1; tail -n 15 $D/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; cut -c1-400 $D/bench_default_line.json
timeout 900 bash profiles/collect.sh r03m > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 2>/dev/null | grep -v amdgpu.ids > $D/bench_world1_rccl.json; cut -c1-300 $D/bench_world1_rccl.json
timeout 900 python profiles/bench_all.py 2>&1 | grep -v amdgpu.ids > $D/bench_all_workloads.jsonl; cut -c1-220 $D/bench_all_workloads.jsonl
ls $D
