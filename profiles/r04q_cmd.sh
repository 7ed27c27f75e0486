# round 4, GPU call q (the state the round ends in, after the generated-only shortcuts): whole GPU suite + smoke, rocprofv3 kernel stats + separate PMC passes of the bench command,
# the contract line WITH that PMC summary (traffic / L2 hit rate), every lowered workload, world-1 RCCL lines, phase profile, table-size A/B
cd /root/repo; D=gpurun_out/r04q; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu.log 2>&1; tail -n 12 $D/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
BENCH_ARGS="--no-atomic-add" timeout 1200 bash profiles/collect.sh r04q > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1; cp $D/pmc.json profiles/r04q_pmc.json
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; cut -c1-400 $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print({k: r[k] for k in ('frac','traffic','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','avg_launch_ms','launches')}); print(json.dumps(d.get('atomic_add'))[:400])"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 1 2>/dev/null | grep metric > $D/bench_world1_rccl.json; cut -c1-200 $D/bench_world1_rccl.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --workload raft5 2>/dev/null | grep metric > $D/bench_world1_rccl_raft5.json; cut -c1-200 $D/bench_world1_rccl_raft5.json
for w in raft5 ssi4x3; do timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload $w 2>/dev/null | grep metric > $D/bench_$w.json; cut -c1-200 $D/bench_$w.json; done
timeout 900 python profiles/bench_all.py 2>&1 | grep -v amdgpu.ids > $D/bench_all_workloads.jsonl; cut -c1-160 $D/bench_all_workloads.jsonl
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err
ls $D
for ts in 1610612736 2684354560 4294967296; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-atomic-add --table-slots $ts 2>/dev/null | grep metric > $D/bench_table_$ts.json; python -c "
import json; d=json.load(open('$D/bench_table_$ts.json')); print('table', $ts, round(d['ms_per_step'],1), d['roofline']['probe_bytes'])"; done
