# round 4, GPU call s (last): the whole GPU suite at the final commit (procedures, all_to_all_others, the driver's launcher line) + smoke + the contract line
cd /root/repo; D=gpurun_out/r04s; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu.log 2>&1; tail -n 10 $D/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 2 $D/smoke.log
timeout 900 python bench.py 2>$D/bench.err | grep metric > $D/bench_default_line.json; cut -c1-330 $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print({k: r[k] for k in ('frac','traffic','l2_hit_rate','pipeline_frac','kernel_ms','traffic_source')})"
export TLAMC_RCCL=$(python -c "import sys; sys.path.insert(0,'tests'); import helpers; print(helpers.build_fakerccl())")
timeout 900 python bench.py --gpus 8 --share-gpu --steps 1 --warmup 0 2>$D/bench_share8.err | grep metric > $D/bench_share_gpu_8.json; python -c "
import json; d=json.load(open('$D/bench_share_gpu_8.json')); print(round(d['ms_per_step'],1), d['config']['shares'], d['config']['levels'], d['xgmi'])"
