# round 5, GPU call e: the product library as it now stands (by-family kernel in workgroups of two wavefronts; device code in
# engine_kernels.h): the whole GPU suite + smoke, rocprofv3 kernel stats + the separate PMC passes of the bench command, the phase
# profile, what PC-sampling configurations this box offers, and the driver's command WITH that PMC summary
cd /root/repo; D=gpurun_out/r05e; mkdir -p $D
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error' $D/pytest_gpu_full.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
BENCH_ARGS="--no-atomic-add" timeout 1200 bash profiles/collect.sh r05e > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1; cp $D/pmc.json profiles/r05e_pmc.json
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err
(cd /tmp; rocprofv3 -L 2>&1 | grep -i -B3 -A25 'pc.sampl' | head -120) > $D/pc_sampling_avail.txt
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), {k: r[k] for k in ('frac','traffic','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','avg_launch_ms','launches')}); print(json.dumps(d.get('atomic_add'))[:300])"
ls $D
