# round 5, GPU call u (the tree the round ends with: survivor list 448, 2^24 - 256 frontier states per launch, nested records in the PlusCal
# front-end): the whole GPU suite with its slowest tests named, smoke(), rocprofv3 kernel stats + the separate PMC passes of the bench
# command + the request mix, the phase profile, the N = 1 point of a LAUNCHED run, and the driver's command WITH that PMC summary
cd /root/repo; D=$PWD/gpurun_out/r05u; mkdir -p $D
B=$PWD/tla_rust_amd/_build
timeout 1800 python -m pytest tests -m gpu -x -q --durations=10 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -14
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
BENCH_ARGS="--no-atomic-add --no-other-configs" timeout 1200 bash profiles/collect.sh r05u > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for set in "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $D/mix_$name -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-atomic-add --no-other-configs > $D/mix_$name.log 2>&1
  cp $D/mix_$name/*/*_counter_collection.csv $D/mix_$name.csv 2>/dev/null; rm -rf $D/mix_$name; rm -f $D/mix_$name.log
done
cd /root/repo
python profiles/summarize_pmc.py $D/request_mix.json $D/mix_*.csv > /dev/null 2>&1; rm -f $D/mix_*.csv
TLAMC_LIB=$B/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err; tail -c 200 $D/phase.err
cp $D/pmc.json profiles/r05u_pmc.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-atomic-add --no-other-configs 2>/dev/null | grep '"metric"' > $D/bench_launched_world1.json; python -c "
import json; d=json.load(open('$D/bench_launched_world1.json')); print('launched N=1:', round(d['ms_per_step'],2), d['roofline']['kernel'], 'xgmi' in d)"
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), {k: r[k] for k in ('frac','traffic','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','frac_of_request_ceiling','traffic_source','launches')}); print(json.dumps(d.get('atomic_add'))[:260])"
