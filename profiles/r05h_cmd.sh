# round 5, GPU call h: the in-wave writer stores every word ONCE (the action's changes stay in registers and the row is copied through
# them) against round 4's copy-then-overwrite form (`tw`: its scattered 8-byte stores made the L2 hand 2.05 G write requests per step to
# the memory for 1.05 G sectors of rows) — golden-gated A/B on t3 / k10 / raft5; which GPU cases of the compiled programs are slow; then
# the counters of the new kernels and the driver's command
cd /root/repo; D=$PWD/gpurun_out/r05h; mkdir -p $D
B=$PWD/tla_rust_amd/_build
run() { # tag lib workload
  local out; out=$(TLAMC_LIB=$2 timeout 300 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-atomic-add --workload $3 2>$D/err_$1_$3.log | grep '"metric"')
  if [ -z "$out" ]; then echo "{\"lib\": \"$1\", \"workload\": \"$3\", \"FAILED\": \"$(tail -c 300 $D/err_$1_$3.log | tr '\n"' '  ')\"}" | tee -a $D/ab.jsonl
  else echo "$out" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps(dict(lib='$1', workload='$3', ms_per_step=round(d['ms_per_step'],2), kernel_ms={k: round(v,1) for k,v in r['kernel_ms'].items()}, frac=round(r['frac'],4))))" | tee -a $D/ab.jsonl; fi
}
for w in t3 k10 raft5; do
  run once $B/libtlamc.so $w; run twice $B/libtlamc_tw.so $w; run once2 $B/libtlamc.so $w; run twice2 $B/libtlamc_tw.so $w
done
timeout 900 python -m pytest tests/test_gpu_pcal.py -m gpu -x -q --durations=12 -k "recursive_sum or even_odd or proc_demo or peterson" 2>&1 | grep -E 'passed|failed|error|s call|s setup' | tail -16 | tee $D/pytest_gpu_pcal_durations.log
BENCH_ARGS="--no-atomic-add" timeout 1200 bash profiles/collect.sh r05h > $D/collect.log 2>&1
python profiles/summarize_pmc.py $D/pmc.json $D/pmc_*.csv > $D/pmc_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for set in "TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $D/mix_$name -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-atomic-add > $D/mix_$name.log 2>&1
  cp $D/mix_$name/*/*_counter_collection.csv $D/mix_$name.csv 2>/dev/null; rm -rf $D/mix_$name; rm -f $D/mix_$name.log
done
cd /root/repo
python profiles/summarize_pmc.py $D/request_mix.json $D/mix_*.csv > /dev/null 2>&1; rm -f $D/mix_*.csv
TLAMC_LIB=$B/libtlamc_prof.so timeout 600 python profiles/phase_prof.py 8 0 > $D/phase_profile_t3.json 2>$D/phase.err
cp $D/pmc.json profiles/r05h_pmc.json
timeout 900 python bench.py 2>$D/bench.err | grep -v amdgpu.ids > $D/bench_default_line.json; python -c "
import json; d=json.load(open('$D/bench_default_line.json')); r=d['roofline']; print(round(d['ms_per_step'],2), round(d['value']/1e9,3), {k: r[k] for k in ('frac','traffic','traffic_lower','l2_hit_rate','pipeline_frac','kernel_ms','frac_of_request_ceiling')})"
