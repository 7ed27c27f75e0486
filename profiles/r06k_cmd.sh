# round 6, GPU call k: the compiled-PlusCal path as GENERATED code (MC_F_JIT) for the first time on a device — its GPU tests against the
# interpreter, the timed comparison on the path's larger models (profiles/bench_jit.py), and the FIRST counters this path ever had
# (VERDICT round 5, next 3): rocprofv3 kernel stats + PMC passes of one run of pagecache N = 3 and ms_queue_counted K = 3 on each back-end
cd /root/repo; D=$PWD/gpurun_out/r06k; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q > $D/pytest_gpu_jit.log 2>&1; tail -n 5 $D/pytest_gpu_jit.log
timeout 1200 python profiles/bench_jit.py msq3 pagecache msq4 > $D/bench_jit.jsonl 2>$D/bench_jit.err; cut -c1-330 $D/bench_jit.jsonl; tail -n 3 $D/bench_jit.err
cd /tmp && export TMPDIR=/tmp
for m in pagecache msq3; do for be in jit vm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/tr_${m}_$be -- python /root/repo/profiles/run_pcal_once.py $m $be > $D/tr_${m}_$be.log 2>&1
  cp $D/tr_${m}_$be/*/*_kernel_stats.csv $D/${m}_${be}_kernel_stats.csv 2>/dev/null; rm -rf $D/tr_${m}_$be
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-24)
    rocprofv3 --pmc $set --output-format csv -d $D/pmc_${m}_${be}_$name -- python /root/repo/profiles/run_pcal_once.py $m $be > /dev/null 2>&1
    cp $D/pmc_${m}_${be}_$name/*/*_counter_collection.csv $D/pmc_${m}_${be}_$name.csv 2>/dev/null; rm -rf $D/pmc_${m}_${be}_$name
  done
  PMC_SPEC=vm python /root/repo/profiles/summarize_pmc.py $D/${m}_${be}_pmc.json $D/pmc_${m}_${be}_*.csv > /dev/null 2>&1; rm -f $D/pmc_${m}_${be}_*.csv
  python - <<PY
import json, csv
d = json.load(open('$D/${m}_${be}_pmc.json'))
for k, v in d.items():
    if k.startswith('k_expand_insert') or k.startswith('k_materialise<'):
        print('$m $be', k[:60], {a: (round(b / 1e6, 1) if isinstance(b, float) else b) for a, b in v.items() if a in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES', 'FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'launches')}, '(millions; SIZE in KB/1e6 = GB)')
PY
done; done
