# round 2, GPU call h: is k_expand_family (95 KB of code, 64 KB I-cache per CU pair) instruction-fetch bound?  I-cache counters of the bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r02h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
  cp $OUT/pmc_$name/*/*_counter_collection.csv $OUT/pmc_$name.csv 2>/dev/null
  tail -2 $OUT/pmc_$name.log | cut -c1-300
  rm -rf $OUT/pmc_$name
done
cd $R && python profiles/summarize_pmc.py $OUT/pmc.json $OUT/pmc_*.csv | grep -E "k_expand_family|k_materialise<" | cut -c1-900
