#!/bin/bash
# profiles/collect.sh TAG — run on the GPU box (via gpurun): kernel-trace stats + PMC passes of bench.py.
# Counters are collected in their own runs (never combined with sys/hip tracing).
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
grep "\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_line.json
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/pmc_$name -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $OUT/pmc_$name.log 2>&1
  cp $OUT/pmc_$name/*/*_counter_collection.csv $OUT/pmc_$name.csv 2>/dev/null
  rm -rf $OUT/pmc_$name $OUT/pmc_$name.log
done
rm -rf $OUT/trace
ls -la $OUT
