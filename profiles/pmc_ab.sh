R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in fam nofam; do
  fl=""; [ $v = nofam ] && fl="--no-family"
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    n=$(echo $set | tr " " "_" | cut -c1-20)
    rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/ab_${v}_$n -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $fl > /dev/null 2>&1
    cp $R/gpurun_out/ab_${v}_$n/*/*_counter_collection.csv $R/gpurun_out/ab_${v}_$n.csv; rm -rf $R/gpurun_out/ab_${v}_$n
  done
done
ls $R/gpurun_out/ab_*
