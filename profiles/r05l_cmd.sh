# round 5, GPU call l: where does the part of a step go that is not inside k_expand_family?  The kernel timeline of one step (rocprofv3
# --kernel-trace, profiles/r03l_gaps.py) at the new default of 2^24 - 256 frontier states per launch and at 2^23
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/r05l; mkdir -p $D
for c in 16776960 8388608; do
  rocprofv3 --kernel-trace --output-format csv -d $D/ktrace_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-atomic-add --chunk $c > $D/ktrace_$c.log 2>&1
  f=$(ls $D/ktrace_$c/*/*_kernel_trace.csv | head -1)
  python $R/profiles/r03l_gaps.py $f | tee $D/gaps_$c.txt
  python $R/profiles/r05l_levels.py $f > $D/levels_$c.txt; tail -n 30 $D/levels_$c.txt
  rm -rf $D/ktrace_$c
done
