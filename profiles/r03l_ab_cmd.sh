# round 3, the A/B runs behind DESIGN.md section 5's "measured and NOT adopted" list (each was one gpurun call with this file holding
# the commands of that experiment; the last one is kept): seen-set bucket width and sparsity, non-temporal parent loads, XCD-aware
# tile order, frontier states per launch, pipelined probes, level-boundary overlap, stream priorities, and the kernel timeline of a
# step (profiles/r03l_gaps.py -> profiles/r03l_gaps.txt)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ktrace -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/ktrace.log 2>&1
f=$(ls $R/gpurun_out/ktrace/*/*_kernel_trace.csv | head -1)
python $R/profiles/r03l_gaps.py $f | tee $R/gpurun_out/r03l_gaps.txt
rm -rf $R/gpurun_out/ktrace
for p in 0 1 2; do TLAMC_PRIO=$p python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/ab_prio$p.json 2> $R/gpurun_out/ab.err; done
