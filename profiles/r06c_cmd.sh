# round 6, GPU call c: rocprofv3 kernel stats + the PMC passes of `bench.py --workload ssi4x3` on the FIRST form of k_expand_pairs<SpecSsi>
cd /root/repo; D=$PWD/gpurun_out/r06c; mkdir -p $D
BENCH_ARGS="--workload ssi4x3 --no-atomic-add --no-other-configs" timeout 1500 bash profiles/collect.sh r06c_ssi4x3 > $D/collect.log 2>&1
S=$PWD/gpurun_out/r06c_ssi4x3
PMC_SPEC=ssi python profiles/summarize_pmc.py $D/ssi4x3_pmc.json $S/pmc_*.csv > $D/ssi4x3_pmc_summary.txt 2>&1
cp $S/kernel_stats.csv $D/ssi4x3_kernel_stats.csv; cp $S/bench_line.json $D/ssi4x3_bench_line_under_rocprof.json; rm -rf $S
grep "k_expand_pairs\|k_check_frontier" $D/ssi4x3_pmc_summary.txt
head -4 $D/ssi4x3_kernel_stats.csv | cut -c1-60,330-480
