#!/bin/bash
# profiles/final_round.sh TAG — run on the GPU box (via gpurun) at the end of a round: the whole GPU test suite, smoke(),
# the contract bench line, the same command under rocprofv3 --kernel-trace --stats, and every workload of bench_all.py.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 300 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
grep "\"metric\"" $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/trace
cd $R
timeout 200 python profiles/bench_all.py > $OUT/bench_all.jsonl 2> $OUT/bench_all.err
ls -la $OUT
