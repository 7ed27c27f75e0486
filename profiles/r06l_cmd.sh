# round 6, GPU call l: generated code on NAMED cells (struct Cells: no array the compiler can index with a variable — call k's kernels kept a
# 192-byte array per lane in scratch memory because a chain of selects had been turned back into an indexed load): tests + the timed comparison
cd /root/repo; D=$PWD/gpurun_out/r06l; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q > $D/pytest_gpu_jit.log 2>&1; tail -n 3 $D/pytest_gpu_jit.log
timeout 1200 python profiles/bench_jit.py msq3 pagecache msq4 > $D/bench_jit.jsonl 2>$D/bench_jit.err; cut -c1-330 $D/bench_jit.jsonl; tail -n 3 $D/bench_jit.err
