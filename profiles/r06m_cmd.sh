# round 6, GPU call m: generated code through the BY-PAIRS kernel with dynamic keys (a wavefront's enabled (parent, slot) pairs sorted by the
# label the slot's process stands at): the JIT tests, the SSI parity cases (the same kernel, static families), and the timed comparison —
# pairs / slot by slot / interpreter
cd /root/repo; D=$PWD/gpurun_out/r06m; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_zz_jit.py -m gpu -x -q > $D/pytest_gpu_jit.log 2>&1; tail -n 3 $D/pytest_gpu_jit.log
timeout 900 python -m pytest tests -m gpu -x -q -k "ssi or SSI or textbook or si_" > $D/pytest_gpu_ssi.log 2>&1; tail -n 2 $D/pytest_gpu_ssi.log
timeout 1200 python profiles/bench_jit.py msq3 pagecache msq4 > $D/bench_jit.jsonl 2>$D/bench_jit.err; cut -c1-360 $D/bench_jit.jsonl; tail -n 3 $D/bench_jit.err
