# round 6, GPU call zj: the whole GPU suite again (call zi's run stopped at the new MC_F_PARK test, whose count of in-wave states forgot the batched
# small levels: 158 113 890 of 158 122 979; the assertion now allows for them) on the unchanged library
cd /root/repo; D=$PWD/gpurun_out/r06zj; mkdir -p $D
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call|s setup' $D/pytest_gpu_full.log | tail -12
