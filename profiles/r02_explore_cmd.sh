cd /root/repo
mkdir -p gpurun_out
timeout 900 python profiles/explore_complete.py \
 '[3,4,2,3,1,1,8,4,16,8]' '[3,4,2,3,1,1,9,4,16,9]' '[3,4,2,3,1,1,10,4,16,10]' '[3,4,2,3,1,1,11,4,16,11]' '[3,4,2,3,1,1,12,4,16,12]' \
 '[3,4,3,3,1,1,7,4,16,7]' '[3,4,3,3,1,1,8,4,16,8]' '[3,4,3,3,1,1,9,4,16,9]' \
 '[3,4,2,3,2,1,7,4,16,7]' '[3,4,2,3,2,1,8,4,16,8]' '[3,4,2,3,2,1,9,4,16,9]' \
 '[3,2,2,3,1,1,12,4,16,12]' '[3,2,2,3,1,1,14,4,16,14]' '[3,3,2,3,1,1,12,4,16,12]' \
 > gpurun_out/explore1.jsonl 2> gpurun_out/explore1.err
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_head.json 2> gpurun_out/bench_head.err
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_head.log 2>&1
tail -3 gpurun_out/pytest_gpu_head.log
