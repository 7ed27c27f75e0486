# the contract invocation at full size with several ranks, all on ONE GPU through the librccl stand-in (a functional run of the N-rank path
# on the default workload: golden gate, shares, frontier balance — NOT a scaling measurement)
cd /root/repo
export TLAMC_RCCL=$(python -c "import sys; sys.path.insert(0,'tests'); import helpers; print(helpers.build_fakerccl())")
for n in 2 8; do
timeout 900 python bench.py --gpus $n --share-gpu --steps 2 --warmup 1 2>gpurun_out/share_$n.err | grep -v amdgpu.ids > gpurun_out/r03n_bench_share_gpu_$n.json
python - <<PY
import json
for l in open('gpurun_out/r03n_bench_share_gpu_$n.json'):
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print($n, d['value'], d['ms_per_step'], c.get('verdict'), c.get('shares'), c.get('levels'), round(c.get('frontier_imbalance',0),3), d.get('xgmi',{}).get('sent_bytes_per_step_per_gpu'))
PY
tail -n 2 gpurun_out/share_$n.err
done
