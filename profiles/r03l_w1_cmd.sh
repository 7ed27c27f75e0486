for c in 21 22 23; do
MASTER_ADDR=127.0.0.1 MASTER_PORT=2953$((c-20)) RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --shard-chunk $((1<<c)) 2>/dev/null | grep -v amdgpu.ids > gpurun_out/w1_$c.json
python - <<PY
import json
for l in open('gpurun_out/w1_$c.json'):
    if l.startswith('{'):
        d=json.loads(l); print($c, d['value'], d['ms_per_step'], d['config'].get('levels'), d['config'].get('verdict'))
PY
done
