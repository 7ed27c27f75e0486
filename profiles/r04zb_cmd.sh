# round 4, GPU call zb (the round's last 5.9 GPU-minutes): the three forms of a stay level's exchange on the GPU — exact sizes (the new
# default: mc_shard_expand_finish + _probe + _keep_slot), fixed-capacity buckets from measured fill / from packed_fanout — through both
# doors (torch + gloo staging, native loop over the nccl* stand-in), then the contract invocation with 2 ranks (xgmi object: sent vs needed)
cd /root/repo; D=gpurun_out/r04zb; mkdir -p $D
timeout 160 python -m pytest tests/test_gpu_sharded.py -x -q -k "three_forms or exchange_forms or stay_mode or full_exchange or eight_engines or replicated_prefix or (native_rccl_loop_with and 2) or (checkpoint_per_rank and 2)" > $D/pytest_exchange.log 2>&1; echo rc=$? >> $D/pytest_exchange.log; tail -6 $D/pytest_exchange.log | cut -c1-900
export TLAMC_RCCL=$PWD/tests/_fakerccl/_build/libfakerccl.so
for x in exact measured; do timeout 50 python bench.py --gpus 2 --share-gpu --steps 1 --warmup 0 --workload k10 --exchange $x --no-cpu-baseline 2>$D/bench_$x.err | grep '"metric"' > $D/bench_share_gpu_2_$x.json; python -c "
import json; d=json.loads(open('$D/bench_share_gpu_2_$x.json').read()); x=d['xgmi']; print('$x', d['ms_per_step'], x['sent_over_model'], x['fp_answer_bytes_per_step'], x['model_bytes_per_step'], d['config']['levels'])" 2>&1 | tail -1; done
