# round 5, GPU call zf (the round's last GPU seconds): the compiled-PlusCal path on its larger models, timed (profiles/bench_channels.py)
cd /root/repo; D=$PWD/gpurun_out/r05zf; mkdir -p $D
timeout 100 python profiles/bench_channels.py 2>$D/chan.err | tee $D/bench_channels.jsonl
