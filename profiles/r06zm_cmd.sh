# round 6, GPU call zm: the largest frontier a BATCHED level takes (2^16 states since round 2; $TLAMC_BLIND_LOG2): the generated PlusCal models run 37 / 105
# levels in 7.6 / 24.6 ms — 0.2 ms per level whatever the kernel's shape (call zl) — i.e. they are bound by the host round trip per level
cd /root/repo; D=$PWD/gpurun_out/r06zm; mkdir -p $D
timeout 1700 python profiles/blind_ab.py 2>$D/err.log | tee $D/blind_ab.jsonl
tail -n 3 $D/err.log
