# round 6, GPU call zl: the shape of the by-pairs kernel in a generated PlusCal unit (206 VGPRs at 2 wavefronts per SIMD, one wavefront per workgroup):
# MC_PAIR_MINW 2 / 3 / 4 x MC_PAIR_WAVES 1 / 2 / 4 on pagecache N = 3 and ms_queue_counted N = 3 K = 3 (profiles/jit_defs_ab.py)
cd /root/repo; D=$PWD/gpurun_out/r06zl; mkdir -p $D
timeout 1500 python profiles/jit_defs_ab.py 2>$D/err.log | tee $D/jit_defs_ab.jsonl
tail -n 3 $D/err.log
