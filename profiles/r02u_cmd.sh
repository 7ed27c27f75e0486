# round 2, GPU call u: `mc specs/MCraft.tla -gpus 1` (native hip-rccl back-end, world 1) on the bench model vs the one-GPU mc
cd /root/repo; mkdir -p gpurun_out/r02u; export TLAMC_UNVERIFIED=1
for i in 1 2; do ./tla_rust_amd/_build/mc specs/MCraft.tla -config specs/MCraft.cfg -gpus 1 -tablelog2 27 -arena 104000000 -chunk 2097152 >> gpurun_out/r02u/mc_gpus1_native.log 2>&1; done
./tla_rust_amd/_build/mc specs/MCraft.tla -config specs/MCraft.cfg -noprogress -tablelog2 27 -arena 104000000 -chunk 4194304 >> gpurun_out/r02u/mc_one_gpu.log 2>&1
grep -E "states generated|RCCL|depth|error|Error|mc\[" gpurun_out/r02u/mc_gpus1_native.log; tail -4 gpurun_out/r02u/mc_one_gpu.log
