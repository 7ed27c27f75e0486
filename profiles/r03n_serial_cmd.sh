# stand-alone kernel times of the final state (TLAMC_SERIAL=1: no expand kernel beside a materialise), t3 and K = 10; then the same with
# small chunks (does a materialise that follows its expand closely find the parent rows in the 256 MB Infinity Cache?  no: 45-48 ms
# at 2^19 .. 2^21 states per launch against 44.0 at 2^23, and the expand kernel loses to the per-launch tails)
for w in t3 k10; do TLAMC_SERIAL=1 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r03n_${w}_serial.json; done
for c in 19 20 21; do TLAMC_SERIAL=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --chunk $((1<<c)) 2>/dev/null | grep -v amdgpu.ids > gpurun_out/serial_c$c.json; done
