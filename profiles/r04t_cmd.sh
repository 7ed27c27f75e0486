# round 4, GPU call t: atomic_add N = 28 with a per-wavefront duplicate filter of 256 / 512 / 1024 entries in the slot-by-slot kernel, now that the
# arena allocation no longer masks it (r04h measured it under the atomicAdd-per-wavefront regression)
cd /root/repo; D=gpurun_out/r04t; mkdir -p $D
for v in "" _f256 _f512 _f1024; do
TLAMC_LIB=$PWD/tla_rust_amd/_build/libtlamc$v.so timeout 300 python -c "
import sys, json; sys.path.insert(0, '.')
import bench, tla_rust_amd as amd
from tla_rust_amd import binding as B
for n in (24, 28):
    d = bench.atomic_add_series(amd, 0, n=n)
    print(json.dumps(dict(lib='$v', n=n, ms=round(d['ms_per_step'],2), kernel_ms=d['roofline']['kernel_ms'], frac=round(d['roofline']['frac'],4))))
" 2>&1 | grep -v amdgpu.ids | tee -a $D/atomic_add_filter_ab.jsonl
done
