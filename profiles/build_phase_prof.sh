#!/bin/bash
# profiling build: libtlamc_prof.so = the library with the raft-3 kernels compiled under -DMC_PHASE_PROF (per-phase cycle counters in
# k_expand_family).  Loaded with TLAMC_LIB=tla_rust_amd/_build/libtlamc_prof.so (profiles/phase_prof.py); never the product library.
set -e
cd "$(dirname "$0")/.."
python -c "import tla_rust_amd.build as b; b.build()"
B=tla_rust_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I include -x hip -DMC_TU=3 -DMC_PHASE_PROF \
    -c tla_rust_amd/csrc/engine.hip -o $B/engine_tu3_prof.o
OBJS=$(ls $B/*.o | grep -v "engine_tu3.o\|engine_tu3_\|_prof.o\|engine_tu1_f")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libtlamc_prof.so $OBJS $B/engine_tu3_prof.o -ldl
ls -la $B/libtlamc_prof.so
