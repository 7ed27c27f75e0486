#!/bin/bash
# A/B build: libtlamc_ssiab.so = the library with the SSI translation unit (MC_TU = 5) compiled with extra defines ($1, e.g. -DMC_PAIR_COMPACT=0, or a macro of an experiment's working tree).
# Loaded with TLAMC_LIB=tla_rust_amd/_build/libtlamc_ssiab.so; never the product library.
set -e
cd "$(dirname "$0")/.."
B=tla_rust_amd/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I include -x hip -DMC_TU=5 $1 \
    -c tla_rust_amd/csrc/engine.hip -o $B/engine_tu5_ab.o
OBJS=$(ls $B/*.o | grep -v "engine_tu5.o\|engine_tu5_\|_prof.o\|_ab.o\|engine_tu1_f")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libtlamc_ssiab.so $OBJS $B/engine_tu5_ab.o -ldl -pthread
ls -la $B/libtlamc_ssiab.so
