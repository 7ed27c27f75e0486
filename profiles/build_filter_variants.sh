#!/bin/bash
# A/B builds: libtlamc_f<N>.so = the library with atomic_add's slot-by-slot kernel carrying a per-wavefront duplicate filter of N entries
# (TU 1 compiled with -DMC_ATOMIC_ADD_FILTER=N); loaded with TLAMC_LIB.  Never the product library.
set -e
cd "$(dirname "$0")/.."
B=tla_rust_amd/_build
for N in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I include -x hip -DMC_TU=1 -DMC_ATOMIC_ADD_FILTER=$N \
      -c tla_rust_amd/csrc/engine.hip -o $B/engine_tu1_f$N.o
  OBJS=$(ls $B/*.o | grep -v "engine_tu1.o\|engine_tu1_f\|_prof.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libtlamc_f$N.so $OBJS $B/engine_tu1_f$N.o -ldl
done
ls -la $B/libtlamc_f*.so
