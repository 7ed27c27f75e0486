#!/bin/bash
# profiles/link_variant.sh TAG TU — libtlamc_<TAG>.so = the product library with translation unit TU replaced by
# tla_rust_amd/_build/engine_tu<TU>_v_<TAG>.o (an A/B build of that unit with extra -D flags); loaded through $TLAMC_LIB.  Never the product library.
set -e
cd "$(dirname "$0")/.."
B=tla_rust_amd/_build
OBJS=$(ls $B/*.o | grep -v "engine_tu$2.o\|engine_tu[0-9]_v_\|_prof.o\|engine_tu1_f")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libtlamc_$1.so $OBJS $B/engine_tu$2_v_$1.o -ldl
