#!/bin/bash
# profiles/build_variants.sh TAG:"-Dflags" ... — A/B libraries: libtlamc_<TAG>.so = the product library with the 3-server raft
# translation unit (MC_TU = 3: the bench kernels; $TU = 4: the 5-server kernels of `--workload raft5`) recompiled with extra -D flags;
# loaded through TLAMC_LIB.  Never the product library.
#   profiles/build_variants.sh a2:"-DMC_ASYNC_PROBE=2" w2:"-DMC_EXPAND_WAVES=2"
set -e
cd "$(dirname "$0")/.."
python -c "import tla_rust_amd.build as b; b.build()"
B=tla_rust_amd/_build
TU=${TU:-3}
pids=()
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result -I include -x hip -DMC_TU=$TU $defs \
      -c tla_rust_amd/csrc/engine.hip -o $B/engine_tu${TU}_v_$tag.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for spec in "$@"; do
  tag=${spec%%:*}
  OBJS=$(ls $B/*.o | grep -v "engine_tu$TU.o\|engine_tu[0-9]_v_\|_prof.o\|engine_tu1_f")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libtlamc_$tag.so $OBJS $B/engine_tu${TU}_v_$tag.o -ldl
  python profiles/kres.py $B/engine_tu${TU}_v_$tag.o "SpecRaftILi$([ $TU = 4 ] && echo 5 || echo 3)EEELb0ELi1ELi4" | sed "s/^/$tag: /"
done
