// profiles/one_kernel.hip — compile ONE instantiation of the by-family expand kernel (the fused form of the 3-server raft model) by
// itself: a 20-second register / spill / code-size check while working on the kernel (profiles/kres.py on the object), instead of
// the minute a whole translation unit of engine.hip takes.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c profiles/one_kernel.hip -o /tmp/k.o [-DMC_ASYNC_PROBE=2 ...] && python profiles/kres.py /tmp/k.o
#define MC_TU 99
#include "../tla_rust_amd/csrc/engine.hip"
#ifndef ONE_ROUTE
#define ONE_ROUTE false
#endif
template __global__ void mc::k_expand_family<mc::SpecRaft<3>, ONE_ROUTE, 1, MC_EXPAND_MINW, MC_EXPAND_WAVES>(
    mc::SpecRaft<3>::Params, const uint64_t *, uint64_t, uint64_t, uint64_t, uint64_t *, uint64_t, uint32_t *, uint64_t, mc::DevCounters *, unsigned,
    mc::RouteArgs, unsigned);
