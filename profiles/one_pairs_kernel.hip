// profiles/one_pairs_kernel.hip — compile ONE instantiation of the by-pairs kernel (engine_pairs.h, the SSI model) by itself: a
// 20-second register / code-size / ISA check while working on the kernel, instead of the four minutes the SSI translation unit takes
// (its slot-by-slot kernels unroll 77 slots).  -DONE_CHECK: k_check_frontier<SpecSsi> too.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -c profiles/one_pairs_kernel.hip -o /tmp/k.o -save-temps=obj && python profiles/kres.py /tmp/k.o expand_pairs
#define MC_TU 99
#include "../tla_rust_amd/csrc/engine.hip"
template __global__ void mc::k_expand_pairs<mc::SpecSsi, MC_PAIR_WAVES>(mc::SpecSsi::Params, const uint64_t *, uint64_t, uint64_t, uint64_t, uint64_t *, uint64_t,
                                                                        mc::DevCounters *, unsigned, mc::RouteArgs);
template __global__ void mc::k_check_frontier<mc::SpecSsi>(mc::SpecSsi::Params, const uint64_t *, uint64_t, uint64_t, mc::DevCounters *);
