// profiles/calib/calib_blockrows.hip — PROFILING AID (not part of the product): how fast can a wavefront-per-arena-block kernel
// read the arena?  k_expand_family's phase A reads, per wavefront, ONE block of 64 states = W rows of 512 bytes (W = 36: 18 KB,
// contiguous) and measured 1.5 TB/s against 5.4 TB/s for a grid-stride streaming read of the same bytes (calib_fetch.hip k_rows8).
// Variants: loads in G groups (a group = loads issued back to back, then one wait); LDS per workgroup and a register budget like
// the real kernel's (occupancy 4 waves / SIMD) or none; one block per wavefront or several (persistent wavefronts).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr int W = 36;
using GlobalWords = const __attribute__((address_space(1))) uint64_t *;
__device__ __forceinline__ GlobalWords uniform_ptr(const uint64_t *p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (GlobalWords)(((uint64_t)hi << 32) | lo);
}

template <int GROUPS, int LDS_BYTES, int BPW>
__global__ void __launch_bounds__(256, 4) k_blockrows(const uint64_t *__restrict__ arena, uint64_t nblocks, unsigned long long *sink) {
    __shared__ uint64_t pad[LDS_BYTES / 8 + 1];
    const unsigned lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int r = 0; r < BPW; ++r) {
        const uint64_t b = wave * BPW + r;
        if (b >= nblocks) break;
        GlobalWords base = uniform_ptr(arena + b * (uint64_t)W * 64);
        constexpr int PER = W / GROUPS;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            uint64_t x[PER];
#pragma unroll
            for (int w = 0; w < PER; ++w) x[w] = base[(unsigned)(g * PER + w) * 64u + lane];
#pragma unroll
            for (int w = 0; w < PER; ++w) acc += x[w];
            if (GROUPS > 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (LDS_BYTES > 8) pad[threadIdx.x] = acc;
    if (acc == 0x1234567ull) *sink = acc + pad[0];
}

int main() {
    const uint64_t nblocks = 1600000;  // 102.4 M states of 288 B = 29.5 GB
    const uint64_t bytes = nblocks * W * 64 * 8;
    uint64_t *buf = nullptr;
    unsigned long long *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char *name, auto &&launch) {
        launch();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("{\"kernel\": \"%s\", \"bytes\": %.0f, \"ms\": %.3f, \"GBs\": %.1f}\n", name, (double)bytes, ms, bytes / ms / 1e6);
    };
    const dim3 block(256);
#define RUN(G, L, B) timed("k_blockrows<groups=" #G ",lds=" #L ",blocks_per_wave=" #B ">", [&] { hipLaunchKernelGGL((k_blockrows<G, L, B>), dim3((unsigned)((nblocks + 4 * B - 1) / (4 * B))), block, 0, 0, buf, nblocks, sink); })
    RUN(1, 8, 1);
    RUN(2, 8, 1);
    RUN(2, 27680, 1);
    RUN(1, 27680, 1);
    RUN(4, 27680, 1);
    RUN(2, 27680, 4);
    RUN(2, 27680, 16);
    RUN(1, 8, 16);
    // chunked launches like the engine (4 M states = 65536 blocks per launch)
    timed("k_blockrows<2,27680,1> in 25 launches of 65536 blocks", [&] {
        for (uint64_t b0 = 0; b0 < nblocks; b0 += 65536) {
            const uint64_t nb = nblocks - b0 < 65536 ? nblocks - b0 : 65536;
            hipLaunchKernelGGL((k_blockrows<2, 27680, 1>), dim3((unsigned)((nb + 3) / 4)), block, 0, 0, buf + b0 * W * 64, nb, sink);
        }
    });
    hipDeviceSynchronize();
    return 0;
}
