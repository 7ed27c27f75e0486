// profiles/calib/calib_fetch.hip — known-byte access patterns for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access
// widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// The four patterns are the engine's own (engine.hip):
//   k_stream16   16 B / lane coalesced streaming read (the guide's reference pattern)
//   k_rows8       8 B / lane, 512-byte rows of a word-major arena block (k_expand_*: parent words; k_materialise: parent copy)
//   k_bucket64   one random 64-byte seen-set bucket per lane, four 16-byte loads (seen_insert)
//   k_write8      8 B / lane row writes (k_materialise: new states)
// Each kernel touches a known number of bytes far beyond the 256 MiB Infinity Cache; run under
//   rocprofv3 --pmc FETCH_SIZE   and   rocprofv3 --pmc WRITE_SIZE   (separate passes) and compare.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) k_stream16(const ulonglong2 *__restrict__ p, uint64_t n16, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const ulonglong2 v = p[i]; acc += v.x ^ v.y; }
    if (acc == 0x1234567ull) *sink = acc;
}
__global__ void __launch_bounds__(256) k_rows8(const uint64_t *__restrict__ p, uint64_t n8, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x1234567ull) *sink = acc;
}
__global__ void __launch_bounds__(256) k_bucket64(const uint64_t *__restrict__ table, uint64_t mask, uint64_t nprobes, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nprobes; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t h = (i + 1) * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + ((h & mask) & ~7ull));
        const ulonglong2 a = line[0], b = line[1], c = line[2], d = line[3];
        acc += a.x ^ a.y ^ b.x ^ b.y ^ c.x ^ c.y ^ d.x ^ d.y;
    }
    if (acc == 0x1234567ull) *sink = acc;
}
__global__ void __launch_bounds__(256) k_write8(uint64_t *__restrict__ p, uint64_t n8) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) p[i] = i;
}

int main() {
    const uint64_t bytes = 8ull << 30, tbytes = 2ull << 30, nprobes = 1ull << 28;
    uint64_t *buf = nullptr, *table = nullptr;
    unsigned long long *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&table, tbytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipMemset(table, 1, tbytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](const char *name, double known_bytes, auto &&launch) {
        launch();  // warm-up (also excluded from nothing: the profiler sees both launches, the summary divides by 2)
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("{\"kernel\": \"%s\", \"known_bytes_per_launch\": %.0f, \"ms\": %.3f, \"GBs\": %.1f}\n", name, known_bytes, ms, known_bytes / ms / 1e6);
    };
    const dim3 grid(256 * 16), block(256);
    timed("k_stream16", (double)bytes, [&] { hipLaunchKernelGGL(k_stream16, grid, block, 0, 0, (const ulonglong2 *)buf, bytes / 16, sink); });
    timed("k_rows8", (double)bytes, [&] { hipLaunchKernelGGL(k_rows8, grid, block, 0, 0, (const uint64_t *)buf, bytes / 8, sink); });
    timed("k_bucket64", (double)nprobes * 64.0, [&] { hipLaunchKernelGGL(k_bucket64, grid, block, 0, 0, (const uint64_t *)table, tbytes / 8 - 1, nprobes, sink); });
    timed("k_write8", (double)bytes, [&] { hipLaunchKernelGGL(k_write8, grid, block, 0, 0, buf, bytes / 8); });
    hipDeviceSynchronize();
    return 0;
}
