// profiles/calib/probe_locality.hip — what a seen-set probe costs as a function of how much table it can land in, and what
// partitioning the candidates by table region costs on top.  NOT part of the product: a measurement aid for the question the
// round-3 VERDICT asks about the synthetic spec (atomic_add, W = 8 B: the 32-byte random probe IS the workload) —
//
//   "Cache-partitioned seen-set probing: per round, emit candidates (fp, src) to a buffer, radix-partition by home-bucket region
//    sized to the 256 MiB Infinity Cache (or the 4 MiB L2 of an XCD), probe partition by partition.  Extra streaming ~ 2 x 12 B
//    per candidate at 4+ TB/s against a 32-B random read at 1.2 TB/s."
//
// and about N = 30's halved probe rate (16.1 G probes in 865 ms against 37 G/s at N = 28: a 25.8 GB table against 8 GB).
//
// Three measurements, one JSON line each (HIP events around the second of two launches):
//   1. k_probe32 / k_probe64: P random 32-byte (4-slot) / 64-byte (8-slot) bucket reads into a table REGION of S bytes, S from
//      1 MiB to the whole table — the random-read rate against the working set: L2 (4 MiB per XCD), Infinity Cache (256 MiB),
//      HBM, and beyond the reach of the address translation caches.  The reads are the engine's (seen_insert: 16-byte loads of one
//      bucket), the addresses a multiply-shift of a mixed counter like its home-bucket function.
//   2. k_insert32: the same with the compare-and-swap of a first-time insert into an empty region (the write side of a new state).
//   2b. k_probe_ilp<1 | 2 | 4>: probes in flight per lane against the wavefronts per SIMD (capped through the workgroup's LDS).
//   3. partition: the candidates of one round (fp 8 B + src 4 B) counting-sorted by table region — k_hist (read 8 B) + k_scatter
//      (read 12 B, write 12 B) — and then probed region by region (k_probe_part): the end-to-end rate of the partitioned scheme
//      against measurement 1 at S = the whole table, for region sizes 4 MiB ... 256 MiB.
//
//   hipcc --offload-arch=gfx950 -O3 -o probe_locality probe_locality.hip && ./probe_locality [table_GiB = 8] [log2_probes = 28]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t i) {
    uint64_t h = (i + 1) * 0x9e3779b97f4a7c15ull;
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
    return h;
}
// home bucket inside a region of `buckets` buckets: multiply-shift of the low 32 bits (the engine's form for tables of any size)
__device__ __forceinline__ uint64_t home(uint64_t h, uint64_t buckets) { return ((h & 0xffffffffull) * buckets) >> 32; }

// BUCKET_WORDS = 4 (32-byte probes: tables at most a third full) or 8 (64-byte probes)
template <int BUCKET_WORDS>
__global__ void __launch_bounds__(256) k_probe(const uint64_t *__restrict__ table, uint64_t buckets, uint64_t nprobes, uint64_t salt,
                                               unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nprobes; i += (uint64_t)gridDim.x * blockDim.x) {
        const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + home(mix(i ^ salt), buckets) * BUCKET_WORDS);
#pragma unroll
        for (int k = 0; k < BUCKET_WORDS / 2; ++k) { const ulonglong2 v = line[k]; acc += v.x ^ v.y; }
    }
    if (acc == 0x1234567ull) *sink = acc;
}
// first-time insert: read the bucket, CAS the fingerprint into its first empty slot (the region starts empty and stays sparse)
__global__ void __launch_bounds__(256) k_insert32(uint64_t *__restrict__ table, uint64_t buckets, uint64_t nprobes, uint64_t salt,
                                                  unsigned long long *sink) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nprobes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t fp = mix(i ^ salt) | 1ull;
        uint64_t *b = table + home(fp, buckets) * 4;
        const ulonglong2 v0 = reinterpret_cast<const ulonglong2 *>(b)[0], v1 = reinterpret_cast<const ulonglong2 *>(b)[1];
        const uint64_t w[4] = {v0.x, v0.y, v1.x, v1.y};
        bool done = false;
#pragma unroll
        for (int k = 0; k < 4 && !done; ++k) {
            if (w[k] == fp) done = true;
            else if (w[k] == 0) {
                const unsigned long long old = atomicCAS((unsigned long long *)(b + k), 0ull, (unsigned long long)fp);
                done = old == 0 || old == fp;
                acc += done;
            }
        }
    }
    if (acc == 0x1234567ull) *sink = acc;
}

// ---- probes in flight: ILP independent 32-byte probes issued before the first is consumed, at an occupancy capped through the
// workgroup's dynamic LDS (160 KB per CU: lds_bytes = 160 KB / workgroups per CU; a 256-thread workgroup is one wavefront per SIMD) —
// what the by-family expand kernel of the raft model has (4 wavefronts per SIMD, one probe batch in flight) against what a fifth
// wavefront or a second probe batch would buy
template <int ILP>
__global__ void __launch_bounds__(256) k_probe_ilp(const uint64_t *__restrict__ table, uint64_t buckets, uint64_t nprobes, uint64_t salt,
                                                   unsigned long long *sink) {
    extern __shared__ unsigned char occupancy_pad[];
    unsigned long long acc = 0;
    if (salt == ~0ull) acc = occupancy_pad[threadIdx.x];  // (keeps the allocation alive)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + (ILP - 1) * stride < nprobes; i += ILP * stride) {
        ulonglong2 a[ILP], b[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + home(mix((i + k * stride) ^ salt), buckets) * 4);
            a[k] = line[0];
            b[k] = line[1];
        }
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc += a[k].x ^ a[k].y ^ b[k].x ^ b[k].y;
    }
    if (acc == 0x1234567ull) *sink = acc;
}

// ---- partition by table region: region = home bucket / buckets_per_region
__global__ void __launch_bounds__(256) k_make(uint64_t *__restrict__ fp, uint32_t *__restrict__ src, uint64_t n, uint64_t salt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { fp[i] = mix(i ^ salt) | 1ull; src[i] = (uint32_t)i; }
}
constexpr int MAXR = 4096;  // regions
__global__ void __launch_bounds__(256) k_hist(const uint64_t *__restrict__ fp, uint64_t n, uint64_t buckets, uint64_t per_region, unsigned nreg,
                                              unsigned long long *__restrict__ hist) {
    __shared__ unsigned lh[MAXR];
    for (unsigned r = threadIdx.x; r < nreg; r += blockDim.x) lh[r] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&lh[(unsigned)(home(fp[i], buckets) / per_region)], 1u);
    __syncthreads();
    for (unsigned r = threadIdx.x; r < nreg; r += blockDim.x) if (lh[r]) atomicAdd(&hist[r], (unsigned long long)lh[r]);
}
// block-local counting sort into the regions' ranges: one global atomic per (block tile, region), writes in runs
__global__ void __launch_bounds__(256) k_scatter(const uint64_t *__restrict__ fp, const uint32_t *__restrict__ src, uint64_t n, uint64_t buckets,
                                                 uint64_t per_region, unsigned nreg, unsigned long long *__restrict__ cursor,
                                                 uint64_t *__restrict__ out_fp, uint32_t *__restrict__ out_src) {
    __shared__ unsigned cnt[MAXR];
    __shared__ unsigned long long base[MAXR];
    constexpr int TILE = 256 * 16;
    for (uint64_t t0 = (uint64_t)blockIdx.x * TILE; t0 < n; t0 += (uint64_t)gridDim.x * TILE) {
        for (unsigned r = threadIdx.x; r < nreg; r += blockDim.x) cnt[r] = 0;
        __syncthreads();
        unsigned reg[16], pos[16];
        uint64_t f[16];
        uint32_t s[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t i = t0 + (uint64_t)k * 256 + threadIdx.x;
            reg[k] = ~0u;
            if (i < n) { f[k] = fp[i]; s[k] = src[i]; reg[k] = (unsigned)(home(f[k], buckets) / per_region); pos[k] = atomicAdd(&cnt[reg[k]], 1u); }
        }
        __syncthreads();
        for (unsigned r = threadIdx.x; r < nreg; r += blockDim.x) if (cnt[r]) base[r] = atomicAdd(&cursor[r], (unsigned long long)cnt[r]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (reg[k] != ~0u) { const unsigned long long o = base[reg[k]] + pos[k]; out_fp[o] = f[k]; out_src[o] = s[k]; }
        __syncthreads();
    }
}
// probe the partitioned candidates in order: consecutive workgroups work on the same region of the table
__global__ void __launch_bounds__(256) k_probe_part(const uint64_t *__restrict__ table, uint64_t buckets, const uint64_t *__restrict__ fp, uint64_t n,
                                                    unsigned long long *sink) {
    unsigned long long acc = 0;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const ulonglong2 *line = reinterpret_cast<const ulonglong2 *>(table + home(fp[i], buckets) * 4);
        const ulonglong2 a = line[0], b = line[1];
        acc = a.x ^ a.y ^ b.x ^ b.y;
    }
    if (acc == 0x1234567ull) *sink = acc;
}

int main(int argc, char **argv) {
    const uint64_t table_gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8, lg = argc > 2 ? strtoull(argv[2], nullptr, 10) : 28;
    const uint64_t tbytes = table_gib << 30, nprobes = 1ull << lg;
    uint64_t *table = nullptr, *fp = nullptr, *pfp = nullptr;
    uint32_t *src = nullptr, *psrc = nullptr;
    unsigned long long *sink = nullptr, *hist = nullptr;
    CK(hipMalloc(&table, tbytes)); CK(hipMalloc(&sink, 8)); CK(hipMalloc(&hist, 2 * MAXR * sizeof(unsigned long long)));
    CK(hipMalloc(&fp, nprobes * 8)); CK(hipMalloc(&pfp, nprobes * 8)); CK(hipMalloc(&src, nprobes * 4)); CK(hipMalloc(&psrc, nprobes * 4));
    CK(hipMemset(table, 0, tbytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(256 * 32), block(256);
    auto timed = [&](auto &&launch) {
        launch();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        return (double)ms;
    };
    // 1. random bucket reads against the working set
    uint64_t sizes[24];
    int nsizes = 0;
    for (uint64_t s = 1ull << 20; s < tbytes; s *= 4) sizes[nsizes++] = s;
    sizes[nsizes++] = tbytes;
    for (int q = 0; q < nsizes; ++q) {
        const uint64_t s = sizes[q];
        const double ms32 = timed([&] { hipLaunchKernelGGL(k_probe<4>, grid, block, 0, 0, table, s / 32, nprobes, 7ull, sink); });
        const double ms64 = timed([&] { hipLaunchKernelGGL(k_probe<8>, grid, block, 0, 0, table, s / 64, nprobes, 9ull, sink); });
        printf("{\"what\": \"random bucket reads\", \"region_MiB\": %.0f, \"probes\": %llu, \"ms_32B\": %.3f, \"Gprobes_s_32B\": %.2f, \"GBs_32B\": %.0f, "
               "\"ms_64B\": %.3f, \"Gprobes_s_64B\": %.2f, \"GBs_64B\": %.0f}\n", (double)s / (1 << 20), (unsigned long long)nprobes, ms32, nprobes / ms32 / 1e6,
               nprobes * 32.0 / ms32 / 1e6, ms64, nprobes / ms64 / 1e6, nprobes * 64.0 / ms64 / 1e6);
        fflush(stdout);
    }
    // 2. first-time inserts (read + CAS) into an empty region: the whole table, and one Infinity-Cache-sized region
    for (uint64_t s : {tbytes, (uint64_t)128 << 20}) {
        CK(hipMemset(table, 0, s));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_insert32, grid, block, 0, 0, table, s / 32, s == tbytes ? nprobes : (s / 32), 11ull, sink);  // (load <= 1/4: stays sparse)
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const uint64_t n = s == tbytes ? nprobes : (s / 32);
        printf("{\"what\": \"first-time inserts (32-byte read + CAS)\", \"region_MiB\": %.0f, \"inserts\": %llu, \"ms\": %.3f, \"Ginserts_s\": %.2f}\n",
               (double)s / (1 << 20), (unsigned long long)n, ms, n / ms / 1e6);
    }
    // 2b. probes in flight against occupancy (whole table)
    for (int wg_per_cu : {3, 4, 5, 6, 8}) {
        const size_t lds = (size_t)(160 * 1024) / wg_per_cu - 512;
        const dim3 g(256 * wg_per_cu);  // one resident set: every workgroup runs from start to end
        auto run = [&](auto kern) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return timed([&] { hipLaunchKernelGGL(kern, g, block, lds, 0, table, tbytes / 32, nprobes, 17ull, sink); });
        };
        const double m1 = run(k_probe_ilp<1>), m2 = run(k_probe_ilp<2>), m4 = run(k_probe_ilp<4>);
        printf("{\"what\": \"probes in flight\", \"wavefronts_per_SIMD\": %d, \"Gprobes_s_ilp1\": %.2f, \"Gprobes_s_ilp2\": %.2f, \"Gprobes_s_ilp4\": %.2f}\n", wg_per_cu,
               nprobes / m1 / 1e6, nprobes / m2 / 1e6, nprobes / m4 / 1e6);
        fflush(stdout);
    }
    CK(hipMemset(table, 0, tbytes));
    // 3. partition the candidates of a round by table region, then probe region by region
    hipLaunchKernelGGL(k_make, grid, block, 0, 0, fp, src, nprobes, 13ull);
    CK(hipDeviceSynchronize());
    const uint64_t buckets = tbytes / 32;
    const double ms_direct = timed([&] { hipLaunchKernelGGL(k_probe_part, dim3((unsigned)((nprobes + 255) / 256)), block, 0, 0, table, buckets, fp, nprobes, sink); });
    printf("{\"what\": \"unpartitioned probes of the candidate buffer\", \"probes\": %llu, \"ms\": %.3f, \"Gprobes_s\": %.2f}\n", (unsigned long long)nprobes, ms_direct,
           nprobes / ms_direct / 1e6);
    for (uint64_t region = 4ull << 20; region <= (256ull << 20); region *= 4) {
        const uint64_t per_region = region / 32;
        const unsigned nreg = (unsigned)((buckets + per_region - 1) / per_region);
        if (nreg > MAXR) { printf("{\"what\": \"partitioned\", \"region_MiB\": %.0f, \"skipped\": \"more than %d regions\"}\n", (double)region / (1 << 20), MAXR); continue; }
        unsigned long long *cursor = hist + MAXR;
        float ms_h = 0, ms_s = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(hist, 0, 2 * MAXR * sizeof(unsigned long long)));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_hist, grid, block, 0, 0, fp, nprobes, buckets, per_region, nreg, hist);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_h, e0, e1);
            // exclusive prefix sum of the histogram on the host (nreg <= 4096 words): the regions' ranges
            unsigned long long h[MAXR], c[MAXR], sum = 0;
            CK(hipMemcpy(h, hist, nreg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (unsigned r = 0; r < nreg; ++r) { c[r] = sum; sum += h[r]; }
            if (sum != nprobes) { fprintf(stderr, "histogram does not add up: %llu\n", sum); return 1; }
            CK(hipMemcpy(cursor, c, nreg * sizeof(unsigned long long), hipMemcpyHostToDevice));
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_scatter, grid, block, 0, 0, fp, src, nprobes, buckets, per_region, nreg, cursor, pfp, psrc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_s, e0, e1);
        }
        const double ms_p = timed([&] { hipLaunchKernelGGL(k_probe_part, dim3((unsigned)((nprobes + 255) / 256)), block, 0, 0, table, buckets, pfp, nprobes, sink); });
        printf("{\"what\": \"partitioned\", \"region_MiB\": %.0f, \"regions\": %u, \"ms_hist\": %.3f, \"ms_scatter\": %.3f, \"ms_probe\": %.3f, \"ms_total\": %.3f, "
               "\"Gprobes_s_end_to_end\": %.2f, \"against_unpartitioned\": %.2f}\n", (double)region / (1 << 20), nreg, ms_h, ms_s, ms_p, ms_h + ms_s + ms_p,
               nprobes / (ms_h + ms_s + ms_p) / 1e6, ms_direct / (ms_h + ms_s + ms_p));
        fflush(stdout);
    }
    return 0;
}
