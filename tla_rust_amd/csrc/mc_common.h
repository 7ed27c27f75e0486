// mc_common.h — shared host/device definitions of the model-checking engine.
//
// Spec lowerings (spec_*.h) are written once as MC_HD functions: hipcc compiles them into the
// gfx950 kernels of engine.hip, and the host pass compiles them for trace reconstruction and
// state formatting (and tests/_shim builds them alone, without HIP, to compare the lowering
// against the oracle on a CPU-only machine).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif

namespace mc {

// status bits returned by Spec::eval for one (state, slot) pair
enum : unsigned {
    ST_ENABLED = 1u,       // the action's guard holds: one successor is generated
    ST_OUT_OF_MODEL = 2u,  // successor violates the CONSTRAINT: generated, not stored
    ST_ASSERT = 4u,        // Assert(...) evaluated to FALSE while generating it
    ST_INVARIANT = 8u,     // successor violates an INVARIANT (index in bits 8..15)
    ST_SPECERR = 16u,      // TLC would raise an evaluation error
    ST_OVERFLOW = 32u,     // a fixed-capacity slot array of the packed state is full
    ST_SELFLOOP = 64u      // the successor IS the expanded state (a stuttering step of the action): generated and counted, but
                           // already in the seen-set by construction, so it needs neither a fingerprint nor a probe
};

// Strided view of one packed state.  In the HBM arena states are stored in blocks of 64,
// word-major inside a block (word w of state (b,l) at ((b*WORDS + w)*64 + l)), so that a
// wavefront whose lane l works on state (b,l) reads and writes 512 contiguous bytes per
// word access.  Host buffers and exchange records are plain (stride 1).
struct WordRef {
    uint64_t *p;
    size_t stride;
    MC_HD uint64_t get(int w) const { return p[(size_t)w * stride]; }
    // MC_NT_ROWSTORE (A/B, device code): state rows are written once and read a BFS level later — stream them past the L2 (`nt`), so
    // that they do not push out the parent rows the in-wave writer re-reads
    MC_HD void set(int w, uint64_t v) const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(MC_NT_ROWSTORE) && MC_NT_ROWSTORE
        __builtin_nontemporal_store(v, p + (size_t)w * stride);
#else
        p[(size_t)w * stride] = v;
#endif
    }
};
struct CWordRef {
    const uint64_t *p;
    size_t stride;
    MC_HD uint64_t get(int w) const { return p[(size_t)w * stride]; }
};

// Writers that re-read a parent row from the arena (Spec::apply*): ask for ALL its words at once, before the first use.  The
// copy-and-patch code below it reads the row group by group with stores in between (source and destination are the same arena,
// so the compiler keeps every load behind every earlier store): five or six dependent round trips to L2 / the Infinity Cache
// per batch of 64 new states, and a wavefront that writes states does little else — the writer was bound by that latency, not
// by its instructions (round 4: 41 of 152 ms of the expand kernel).  After this pass the groups hit the CU's L1.
template <class Ref>
MC_HD void prefetch_row(Ref s, int words) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t acc = 0;
#pragma unroll
    for (int w0 = 0; w0 < 64; w0 += 16) {
        if (w0 < words) {
            uint64_t t[16];
#pragma unroll
            for (int u = 0; u < 16; u++) t[u] = s.get(w0 + u < words ? w0 + u : words - 1);
#pragma unroll
            for (int u = 0; u < 16; u++) acc ^= t[u];
        }
    }
    asm volatile("" ::"v"((uint32_t)acc), "v"((uint32_t)(acc >> 32)));  // the loads are kept although nothing uses their values
#else
    (void)s; (void)words;
#endif
}

MC_HD uint64_t fmix64(uint64_t h) {
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}
// H(x, salt): contribution of one element to the additive (multiset) fingerprint.
MC_HD uint64_t hmix(uint64_t x, uint64_t salt) { return fmix64(x ^ salt); }
// hmum: H for lowerings whose fingerprint is a SUM of many per-element terms recomputed for every successor (raft).  fmix64
// costs two 64-bit multiplies = six quarter-rate 32-bit multiplies on gfx950 (v_mul_lo_u32 / v_mad_u64_u32 issue at 1/4
// rate): ~40 % of the VALU time of the raft expand kernel.  This mix is two 32 x 32 -> 64 multiply-accumulates (one
// v_mad_u64_u32 each; the wyhash "mum" step on 32-bit halves): round 1 multiplies the low half by (high half ^ rotated low
// half) — quadratic in the low half even when the high half is constant, as for every word narrower than 32 bits — and
// adds the input back (a zero factor cannot erase it); round 2 multiplies the two halves of that sum and adds it back; the
// fold whitens the top bits of the product (biased towards 0) with its middle bits.  tests/test_hash_quality.py: avalanche
// (every input bit flips every output bit with p = 0.5 +- 0.015), no 64-bit collision and birthday-rate 32-bit collisions
// on 2^24 structured words, on 8.4 M pair sums and 16.8 M differences of structured words (the additive use).
// Not a bijection: fmix64 stays where the fingerprint must be one (one-word specs).
MC_HD uint64_t hmum(uint64_t x, uint64_t salt) {
    x ^= salt;
    const uint32_t a = (uint32_t)x, b = (uint32_t)(x >> 32);
    const uint64_t p = (uint64_t)a * (uint64_t)(b ^ ((a << 16) | (a >> 16)) ^ 0x74743c1bu) + x;
    const uint64_t q = (uint64_t)((uint32_t)p ^ 0x53c5ca59u) * (uint64_t)((uint32_t)(p >> 32) ^ 0x2d358dccu) + p;
    uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    lo ^= hi;
    hi ^= (lo << 13) | (lo >> 19);
    return ((uint64_t)hi << 32) | lo;
}
// fingerprint 0 is the seen-set's EMPTY marker
MC_HD uint64_t fp_nonzero(uint64_t fp) { return fp ? fp : 0x9e3779b97f4a7c15ull; }

MC_HD uint64_t salt_of(unsigned i) { return 0x9e3779b97f4a7c15ull * (2ull * i + 1ull) + 0x632be59bd9b4e019ull; }

MC_HD uint32_t fp_owner(uint64_t fp, uint32_t shards) {
    // high bits pick the owner, low bits index the owner's table
    return shards <= 1 ? 0u : (uint32_t)(((fp >> 40) * (uint64_t)shards) >> 24);
}

MC_HD uint64_t bits_get(uint64_t w, int pos, int n) { return (w >> pos) & ((1ull << n) - 1ull); }
MC_HD uint64_t bits_set(uint64_t w, int pos, int n, uint64_t v) {
    const uint64_t m = ((1ull << n) - 1ull) << pos;
    return (w & ~m) | ((v << pos) & m);
}

}  // namespace mc
