// snn_rng.hpp -- device-resident, bit-exact emulation of the stream torch.multinomial consumes
// on the CPU generator (verified in tests/ against torch and glibc):
//
//   torch CPU generator      = mt19937 (at::mt19937: 624-word state, lazy twist)
//   tensor.exponential_(1)   = per element: r = random64() = (out32 << 32) | out32',
//                              u = (r & (2^53-1)) * 2^-53,  q = (float)(-1.0 * log1p(-u))   [double]
//   torch.multinomial(p, 1)  = argmax(p / q)
//
// log1p is glibc 2.35's __log1p (sysdeps/ieee754/dbl-64/s_log1p.c, the fdlibm algorithm with the
// R1..R4 polynomial split), restated operation for operation; compiled with -ffp-contract=off it
// returns the same double as the host libm for every input (checked on 2e8 inputs), so the float
// the reference divides by is reproduced exactly.
//
// The state lives in global memory (snn_rng_state); a workgroup stages it in LDS, walks the
// requested draws block by block (624 outputs per twist) and twists cooperatively.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/snnhip.h"

namespace snn {

__host__ __device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__host__ __device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// dst = next 624-word block after src.  All threads of the workgroup must call it; three
// dependent waves of work (i<227 | 227<=i<454 | 454<=i<624) separated by barriers.
__device__ __forceinline__ void mt_twist_block(const uint32_t *src, uint32_t *dst, int tid, int nthreads) {
    for (int i = tid; i < 227; i += nthreads) dst[i] = src[i + 397] ^ mt_mix(src[i], src[i + 1]);
    __syncthreads();
    for (int i = 227 + tid; i < 454; i += nthreads) dst[i] = dst[i - 227] ^ mt_mix(src[i], src[i + 1]);
    __syncthreads();
    for (int i = 454 + tid; i < 624; i += nthreads)
        dst[i] = dst[i - 227] ^ mt_mix(src[i], i == 623 ? dst[0] : src[i + 1]);
    __syncthreads();
}

// Same twist performed by ONE wave in lockstep, no workgroup barrier: three dependent stripes, lane l producing the
// elements base + l + 64 k of a stripe (consecutive lanes <-> consecutive words: every LDS access of the wave is
// bank-conflict free; the former 4-elements-per-lane layout put lanes 16 apart on the same bank).  LDS operations of a
// wave execute in program order, so a stripe's reads see the previous stripe's writes.  All 64 lanes must call it.
__device__ __forceinline__ void mt_twist_block_wave(const uint32_t *src, uint32_t *dst, int lane) {
    {   // i in [0, 227): dst[i] = src[i+397] ^ mix(src[i], src[i+1])
        uint32_t a[4], a1[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = min(lane + 64 * k, 226); a[k] = src[i]; a1[k] = src[i + 1]; b[k] = src[i + 397]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; if (i < 227) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
    {   // i in [227, 454): dst[i] = dst[i-227] ^ mix(src[i], src[i+1])
        uint32_t a[4], a1[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = min(227 + lane + 64 * k, 453); a[k] = src[i]; a1[k] = src[i + 1]; b[k] = dst[i - 227]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = 227 + lane + 64 * k; if (i < 454) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
    {   // i in [454, 624): dst[i] = dst[i-227] ^ mix(src[i], i == 623 ? dst[0] : src[i+1])   (element 623 pairs with NEW dst[0])
        uint32_t a[3], a1[3], b[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int i = min(454 + lane + 64 * k, 623); a[k] = src[i]; a1[k] = (i == 623) ? dst[0] : src[i + 1]; b[k] = dst[i - 227]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int i = 454 + lane + 64 * k; if (i < 624) dst[i] = b[k] ^ mt_mix(a[k], a1[k]); }
    }
}

__host__ __device__ __forceinline__ int32_t dbl_hi(double x) { return (int32_t)(__builtin_bit_cast(long long, x) >> 32); }
__host__ __device__ __forceinline__ double dbl_set_hi(double x, int32_t h) {
    const uint64_t u = ((uint64_t)__builtin_bit_cast(long long, x) & 0xffffffffull) | ((uint64_t)(uint32_t)h << 32);
    return __builtin_bit_cast(double, u);
}

// glibc 2.35 __log1p for -1 < x <= 0 ... the general algorithm, branches for NaN/inf dropped
// because the argument is always -u with u in [0, 1).
__host__ __device__ __forceinline__ double log1p_glibc(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    double hfsq, f = 0.0, c = 0.0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
    int32_t k = 1, hu = 0;
    const int32_t hx = dbl_hi(x), ax = hx & 0x7fffffff;
    if (hx < 0x3FDA827A) {
        if (ax >= 0x3ff00000) return x == -1.0 ? -__builtin_huge_val() : __builtin_nan("");
        if (ax < 0x3e200000) {
            if (ax < 0x3c900000) return x;
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int32_t)0xbfd2bec3) { k = 0; f = x; hu = 1; }
    }
    if (k != 0) {
        u = 1.0 + x;
        hu = dbl_hi(u);
        k = (hu >> 20) - 1023;
        c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
        c = c / u;
        hu &= 0x000fffff;
        if (hu < 0x6a09e) {
            u = dbl_set_hi(u, hu | 0x3ff00000);
        } else {
            k += 1;
            u = dbl_set_hi(u, hu | 0x3fe00000);
            hu = (0x00100000 - hu) >> 2;
        }
        f = u - 1.0;
    }
    hfsq = 0.5 * f * f;
    if (hu == 0) {
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c = c + (double)k * ln2_lo;
            return (double)k * ln2_hi + c;
        }
        R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return (double)k * ln2_hi - ((R - ((double)k * ln2_lo + c)) - f);
    }
    s = f / (2.0 + f);
    z = s * s;
    R1 = z * Lp1; z2 = z * z;
    R2 = Lp2 + z * Lp3; z4 = z2 * z2;
    R3 = Lp4 + z * Lp5; z6 = z4 * z2;
    R4 = Lp6 + z * Lp7;
    R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return (double)k * ln2_hi - ((hfsq - (s * (hfsq + R) + ((double)k * ln2_lo + c))) - f);
}

// Exp(1) draw from two consecutive tempered mt19937 outputs (hi first), as torch does.
__host__ __device__ __forceinline__ float exp1_from_words(uint32_t hi, uint32_t lo) {
    const uint64_t r = ((uint64_t)hi << 32) | lo;
    const double u = (double)(long long)(r & ((1ull << 53) - 1ull)) * 0x1.0p-53;
    return (float)(-1.0 * log1p_glibc(-u));
}

}  // namespace snn
