// snn_encode.hip -- spike encoders on the MI355X (bindsnet/encoding/encodings.py).
//
//  * snn_encode_bernoulli: bit-for-bit what torch.bernoulli(max_prob * datum.repeat([time, 1])) draws from the HOST
//    generator (encodings.py:51-98).  ATen's CPU kernel consumes ONE 32-bit mt19937 output per element, in element
//    order: u = (r & 0xFFFFFF) * 2^-24, spike = u < p (bernoulli_distribution<float> over uniform_real_distribution).
//    mt19937 has no cheap jump-ahead, so one workgroup walks the stream: twist a 624-word block cooperatively, turn it
//    into 624 spikes, next block.  The advanced state is written back for the host to re-install.
//  * snn_encode_poisson: Poisson spike trains from a counter-based stream, specified operation by operation below and restated in
//    oracle/snn_oracle.c (bit-exact on both sides); same distribution as the reference's, not its stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"
#include "snn_rng.hpp"

namespace {
using namespace snn;

constexpr int ENT = 1024;

__global__ __launch_bounds__(ENT) void k_encode_bernoulli(snn_rng_state *rng, const float *__restrict__ datum, int n,
                                                          long long total, float max_prob, uint8_t *__restrict__ out) {
    __shared__ uint32_t mt[2][624];
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += ENT) mt[0][k] = rng->mt[k];
    int pos = rng->pos, cur = 0;
    __syncthreads();
    long long e = 0;
    while (e < total) {
        if (pos >= 624) { mt_twist_block(mt[cur], mt[cur ^ 1], tid, ENT); cur ^= 1; pos = 0; }     // (ends with a barrier)
        const int avail = (int)((long long)(624 - pos) < total - e ? (long long)(624 - pos) : total - e);
        for (int k = tid; k < avail; k += ENT) {
            const uint32_t r = mt_temper(mt[cur][pos + k]);
            const float u = (float)(r & 0xFFFFFFu) * 5.9604644775390625e-08f;       // * 2^-24, exact
            const float p = max_prob * datum[(e + k) % n];                           // one f32 multiply, like the reference
            out[e + k] = (uint8_t)(u < p);
        }
        pos += avail; e += avail;
        __syncthreads();
    }
    for (int k = tid; k < 624; k += ENT) rng->mt[k] = mt[cur][k];
    if (tid == 0) rng->pos = pos;
}

// ---- snn_encode_poisson.  The construction of encodings.py:101-152 -- inter-spike intervals ~ Poisson(lambda), lambda = 1 / x * (1000 / dt) in
// f32 as the reference computes it, zero intervals bumped to one, cumulated into spike times -- one thread per input element, from a
// counter-based stream.  Same distribution, NOT the reference's stream (ATen's Poisson sampler consumes a data-dependent number of
// generator outputs per element: not parallelisable exactly), so the stream is SPECIFIED here, operation by operation, in terms every IEEE
// machine computes identically -- f32 / f64 add, multiply, divide, f32 square root, integer conversions; no libm call: exp, log and
// log-gamma are the fixed series below -- and oracle/snn_oracle.c (orc_encode_poisson) restates it in plain C: the two agree bit for bit
// (tests/test_gpu_encoding.py).
//   uniforms   Philox-4x32-10 (Salmon et al., SC'11), key = seed (lo, hi), counter = (block, 0, element lo, element hi), block = 0, 1, ...;
//              the four outputs of a block are used last first; u = ((r >> 8) + 1) * 2^-24 in (0, 1]
//   sampler    lambda < 30: multiplication method -- k = number of further uniforms until their running product (f64) drops to exp(-lambda);
//              lambda >= 30: Hoermann's transformed rejection (PTRS, 1993), all in f64 but sqrt(lambda), which is the f32 square root
struct Philox {
    uint32_t key[2], ctr[4], out[4];
    int have;
    __device__ void init(unsigned long long seed, unsigned long long stream) {
        key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0; ctr[1] = 0; ctr[2] = (uint32_t)stream; ctr[3] = (uint32_t)(stream >> 32);
        have = 0;
    }
    __device__ void round10() {
        uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
    __device__ float uniform() {                     // (0, 1]
        if (!have) { round10(); if (++ctr[0] == 0) ++ctr[1]; have = 4; }
        const uint32_t r = out[--have];
        return ((float)(r >> 8) + 1.0f) * 5.9604644775390625e-08f;
    }
};

// floor for |x| < 2^62 (exact: the conversion truncates)
__device__ __forceinline__ double pz_floor(double x) { double t = (double)(long long)x; if (t > x) t -= 1.0; return t; }
// x * 2^k through the exponent field (x normal, result normal)
__device__ __forceinline__ double pz_scale(double x, int k) { return __longlong_as_double(__double_as_longlong(x) + ((long long)k << 52)); }
// exp(y), y in [-40, 0]: y = k ln2 + r, |r| <= ln2 / 2; Taylor to r^13 (Horner, multiply then add), scaled by 2^k
__device__ double pz_exp(double y) {
    const double k = pz_floor(y * 1.4426950408889634 + 0.5);
    const double r = (y - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0; p = p * r + 1.0 / 39916800.0; p = p * r + 1.0 / 3628800.0; p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0; p = p * r + 1.0 / 5040.0; p = p * r + 1.0 / 720.0; p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0; p = p * r + 1.0 / 6.0; p = p * r + 0.5; p = p * r + 1.0; p = p * r + 1.0;
    return pz_scale(p, (int)k);
}
// log(x), x > 0 normal: x = m 2^e with m in [sqrt(1/2), sqrt(2)); s = (m - 1) / (m + 1); log m = 2 (s + s^3/3 + ... + s^23/23)
__device__ double pz_log(double x) {
    long long bits = __double_as_longlong(x);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);      // [1, 2)
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0; p = p * z + 1.0 / 7.0; p = p * z + 1.0 / 5.0; p = p * z + 1.0 / 3.0; p = p * z + 1.0;
    return (double)e * 6.93147180559945286227e-01 + 2.0 * s * p;
}
// log(k!) for an integer k >= 0: the product itself up to 9!, Stirling's series in z = k + 1 beyond
__device__ double pz_lfact(double k) {
    if (k < 10.0) {
        double f = 1.0;
        for (double i = 2.0; i <= k; i += 1.0) f = f * i;
        return pz_log(f);
    }
    const double z = k + 1.0, zi = 1.0 / z, z2 = zi * zi;
    double c = -1.0 / 1680.0;
    c = c * z2 + 1.0 / 1260.0; c = c * z2 - 1.0 / 360.0; c = c * z2 + 1.0 / 12.0;
    return ((z - 0.5) * pz_log(z) - z) + 0.91893853320467278056 + c * zi;
}

__device__ double poisson_sample(Philox &g, float lamf) {
    if (!(lamf > 0.f)) return 0.0;
    const double lam = (double)lamf;
    if (lamf < 30.f) {
        const double limit = pz_exp(-lam);
        double prod = (double)g.uniform();
        double k = 0.0;
        while (prod > limit) { prod = prod * (double)g.uniform(); k += 1.0; }
        return k;
    }
    const double slam = (double)sqrtf(lamf), loglam = pz_log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (;;) {
        const double U = (double)g.uniform() - 0.5, V = (double)g.uniform();
        const double us = 0.5 - (U < 0.0 ? -U : U);
        const double k = pz_floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (pz_log(V) + pz_log(invalpha) - pz_log(a / (us * us) + b) <= (k * loglam - lam) - pz_lfact(k)) return k;
    }
}

// One thread per input element walks ITS spike times -- sample an interval, advance, mark -- into a zeroed train; the time axis is not walked
// (round 6: a loop over the timesteps made every wave pay its slowest lane's sampler at nearly every step -- lanes spike at different times, and
// the two sampler branches diverge: 0.52 ms per MNIST-sized sample; this form: the trips of the element with the most spikes).
__global__ __launch_bounds__(64) void k_encode_poisson(const float *__restrict__ datum, int n, int steps, float dt,
                                                       unsigned long long seed, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float x = datum[i];
    if (x == 0.f) return;
    const float lam = 1.0f / x * (1000.0f / dt);
    Philox g; g.init(seed, (unsigned long long)i);
    // spike "times" are the running sums of the intervals; time index 0 is dropped (spikes[1:] in the reference)
    long long next = 0;
    for (;;) {
        double k = poisson_sample(g, lam);
        if (k == 0.0) k = 1.0;
        next += (long long)k;
        if (next > (long long)steps) break;
        out[(size_t)(next - 1) * n + i] = 1;
    }
}

}  // namespace

extern "C" int snn_encode_bernoulli(snn_rng_state *rng, const float *datum, int n, int steps, float max_prob, uint8_t *out,
                                    snn_stream_t stream) {
    if (!rng || !datum || !out || n <= 0 || steps <= 0) return SNN_ERR_INVALID;
    hipLaunchKernelGGL(k_encode_bernoulli, dim3(1), dim3(ENT), 0, (hipStream_t)stream, rng, datum, n, (long long)n * steps, max_prob, out);
    return snn_check_launch();
}

extern "C" int snn_encode_poisson(const float *datum, int n, int steps, float dt, unsigned long long seed, uint8_t *out,
                                  snn_stream_t stream) {
    if (!datum || !out || n <= 0 || steps <= 0 || !(dt > 0.f)) return SNN_ERR_INVALID;
    if (hipMemsetAsync(out, 0, (size_t)n * steps, (hipStream_t)stream) != hipSuccess) return snn_check_launch();
    hipLaunchKernelGGL(k_encode_poisson, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, datum, n, steps, dt, seed, out);
    return snn_check_launch();
}
