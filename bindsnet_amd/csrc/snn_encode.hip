// snn_encode.hip -- spike encoders on the MI355X (bindsnet/encoding/encodings.py).
//
//  * snn_encode_bernoulli: bit-for-bit what torch.bernoulli(max_prob * datum.repeat([time, 1])) draws from the HOST
//    generator (encodings.py:51-98).  ATen's CPU kernel consumes ONE 32-bit mt19937 output per element, in element
//    order: u = (r & 0xFFFFFF) * 2^-24, spike = u < p (bernoulli_distribution<float> over uniform_real_distribution).
//    mt19937 has no cheap jump-ahead, so one workgroup walks the stream: twist a 624-word block cooperatively, turn it
//    into 624 spikes, next block.  The advanced state is written back for the host to re-install.
//  * snn_encode_poisson: the construction of encodings.py:101-152 (inter-spike intervals ~ Poisson(1000 / (x dt)), zero
//    intervals bumped to one, cumulated into spike times), one thread per input element, from a counter-based
//    Philox-4x32-10 stream keyed by (seed, element).  Same distribution, NOT the reference's stream: ATen's Poisson
//    sampler consumes a data-dependent number of generator outputs per element, which cannot be parallelised exactly.
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

// ---- Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
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

// Poisson(lam) sample: multiplication method below 30, Hoermann's transformed rejection (PTRS) above.
__device__ float poisson_sample(Philox &g, float lam) {
    if (lam <= 0.f) return 0.f;
    if (lam < 30.f) {
        const float limit = expf(-lam);
        float prod = g.uniform();
        int k = 0;
        while (prod > limit) { prod *= g.uniform(); ++k; }
        return (float)k;
    }
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
    for (;;) {
        const float U = g.uniform() - 0.5f, V = g.uniform();
        const float us = 0.5f - fabsf(U);
        const float k = floorf((2.f * a / us + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return k;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (logf(V) + logf(invalpha) - logf(a / (us * us) + b) <= -lam + k * loglam - lgammaf(k + 1.f)) return k;
    }
}

__global__ __launch_bounds__(256) void k_encode_poisson(const float *__restrict__ datum, int n, int steps, float dt,
                                                        unsigned long long seed, uint8_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float x = datum[i];
    const float lam = x != 0.f ? 1.0f / x * (1000.0f / dt) : 0.f;
    Philox g; g.init(seed, (unsigned long long)i);
    // spike "times" are the running sums of the intervals; time index 0 is dropped (spikes[1:] in the reference)
    long long next = 0;
    auto advance = [&]() { float k = poisson_sample(g, lam); if (x != 0.f && k == 0.f) k = 1.f; next += (long long)k; };
    advance();
    for (int t = 1; t <= steps; ++t) {
        uint8_t s = 0;
        if (x != 0.f) {
            while (next < t) advance();
            if (next == t) s = 1;
        }
        out[(size_t)(t - 1) * n + i] = s;
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
    hipLaunchKernelGGL(k_encode_poisson, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, datum, n, steps, dt, seed, out);
    return snn_check_launch();
}
