// snn_mfma.hip -- Connection.compute (bindsnet/network/topology.py:332-346) as an f32-MFMA GEMM:
// out[B, N] (+)= s[B, Nin] @ W[Nin, N] (+ bias) on v_mfma_f32_16x16x4_f32.
//
// gfx950's f32-input MFMA is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md, "FP32-input MFMA"), and with
// 0/1 spikes every product is exact, so one accumulator chain over k = 0 .. Nin-1 reproduces the canonical
// ascending-source sequential order of snn_prop_dense_f32 / orc_prop_dense exactly -- PROVIDED the chain is never split:
// no split-K, one accumulator per output tile.  That is also what bounds it: Nin / 4 dependent MFMAs of 40 cycles each
// per 16x16 output tile, however sparse the spikes are, whereas the event-driven kernel only touches the ~1 % of rows
// that spiked.  tools/bench_dense_prop.py measures both (numbers: profiles/r02_dense_prop_mfma_vs_event.json); the
// fused plans keep the event-driven form.  The 16x16x4 shape is used because its dependent-issue latency per k
// (40 cycles / 4) is a third of the 32x32x2 shape's (64 cycles / 2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KC = 256;                 // source rows staged per pass

// block = 4 waves = one 16-sample tile x four adjacent 16-column tiles
__global__ __launch_bounds__(256) void k_prop_dense_mfma(const float *__restrict__ W, const float *__restrict__ bias,
                                                         const uint8_t *__restrict__ s, float *__restrict__ out, int B, int Nin,
                                                         int N, int accumulate) {
    __shared__ uint8_t st[16][KC + 4];                         // (+4: rows land in different LDS banks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 16, n0 = (blockIdx.x * 4 + wave) * 16;
    const int r = lane & 15, kq = lane >> 4;                   // A: row r, k = kq;  B: k = kq, column r
    const int col = n0 + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < Nin; k0 += KC) {
        __syncthreads();
        {   // stage the spike bytes of this K-chunk: thread -> (sample row, 16-byte piece); beyond B / Nin: zeros
            const int row = threadIdx.x >> 4, piece = threadIdx.x & 15, b = m0 + row, kk = k0 + piece * 16;
#pragma unroll
            for (int u = 0; u < 16; ++u) st[row][piece * 16 + u] = (b < B && kk + u < Nin) ? s[(size_t)b * Nin + kk + u] : (uint8_t)0;
        }
        __syncthreads();
        const int kend = Nin - k0 < KC ? Nin - k0 : KC;
        for (int kk = 0; kk < kend; kk += 16) {                 // four MFMAs per round: their weight loads are issued together
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kk + 4 * u + kq, krow = k0 + k;
                a[u] = (float)st[r][k < KC ? k : KC - 1];
                b[u] = (k < kend && col < N) ? W[(size_t)krow * N + col] : 0.f;
                if (k >= kend) a[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (kk + 4 * u < kend) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
        }
    }
    // C/D layout of the 16x16 shapes: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int b = m0 + kq * 4 + reg;
        if (b < B && col < N) {
            float v = acc[reg];
            if (bias) v = v + bias[col];
            const size_t o = (size_t)b * N + col;
            out[o] = (accumulate ? out[o] : 0.0f) + v;
        }
    }
}

}  // namespace

extern "C" int snn_prop_dense_mfma_f32(const float *W, const float *bias, const uint8_t *s, float *out, int B, int Nin, int N,
                                       int accumulate, snn_stream_t stream) {
    if (!W || !s || !out || B <= 0 || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    hipLaunchKernelGGL(k_prop_dense_mfma, dim3((N + 63) / 64, (B + 15) / 16), dim3(256), 0, (hipStream_t)stream, W, bias, s, out, B,
                       Nin, N, accumulate);
    return snn_check_launch();
}
