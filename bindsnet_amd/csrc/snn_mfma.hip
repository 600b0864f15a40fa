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
constexpr int KC = 1024;                // source rows whose spike bytes are staged per pass (16 KB of LDS)
constexpr int RD = 8;                   // rounds (of four MFMAs) whose weight loads are in flight ahead of the chain

// block = 4 waves = one 16-sample tile x four adjacent 16-column tiles; a wave = one output tile = ONE accumulator chain over k.
// Round 6: the chain used to wait for its four weight loads every round (global-load latency x Nin / 16: 22 .. 32 us for 784 sources
// against 3.3 us of dependent MFMAs).  Now the loads of round r + RD are issued when round r's MFMAs are: 32 loads in flight per lane, the
// chain sees LDS (the staged spike bytes) and registers only.
__global__ __launch_bounds__(256) void k_prop_dense_mfma(const float *__restrict__ W, const float *__restrict__ bias,
                                                         const uint8_t *__restrict__ s, float *__restrict__ out, int B, int Nin,
                                                         int N, int accumulate) {
    __shared__ uint8_t st[16][KC + 4];                         // (+4: rows land in different LDS banks)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware block -> tile map (workgroup b runs on XCD b % 8, each with an L2 of its own): the ny sample tiles of one 64-column group sit
    // on ONE XCD, back to back, so the group's [Nin x 64] weight slice is fetched from HBM once and re-read from that L2 (the whole matrix
    // does not fit one L2: with the (column group, sample tile) grid every XCD streamed it once per sample tile)
    const int ny = (B + 15) >> 4, ncg = (N + 63) >> 6;
    const int xcd = blockIdx.x & 7, w_ = blockIdx.x >> 3;
    const int cg = xcd + 8 * (w_ / ny), ty = w_ - (w_ / ny) * ny;
    if (cg >= ncg) return;
    const int m0 = ty * 16, n0 = (cg * 4 + wave) * 16;
    const int r = lane & 15, kq = lane >> 4;                   // A: row r, k = kq;  B: k = kq, column r
    const int col = n0 + r;
    const bool colv = col < N;
    const float *Wc = W + (colv ? col : 0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nrounds = (Nin + 15) >> 4;                        // a round = 16 sources = four MFMAs
    float wq[RD][4];
    auto load_round = [&](int rd, float (&dst)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = rd * 16 + 4 * u + kq;
            dst[u] = (k < Nin && colv) ? Wc[(size_t)k * N] : 0.f;
        }
    };
#pragma unroll
    for (int j = 0; j < RD; ++j) load_round(j, wq[j]);
    for (int k0 = 0; k0 < Nin; k0 += KC) {
        __syncthreads();
        {   // stage the spike bytes of this K-chunk: thread -> (sample row, 16-byte pieces); beyond B / Nin: zeros
            const int row = threadIdx.x >> 4, b = m0 + row;
            for (int piece = threadIdx.x & 15; piece * 16 < KC; piece += 16) {
                const int kk = k0 + piece * 16;
#pragma unroll
                for (int u = 0; u < 16; ++u) st[row][piece * 16 + u] = (b < B && kk + u < Nin) ? s[(size_t)b * Nin + kk + u] : (uint8_t)0;
            }
        }
        __syncthreads();
        const int r0 = k0 >> 4, r1 = min(nrounds, (k0 + KC) >> 4);
        for (int rb = r0; rb < r1; rb += RD) {
#pragma unroll
            for (int j = 0; j < RD; ++j) {
                const int rd = rb + j;
                if (rd < r1) {                                  // (uniform)
                    float a[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u] = (float)st[r][(rd - r0) * 16 + 4 * u + kq];      // (0 beyond Nin: staged zeros)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wq[j][u], acc, 0, 0, 0);
                    load_round(rd + RD, wq[j]);
                }
            }
        }
    }
    // C/D layout of the 16x16 shapes: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int b = m0 + kq * 4 + reg;
        if (b < B && colv) {
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
    const int ny = (B + 15) / 16, ncg = (N + 63) / 64;
    hipLaunchKernelGGL(k_prop_dense_mfma, dim3((unsigned)(8 * ((ncg + 7) / 8) * ny)), dim3(256), 0, (hipStream_t)stream, W, bias, s, out, B,
                       Nin, N, accumulate);
    return snn_check_launch();
}
