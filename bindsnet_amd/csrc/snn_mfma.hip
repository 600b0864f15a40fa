// snn_mfma.hip -- Connection.compute (bindsnet/network/topology.py:332-346) as an f32-MFMA GEMM:
// out[B, N] (+)= s[B, Nin] @ W[Nin, N] (+ bias) on v_mfma_f32_16x16x4_f32.
//
// gfx950's f32-input MFMA is bit-for-bit a k-ordered fmaf chain (cdna_hip_programming.md, "FP32-input MFMA"), and with
// 0/1 spikes every product is exact, so one accumulator chain over k = 0 .. Nin-1 reproduces the canonical
// ascending-source sequential order of snn_prop_dense_f32 / orc_prop_dense exactly -- PROVIDED the chain is never split:
// no split-K, one accumulator per output tile.  That is also what bounds it: Nin / 4 dependent MFMAs of 40 cycles each
// per 16x16 output tile, however sparse the spikes are, whereas the event-driven kernel only touches the rows that spiked.
// tools/bench_dense_prop.py measures both over the input density (profiles/r06_dense_mfma_density_map.json: 8.7 us for
// 128 x 784 -> 1600 at any density, the event-driven kernel 8.1 us at 1 %, 32 us at 20 %); the
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
    __shared__ __attribute__((aligned(16))) uint8_t st[16][KC + 16];   // (+16: a row starts 4 banks after the one before, so the 16 rows' 16-byte reads
                                                                      //  cover the 64 banks once; and the look-ahead read of the last round stays inside)
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
    const bool colv = col < N;                                  // (a column beyond N reads column 0 and is never stored: output columns are independent)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nfull = Nin >> 4;                                 // a round = 16 sources = four MFMAs; Nin & 15 sources are left for a partial round
    // the weights of round rd, element u: W[(rd * 16 + 4u + kq) * N + col] = (uniform row block) + (per-lane offset that never changes)
    unsigned voff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) voff[u] = (unsigned)(4 * u + kq) * (unsigned)N + (unsigned)(colv ? col : 0);
    float wq[RD][4];
    // always a load of a valid address (a lane- or round-conditional one becomes a branch, and across a branch the compiler waits for EVERY
    // load in flight -- vmcnt(0) in front of each round, the memory latency back in the chain): a round beyond the last full one re-reads that one
    auto load_round = [&](int rd, float (&dst)[4]) __attribute__((always_inline)) {
        const float *Wr = W + (size_t)min(rd, nfull - 1) * 16 * N;
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[u] = Wr[voff[u]];
    };
    auto spikes = [&](const uint4 &q, float (&a)[4]) __attribute__((always_inline)) {    // bytes kq, 4 + kq, 8 + kq, 12 + kq of the round's 16
        a[0] = (float)((q.x >> (8 * kq)) & 0xFFu); a[1] = (float)((q.y >> (8 * kq)) & 0xFFu);
        a[2] = (float)((q.z >> (8 * kq)) & 0xFFu); a[3] = (float)((q.w >> (8 * kq)) & 0xFFu);
    };
    if (nfull > 0) {
#pragma unroll
        for (int j = 0; j < RD; ++j) load_round(j, wq[j]);
    }
    for (int k0 = 0; k0 < Nin; k0 += KC) {
        __syncthreads();
        {   // stage the spike bytes of this K-chunk: thread -> (sample row, 16-byte pieces); beyond B / Nin: zeros
            const int row = threadIdx.x >> 4, b = m0 + row;
            // (one 16-byte load per piece where the row allows it: byte loads were issued -- and waited for -- one at a time, which was most of
            //  the kernel's time)
            uint4 v[KC / 256];
#pragma unroll
            for (int q = 0; q < KC / 256; ++q) {
                const int kk = k0 + ((threadIdx.x & 15) + 16 * q) * 16;
                v[q] = make_uint4(0, 0, 0, 0);
                if (b < B && kk < Nin) {
                    const uint8_t *src = s + (size_t)b * Nin + kk;
                    if ((Nin & 15) == 0) v[q] = *(const uint4 *)src;
                    else {
                        uint32_t w[4] = {0, 0, 0, 0};
                        for (int u = 0; u < 16; ++u) if (kk + u < Nin) w[u >> 2] |= (uint32_t)src[u] << (8 * (u & 3));
                        v[q] = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < KC / 256; ++q) *(uint4 *)&st[row][((threadIdx.x & 15) + 16 * q) * 16] = v[q];
        }
        __syncthreads();
        const int r0 = k0 >> 4, r1 = min(nfull, (k0 + KC) >> 4);        // this chunk's FULL rounds
        const uint8_t *srow = &st[r][0];
        // whole groups of RD rounds, branch-free: round rd's MFMAs, round rd + 1's spike bytes (one 16-byte LDS read) and round rd + RD's weights
        int rb = r0;
        if (rb + RD <= r1) {
            // two rounds deep: round rd's MFMAs run on floats made a round ago, while round rd + 1's bytes (read a round ago) are converted and
            // round rd + 2's are read -- neither the LDS latency nor the conversions sit between two dependent MFMAs
            float an[4];
            spikes(*(const uint4 *)(srow + (rb - r0) * 16), an);
            uint4 qn = *(const uint4 *)(srow + (rb + 1 - r0) * 16);
            for (; rb + RD <= r1; rb += RD) {
#pragma unroll
                for (int j = 0; j < RD; ++j) {
                    const int rd = rb + j;
                    float a[4] = {an[0], an[1], an[2], an[3]};
                    spikes(qn, an);
                    qn = *(const uint4 *)(srow + min(rd + 2 - r0, KC / 16) * 16);                       // (past the chunk's last round: the pad)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wq[j][u], acc, 0, 0, 0);
                    load_round(rd + RD, wq[j]);
                    __builtin_amdgcn_sched_barrier(0);          // (pin that order: the wait in front of a round is then vmcnt(28), not vmcnt(0))
                }
            }
        }
        // the chunk's last full rounds (fewer than RD, and only in the LAST chunk -- a whole chunk is a whole number of groups): their weights are in
        // flight already, nothing is left to prefetch
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int rd = rb + j;
            if (rd < r1) {                                      // (uniform)
                float a[4];
                spikes(*(const uint4 *)(srow + (rd - r0) * 16), a);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], wq[j][u], acc, 0, 0, 0);
            }
        }
        if ((Nin & 15) && k0 + KC >= Nin) {                     // the partial round: loaded here, lane by lane (sources beyond Nin: exact zeros)
            float a[4];
            spikes(*(const uint4 *)(srow + (nfull - r0) * 16), a);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = nfull * 16 + 4 * u + kq;
                const float w = k < Nin ? W[(size_t)k * N + (colv ? col : 0)] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], w, acc, 0, 0, 0);
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
