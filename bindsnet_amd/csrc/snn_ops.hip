// snn_ops.hip -- per-operator gfx950 kernels + their C-ABI entry points (include/snnhip.h).
// One entry point per reference function of SURVEY.md 8(a); the fused multi-step drivers live in
// snn_run.hip and reuse the device code here.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).  -ffp-contract=off
// is load-bearing: the reference rounds after every multiply and every add.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/snnhip.h"
#include "snn_conv_events.hpp"
#include "snn_order.hpp"
#include "snn_common.hpp"
#include "snn_rng.hpp"

using namespace snn;

// =============================================================================================
// a5 / a6: spike propagation, event-driven.
// grid (ceil(N/256), B), 256 threads: thread <-> target column j, block <-> sample b.
// Per 1024-source chunk each of the 4 waves compacts its 256 sources into an ascending list in
// LDS (wave ballot + popcount), then every thread walks the 4 lists in order and accumulates
// W[i,j] (coalesced 1 KiB row segments) through the ordered-sum state machine.
// =============================================================================================
template <class SUM>
__global__ __launch_bounds__(256) void k_prop(const float *__restrict__ W, const float *__restrict__ bias,
                                              const uint8_t *__restrict__ s, float *__restrict__ out,
                                              int B, int Nin, int N, int accumulate) {
    __shared__ uint32_t list[4][256];
    __shared__ int cnt[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.y, j = blockIdx.x * 256 + tid;
    const bool valid = j < N;
    const int jc = valid ? j : N - 1;
    const uint8_t *srow = s + (size_t)b * Nin;
    SUM acc;
    acc.init(j >= (N / 32) * 32);
    const uint64_t lt = (1ull << lane) - 1ull;

    for (int base = 0; base < Nin; base += 1024) {
        int n_w = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = base + wave * 256 + p * 64 + lane;
            const uint32_t sv = (i < Nin) ? srow[i] : 0u;
            const uint64_t m = __ballot(sv != 0);
            if (sv) list[wave][n_w + __popcll(m & lt)] = ((uint32_t)i << 8) | sv;
            n_w += __popcll(m);
        }
        if (lane == 0) cnt[wave] = n_w;
        __syncthreads();
        for (int w = 0; w < 4; ++w) {
            const int n = cnt[w];
            for (int k = 0; k < n; k += 4) {
                uint32_t e[4]; float wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    e[u] = list[w][(k + u < n) ? k + u : k];
                    wv[u] = W[(size_t)(e[u] >> 8) * N + jc];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (k + u < n) acc.add((int)(e[u] >> 8), wv[u] * (float)(e[u] & 255u), Nin);
            }
        }
        __syncthreads();
    }
    if (valid) {
        float r = acc.finish(Nin);
        if (bias) r = r + bias[j];
        const size_t o = (size_t)b * N + j;
        out[o] = (accumulate ? out[o] : 0.0f) + r;
    }
}

extern "C" int snn_prop_cascade_f32(const float *W, const uint8_t *s, float *out, int B, int Nin, int N,
                                    int accumulate, snn_stream_t stream) {
    if (!W || !s || !out || B <= 0 || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (Nin > kMaxTerms || B > 65535) return SNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_prop<OuterSum>, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, W,
                       (const float *)nullptr, s, out, B, Nin, N, accumulate);
    return snn_check_launch();
}

extern "C" int snn_prop_dense_f32(const float *W, const float *bias, const uint8_t *s, float *out, int B,
                                  int Nin, int N, int accumulate, snn_stream_t stream) {
    if (!W || !s || !out || B <= 0 || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (Nin >= (1 << 24) || B > 65535) return SNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_prop<SeqSum>, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, W, bias, s,
                       out, B, Nin, N, accumulate);
    return snn_check_launch();
}

// =============================================================================================
// a7: conv2d propagation.  thread <-> (b, oy, ox, chunk of 8 output channels) -- blockIdx.y = the chunk (round 6: with all channels of a
// pixel on one thread the conv_mnist.py shapes gave 36 workgroups and 58 us per call; 8 us like this); weights broadcast from LDS.
// Tap order (kh, kw, cin): taps row-major, input channels innermost (what oneDNN does for C_in <= 16), sequential, + bias.
// =============================================================================================
__global__ __launch_bounds__(256) void k_conv2d(const float *__restrict__ W, const float *__restrict__ bias,
                                                const uint8_t *__restrict__ s, float *__restrict__ out, int B,
                                                int Cin, int H, int Wd, int Cout, int KH, int KW, int stride,
                                                int pad, int OH, int OW, int accumulate, int nstage) {
    extern __shared__ float wsm[];  // [Cout][Cin*KH*KW] | staged input images of the block's samples (u8, when they fit: nstage > 0)
    const int taps = Cin * KH * KW;
    for (int k = threadIdx.x; k < Cout * taps; k += blockDim.x) wsm[k] = W[k];
    const long npix = (long)B * OH * OW;
    const long p0 = (long)blockIdx.x * blockDim.x;
    const int img = Cin * H * Wd, bfirst = (int)(p0 / ((long)OW * OH));
    uint8_t *simg = (uint8_t *)(wsm + Cout * taps);
    if (nstage > 0) {                                    // the images of samples bfirst .. bfirst + nstage - 1: every window of the block reads them
        const long n = (long)min(nstage, B - bfirst) * img;
        const uint8_t *src = s + (size_t)bfirst * img;
        for (long k = threadIdx.x; k < n; k += blockDim.x) simg[k] = src[k];
    }
    __syncthreads();
    const long p = p0 + threadIdx.x;
    if (p >= npix) return;
    const int ox = (int)(p % OW), oy = (int)((p / OW) % OH), b = (int)(p / ((long)OW * OH));
    const uint8_t *sb = nstage > 0 ? simg + (size_t)(b - bfirst) * img : s + (size_t)b * img;
    {
        const int c0 = (int)blockIdx.y * 8;
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        for (int ky = 0; ky < KH; ++ky)               // reference order: taps row-major, input channels innermost
            for (int kx = 0; kx < KW; ++kx)
                for (int ci = 0; ci < Cin; ++ci) {
                    const int tap = (ci * KH + ky) * KW + kx;
                    const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                    if (iy < 0 || iy >= H || ix < 0 || ix >= Wd) continue;
                    const uint8_t sv = sb[((size_t)ci * H + iy) * Wd + ix];
                    if (!sv) continue;
                    const float fs = (float)sv;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c0 + u < Cout) acc[u] += fs * wsm[(c0 + u) * taps + tap];
                }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c0 + u < Cout) {
                float r = acc[u];
                if (bias) r = r + bias[c0 + u];
                const size_t o = (((size_t)b * Cout + c0 + u) * OH + oy) * OW + ox;
                out[o] = (accumulate ? out[o] : 0.0f) + r;
            }
    }
}

extern "C" int snn_prop_conv2d_f32(const float *W, const float *bias, const uint8_t *s, float *out, int B,
                                   int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad,
                                   int accumulate, snn_stream_t stream) {
    if (!W || !s || !out || B <= 0 || Cin <= 0 || H <= 0 || Wd <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || pad < 0)
        return SNN_ERR_INVALID;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return SNN_ERR_INVALID;
    if (Cin > 16) return SNN_ERR_UNSUPPORTED;      // the reference's accumulation order is only characterised up to 16 channels
    size_t lds = sizeof(float) * (size_t)Cout * Cin * KH * KW;
    if (lds > 64 * 1024) return SNN_ERR_UNSUPPORTED;
    const long npix = (long)B * OH * OW;
    // a block of 256 consecutive output pixels touches at most 255 / (OH * OW) + 2 samples: their input images go to LDS when they fit 32 KB
    int nstage = 255 / (OH * OW) + 2;
    if ((size_t)nstage * Cin * H * Wd > 32 * 1024) nstage = 0; else lds += (size_t)nstage * Cin * H * Wd;
    hipLaunchKernelGGL(k_conv2d, dim3((unsigned)((npix + 255) / 256), (unsigned)((Cout + 7) / 8)), dim3(256), lds, (hipStream_t)stream, W,
                       bias, s, out, B, Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW, accumulate, nstage);
    return snn_check_launch();
}

// =============================================================================================
// a2: input layer: trace + raster.
// =============================================================================================
__global__ __launch_bounds__(256) void k_input(const uint8_t *__restrict__ s, float *__restrict__ x, long n,
                                               float trace_decay, float trace_scale, int additive,
                                               uint8_t *__restrict__ raster) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        const uint8_t sv = s[k];
        if (x) x[k] = trace_next(x[k], sv, trace_decay, trace_scale, additive);
        if (raster) raster[k] = sv;
    }
}

extern "C" int snn_input_step(const uint8_t *s, float *x, long n_total, float trace_decay, float trace_scale,
                              int additive, uint8_t *raster_out, snn_stream_t stream) {
    if (!s || n_total <= 0) return SNN_ERR_INVALID;
    if (!x && !raster_out) return SNN_OK;
    const unsigned grid = (unsigned)((n_total + 255) / 256 < 2048 ? (n_total + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_input, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, x, n_total, trace_decay,
                       trace_scale, additive, raster_out);
    return snn_check_launch();
}

// =============================================================================================
// a3: LIF step, fully elementwise (state streaming: v, refrac r/w, I r/w, s w, x r/w).
// =============================================================================================
template <bool VTH>
__global__ __launch_bounds__(256) void k_lif(float *__restrict__ v, float *__restrict__ refrac,
                                             uint8_t *__restrict__ s, float *__restrict__ x,
                                             float *__restrict__ I, long n, snn_lif_params p,
                                             uint8_t *__restrict__ raster_s, float *__restrict__ raster_v,
                                             const float *__restrict__ thresh_vec, int N) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        float vv = v[k], rc = refrac[k], cur = I[k];
        if (rc > 0.f) { cur = 0.f; I[k] = 0.f; }   // nodes.py:511 masks the caller's tensor in place
        if (VTH) p.thresh = thresh_vec[k % N];     // nodes.py:519 with a tensor-valued `thresh`: broadcast over the batch
        const uint8_t sp = lif_update(vv, rc, cur, p);
        v[k] = vv; refrac[k] = rc; s[k] = sp;
        if (p.traces) x[k] = trace_next(x[k], sp, p.trace_decay, p.trace_scale, p.traces_additive);
        if (raster_s) raster_s[k] = sp;
        if (raster_v) raster_v[k] = vv;
    }
}

extern "C" int snn_lif_step_vth(float *v, float *refrac, uint8_t *s, float *x, float *I, int B, int N,
                                const snn_lif_params *h_p, const float *thresh_vec, uint8_t *raster_s, float *raster_v,
                                snn_stream_t stream) {
    if (!v || !refrac || !s || !I || !h_p || B <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (h_p->traces && !x) return SNN_ERR_INVALID;
    const long n = (long)B * N;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (thresh_vec)
        hipLaunchKernelGGL(k_lif<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, refrac, s, x, I, n, *h_p,
                           raster_s, raster_v, thresh_vec, N);
    else
        hipLaunchKernelGGL(k_lif<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, v, refrac, s, x, I, n, *h_p,
                           raster_s, raster_v, thresh_vec, N);
    return snn_check_launch();
}

extern "C" int snn_lif_step(float *v, float *refrac, uint8_t *s, float *x, float *I, int B, int N,
                            const snn_lif_params *h_p, uint8_t *raster_s, float *raster_v, snn_stream_t stream) {
    return snn_lif_step_vth(v, refrac, s, x, I, B, N, h_p, nullptr, raster_s, raster_v, stream);
}

// =============================================================================================
// a4: Diehl&Cook step = two kernels.
//  k_dc_membrane: thread <-> neuron j, loops the batch (theta is shared by the batch and needs the
//                 per-neuron spike count over b) -> crossings written to s; snapshots the cursor.
//  k_dc_arbitrate: block <-> sample b: rank of b among rows with a crossing (noise offset),
//                 argmax of 1/q over the candidates (first maximal index), winner written back,
//                 trace + raster for the row.  The block of the last row publishes the new cursor.
// =============================================================================================
__global__ __launch_bounds__(256) void k_dc_membrane(float *__restrict__ v, float *__restrict__ refrac,
                                                     uint8_t *__restrict__ s, float *__restrict__ theta,
                                                     const float *__restrict__ I, int B, int N,
                                                     snn_dc_params p, long long *__restrict__ cursor,
                                                     float *__restrict__ raster_v) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0 && cursor) cursor[1] = cursor[0];   // snapshot read by k_dc_arbitrate
    if (j >= N) return;
    float th = theta[j];
    if (p.learning) th = th * p.theta_decay;            // nodes.py:1079
    const float thr = p.lif.thresh + th;                // nodes.py:1088
    int cnt = 0;
    for (int b = 0; b < B; ++b) {
        const size_t k = (size_t)b * N + j;
        float vv = v[k], rc = refrac[k];
        const uint8_t sp = dc_update(vv, rc, I[k], thr, p.lif);
        v[k] = vv; refrac[k] = rc; s[k] = sp;
        cnt += sp;
        if (raster_v) raster_v[k] = vv;
    }
    if (p.learning) th = th + p.theta_plus * (float)cnt;   // nodes.py:1094 (count is exact in f32)
    theta[j] = th;
}

__global__ __launch_bounds__(256) void k_dc_arbitrate(uint8_t *__restrict__ s, float *__restrict__ x, int B,
                                                      int N, snn_dc_params p, const float *__restrict__ Q,
                                                      long long q_len, long long *__restrict__ cursor,
                                                      int *__restrict__ status, uint8_t *__restrict__ raster_s) {
    __shared__ int any_row[1024];
    __shared__ float red_v[4];
    __shared__ int red_j[4];
    __shared__ int s_win;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x;
    int win = -2;  // -2: keep crossings as they are (one_spike off / no crossing / noise exhausted)
    if (p.one_spike) {
        // which rows 0..b have a crossing: one wave per row, strided
        for (int r = wave; r <= b; r += 4) {
            bool a = false;
            for (int j = lane; j < N; j += 64) a |= s[(size_t)r * N + j] != 0;
            const uint64_t m = __ballot(a);
            if (lane == 0) any_row[r] = m != 0;
        }
        __syncthreads();
        int rank = 0;
        for (int r = 0; r < b; ++r) rank += any_row[r];
        const int mine = any_row[b];
        const long long cur = cursor[1];
        const long long off = cur + (long long)rank * N;
        if (b == B - 1) {
            const long long nxt = cur + (long long)(rank + mine) * N;
            if (tid == 0) { cursor[0] = nxt <= q_len ? nxt : cur; if (nxt > q_len) atomicExch(status, SNN_ERR_NOISE); }
        }
        if (mine && off + N <= q_len) {
            float bv = -1.f; int bj = 0x7fffffff;
            for (int j = tid; j < N; j += 256)
                if (s[(size_t)b * N + j]) {
                    const float val = 1.0f / Q[off + j];        // p / q with p = 1 (nodes.py:1100-1102)
                    if (val > bv) { bv = val; bj = j; }         // ascending j: strict > keeps the first max
                }
            // wave reduce: max value, ties -> lowest index
            for (int d = 32; d >= 1; d >>= 1) {
                const float ov = __shfl_xor(bv, d); const int oj = __shfl_xor(bj, d);
                if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
            }
            if (lane == 0) { red_v[wave] = bv; red_j[wave] = bj; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < 4; ++w)
                    if (red_v[w] > bv || (red_v[w] == bv && red_j[w] < bj)) { bv = red_v[w]; bj = red_j[w]; }
                s_win = bj;
            }
            __syncthreads();
            win = s_win;
        } else if (mine) {
            if (tid == 0) atomicExch(status, SNN_ERR_NOISE);
        }
    }
    for (int j = tid; j < N; j += 256) {
        const size_t k = (size_t)b * N + j;
        uint8_t sp = s[k];
        if (win >= 0) { sp = (j == win); s[k] = sp; }
        if (p.lif.traces) x[k] = trace_next(x[k], sp, p.lif.trace_decay, p.lif.trace_scale, p.lif.traces_additive);
        if (raster_s) raster_s[k] = sp;
    }
}

int snn_launch_dc_membrane(float *v, float *refrac, uint8_t *s, float *theta, const float *I, int B, int N,
                           const snn_dc_params &p, long long *cursor, float *raster_v, hipStream_t st) {
    hipLaunchKernelGGL(k_dc_membrane, dim3((N + 255) / 256), dim3(256), 0, st, v, refrac, s, theta, I, B, N, p, cursor,
                       raster_v);
    return snn_check_launch();
}

int snn_launch_dc_arbitrate(uint8_t *s, float *x, int B, int N, const snn_dc_params &p, const float *Q, long long q_len,
                            long long *cursor, int *status, uint8_t *raster_s, hipStream_t st) {
    hipLaunchKernelGGL(k_dc_arbitrate, dim3(B), dim3(256), 0, st, s, x, B, N, p, Q, q_len, cursor, status, raster_s);
    return snn_check_launch();
}

extern "C" int snn_dc_step(float *v, float *refrac, uint8_t *s, float *x, float *theta, const float *I, int B,
                           int N, const snn_dc_params *h_p, const float *noise_q, long long q_len,
                           long long *cursor, int *status, uint8_t *raster_s, float *raster_v,
                           snn_stream_t stream) {
    if (!v || !refrac || !s || !theta || !I || !h_p || B <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (h_p->lif.traces && !x) return SNN_ERR_INVALID;
    if (h_p->one_spike && (!noise_q || !cursor || !status)) return SNN_ERR_INVALID;
    if (B > 1024) return SNN_ERR_UNSUPPORTED;
    int rc = snn_launch_dc_membrane(v, refrac, s, theta, I, B, N, *h_p, cursor, raster_v, (hipStream_t)stream);
    if (rc) return rc;
    return snn_launch_dc_arbitrate(s, x, B, N, *h_p, noise_q, q_len, cursor, status, raster_s, (hipStream_t)stream);
}

extern "C" int snn_dc_arbitrate(uint8_t *s, float *x, int B, int N, const snn_dc_params *h_p, const float *noise_q,
                                long long q_len, long long *cursor, int *status, uint8_t *raster_s,
                                snn_stream_t stream) {
    if (!s || !h_p || B <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (h_p->lif.traces && !x) return SNN_ERR_INVALID;
    if (h_p->one_spike && (!noise_q || !cursor || !status)) return SNN_ERR_INVALID;
    if (B > 1024) return SNN_ERR_UNSUPPORTED;
    return snn_launch_dc_arbitrate(s, x, B, N, *h_p, noise_q, q_len, cursor, status, raster_s, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Device-side noise for one_spike: materialise, for the rows that crossed threshold this step,
// exactly the Exp(1) draws torch.multinomial would consume (row-major over [rows_with_crossing, N]),
// but only at the candidate positions (the others are never read by the arbitration).
// One workgroup: the mt19937 state is staged in LDS and advanced block by block.
// qbuf[rank*N + j] receives the draw of candidate (b, j); cursor[1] is zeroed so that
// k_dc_arbitrate indexes qbuf from 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rng_fill(snn_rng_state *__restrict__ rng, const uint8_t *__restrict__ s,
                                                   int B, int N, float *__restrict__ qbuf,
                                                   long long *__restrict__ cursor) {
    __shared__ uint32_t mt[2][624];
    __shared__ int rank[1025];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 624; i += 1024) mt[0][i] = rng->mt[i];
    for (int r = wave; r < B; r += 16) {
        bool a = false;
        for (int j = lane; j < N; j += 64) a |= s[(size_t)r * N + j] != 0;
        const uint64_t m = __ballot(a);
        if (lane == 0) rank[r + 1] = m != 0;
    }
    __syncthreads();
    if (tid == 0) {
        rank[0] = 0;
        for (int r = 0; r < B; ++r) rank[r + 1] += rank[r];
        cursor[1] = 0;
    }
    __syncthreads();
    const int rows = rank[B];
    if (rows == 0) return;
    const long long pos = rng->pos;
    const long long E = pos + 2ll * rows * N;          // one past the last 32-bit output consumed
    const int ntw = (int)((E - 1) / 624);              // twists the lazy generator performs
    const long long total = (long long)B * N;
    for (int m = 0; m <= ntw; ++m) {
        const uint32_t *cur = mt[m & 1];
        for (long long idx = tid; idx < total; idx += 1024) {
            if (!s[idx]) continue;
            const int b = (int)(idx / N), j = (int)(idx - (long long)b * N);
            if (rank[b + 1] == rank[b]) continue;
            const long long d = (long long)rank[b] * N + j;
            const long long w0 = pos + 2 * d, w1 = w0 + 1;
            const int m0 = (int)(w0 / 624), m1 = (int)(w1 / 624);
            if (m0 == m && m1 == m) {
                qbuf[d] = exp1_from_words(mt_temper(cur[w0 - 624ll * m]), mt_temper(cur[w1 - 624ll * m]));
            } else if (m0 == m) {                        // pair straddles a twist: park the high word
                qbuf[d] = __uint_as_float(mt_temper(cur[w0 - 624ll * m]));
            } else if (m1 == m) {
                qbuf[d] = exp1_from_words(__float_as_uint(qbuf[d]), mt_temper(cur[w1 - 624ll * m]));
            }
        }
        __syncthreads();
        if (m < ntw) mt_twist_block(mt[m & 1], mt[(m + 1) & 1], tid, 1024);
    }
    for (int i = tid; i < 624; i += 1024) rng->mt[i] = mt[ntw & 1][i];
    if (tid == 0) {
        rng->pos = (int)(E - 624ll * ntw);
        rng->consumed += (long long)rows * N;
    }
}

int snn_launch_rng_fill(snn_rng_state *rng, const uint8_t *s, int B, int N, float *qbuf, long long *cursor,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_rng_fill, dim3(1), dim3(1024), 0, st, rng, s, B, N, qbuf, cursor);
    return snn_check_launch();
}

extern "C" int snn_rng_fill_exponential(snn_rng_state *rng, const uint8_t *crossings, int B, int N, float *qbuf,
                                        long long *cursor, snn_stream_t stream) {
    if (!rng || !crossings || !qbuf || !cursor || B <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (B > 1024) return SNN_ERR_UNSUPPORTED;
    return snn_launch_rng_fill(rng, crossings, B, N, qbuf, cursor, (hipStream_t)stream);
}

// =============================================================================================
// a8 / a9 / a10: outer-product plasticity.  grid (ceil(N/TJ), ceil(Nin/16)), 256 threads.
// A block owns a 16 x TJ tile of W.  The batch-side factors of the tile's rows and columns are
// staged in LDS once; per element the batch reduction walks only the samples in which the row's
// source or the column's target spiked (bit masks over b), through the ATen-ordered accumulator.
// =============================================================================================
struct StdpArgs {
    float *W;
    const uint8_t *s_src; const float *x_src; const uint8_t *s_tgt; const float *x_tgt;
    int B, Nin, N;
    float nu0, nu1; int use_dt; float dt, decay;
    int has_min; float wmin; int has_max; float wmax;
    int assume_clamped;
    // MSTDP (mode 1): x_src = p_plus, x_tgt = p_minus, s_* = previous-step spikes
    // mode 2: Hebbian (learning.py:1110-1135), mode 3: WeightDependentPostPre (learning.py:626-653): raw outer products
    // reduced over the batch first, learning rates (and the weight-dependent factors) applied to the sums
    int mode; float reward; const float *reward_vec;
};

constexpr int kTI = 16;

template <int TJ>
__global__ __launch_bounds__(256) void k_plasticity(StdpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = a.B, Nin = a.Nin, N = a.N;
    const int MW = (B + 31) / 32;                       // mask words per row / column
    float *xt = (float *)smem;                          // [B][TJ]  target-side float factor
    float *xs = xt + (size_t)B * TJ;                    // [B][kTI] source-side float factor
    uint32_t *mT = (uint32_t *)(xs + (size_t)B * kTI);  // [TJ][MW] target spike masks
    uint32_t *mS = mT + (size_t)TJ * MW;                // [kTI][MW]
    uint8_t *vT = (uint8_t *)(mS + (size_t)kTI * MW);   // [B][TJ] target spike values
    uint8_t *vS = vT + (size_t)B * TJ;                  // [B][kTI]
    __shared__ int any_flag;

    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * TJ, i0 = blockIdx.y * kTI;
    if (tid == 0) any_flag = 0;
    __syncthreads();
    int local_any = 0;
    for (int k = tid; k < B * TJ; k += 256) {
        const int b = k / TJ, jj = k - b * TJ, j = j0 + jj;
        float f = 0.f; uint8_t sv = 0;
        if (j < N) {
            sv = a.s_tgt[(size_t)b * N + j];
            f = a.x_tgt[(size_t)b * N + j];
            if (a.mode == 0) f = f * a.nu0;             // target_x * nu[0]  (MCC_learning.py:235)
        }
        xt[k] = f; vT[k] = sv; local_any |= sv;
    }
    for (int k = tid; k < B * kTI; k += 256) {
        const int b = k / kTI, ii = k - b * kTI, i = i0 + ii;
        float f = 0.f; uint8_t sv = 0;
        if (i < Nin) { sv = a.s_src[(size_t)b * Nin + i]; f = a.x_src[(size_t)b * Nin + i]; }
        xs[k] = f; vS[k] = sv; local_any |= sv;
    }
    if (local_any) any_flag = 1;
    __syncthreads();
    if (a.assume_clamped && !any_flag) return;          // nothing in this tile can change
    for (int k = tid; k < (TJ + kTI) * MW; k += 256) {
        const bool tgt = k < TJ * MW;
        const int kk = tgt ? k : k - TJ * MW;
        const int col = kk / MW, w = kk - col * MW;
        uint32_t m = 0;
        for (int bit = 0; bit < 32; ++bit) {
            const int b = w * 32 + bit;
            if (b < B && (tgt ? vT[(size_t)b * TJ + col] : vS[(size_t)b * kTI + col])) m |= 1u << bit;
        }
        (tgt ? mT : mS)[kk] = m;
    }
    __syncthreads();

    const long E = (long)Nin * N, Emain = (E / 32) * 32;
    constexpr int RG = 256 / TJ;                        // row groups handled concurrently
    const int jj = tid % TJ, rg = tid / TJ, j = j0 + jj;
    if (j >= N) return;
    for (int ii = rg; ii < kTI; ii += RG) {
        const int i = i0 + ii;
        if (i >= Nin) break;
        uint32_t anyS = 0, anyT = 0;
        for (int w = 0; w < MW; ++w) { anyS |= mS[ii * MW + w]; anyT |= mT[jj * MW + w]; }
        if (a.assume_clamped && !anyS && !anyT) continue;
        const long e = (long)i * N + j;
        const bool tail = e >= Emain;
        float w_ = a.W[e];
        if (a.mode == 0) {
            if (a.nu0 != 0.f) {                         // pre: sum_b s_src[b,i] * (x_tgt[b,j]*nu0)
                OuterSum acc; acc.init(tail);
                for (int w = 0; w < MW; ++w) {
                    uint32_t m = mS[ii * MW + w];
                    while (m) {
                        const int b = w * 32 + __ffs(m) - 1; m &= m - 1;
                        acc.add(b, (float)vS[(size_t)b * kTI + ii] * xt[(size_t)b * TJ + jj], B);
                    }
                }
                float u = acc.finish(B);
                if (a.use_dt) u = u * a.dt;
                w_ = w_ - u;
            }
            if (a.nu1 != 0.f) {                         // post: sum_b x_src[b,i] * (s_tgt[b,j]*nu1)
                OuterSum acc; acc.init(tail);
                for (int w = 0; w < MW; ++w) {
                    uint32_t m = mT[jj * MW + w];
                    while (m) {
                        const int b = w * 32 + __ffs(m) - 1; m &= m - 1;
                        const float st = (float)vT[(size_t)b * TJ + jj] * a.nu1;
                        acc.add(b, xs[(size_t)b * kTI + ii] * st, B);
                    }
                }
                float u = acc.finish(B);
                if (a.use_dt) u = u * a.dt;
                w_ = w_ + u;
            }
        } else if (a.mode == 2 || a.mode == 3) {
            float u1 = 0.f, u2 = 0.f;
            {   // U1 = sum_b s_src[b,i] * x_tgt[b,j]
                OuterSum acc; acc.init(tail);
                for (int w = 0; w < MW; ++w) {
                    uint32_t m = mS[ii * MW + w];
                    while (m) { const int b = w * 32 + __ffs(m) - 1; m &= m - 1; acc.add(b, (float)vS[(size_t)b * kTI + ii] * xt[(size_t)b * TJ + jj], B); }
                }
                u1 = acc.finish(B);
            }
            {   // U2 = sum_b x_src[b,i] * s_tgt[b,j]
                OuterSum acc; acc.init(tail);
                for (int w = 0; w < MW; ++w) {
                    uint32_t m = mT[jj * MW + w];
                    while (m) { const int b = w * 32 + __ffs(m) - 1; m &= m - 1; acc.add(b, xs[(size_t)b * kTI + ii] * (float)vT[(size_t)b * TJ + jj], B); }
                }
                u2 = acc.finish(B);
            }
            if (a.mode == 2) {                          // w += nu0 * U1; w += nu1 * U2
                w_ = w_ + a.nu0 * u1;
                w_ = w_ + a.nu1 * u2;
            } else {                                    // update = 0 - (nu0 U1)(w - wmin) + (nu1 U2)(wmax - w); w += update
                float upd = 0.f; bool have = false;
                if (a.nu0 != 0.f) { upd = 0.0f - (a.nu0 * u1) * (w_ - a.wmin); have = true; }
                if (a.nu1 != 0.f) { const float y = (a.nu1 * u2) * (a.wmax - w_); upd = have ? upd + y : y; have = true; }
                if (have) w_ = w_ + upd;
            }
        } else {                                        // MSTDP: sum_b reward * elig[b][i,j]
            OuterSum acc; acc.init(tail);
            for (int w = 0; w < MW; ++w) {
                uint32_t m = mS[ii * MW + w] | mT[jj * MW + w];
                while (m) {
                    const int b = w * 32 + __ffs(m) - 1; m &= m - 1;
                    const float e1 = xs[(size_t)b * kTI + ii] * (float)vT[(size_t)b * TJ + jj];   // p_plus (x) s_tgt
                    const float e2 = (float)vS[(size_t)b * kTI + ii] * xt[(size_t)b * TJ + jj];   // s_src (x) p_minus
                    const float el = e1 + e2;
                    const float r = a.reward_vec ? a.reward_vec[b] : a.reward;
                    acc.add(b, r * el, B);
                }
            }
            const float u = acc.finish(B);
            w_ = w_ + a.nu0 * u;                        // learning.py:1561
        }
        w_ = w_ * a.decay;
        if (a.has_min && w_ < a.wmin) w_ = a.wmin;
        if (a.has_max && w_ > a.wmax) w_ = a.wmax;
        a.W[e] = w_;
    }
}

static int launch_plasticity(const StdpArgs &a, hipStream_t st) {
    const int B = a.B, MW = (B + 31) / 32;
    auto lds = [&](int TJ) {
        return (size_t)B * TJ * 4 + (size_t)B * kTI * 4 + (size_t)(TJ + kTI) * MW * 4 + (size_t)B * TJ + (size_t)B * kTI;
    };
    const dim3 blk(256);
    const unsigned gy = (a.Nin + kTI - 1) / kTI;
    constexpr size_t kSmall = 64 * 1024, kBig = 144 * 1024;   // 160 KiB LDS per CU on gfx950
    static bool attr_set = false;
    if (!attr_set) {   // allow > 64 KiB dynamic LDS for the narrow-tile instantiation
        if (snn_check(hipFuncSetAttribute((const void *)k_plasticity<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)kBig)))
            return SNN_ERR_LAUNCH;
        attr_set = true;
    }
    if (lds(256) <= kSmall) {
        hipLaunchKernelGGL(k_plasticity<256>, dim3((a.N + 255) / 256, gy), blk, lds(256), st, a);
    } else if (lds(128) <= kSmall) {
        hipLaunchKernelGGL(k_plasticity<128>, dim3((a.N + 127) / 128, gy), blk, lds(128), st, a);
    } else if (lds(64) <= kSmall) {
        hipLaunchKernelGGL(k_plasticity<64>, dim3((a.N + 63) / 64, gy), blk, lds(64), st, a);
    } else if (lds(32) <= kBig) {
        hipLaunchKernelGGL(k_plasticity<32>, dim3((a.N + 31) / 32, gy), blk, lds(32), st, a);
    } else {
        return SNN_ERR_UNSUPPORTED;
    }
    return snn_check_launch();
}

extern "C" int snn_stdp_postpre(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt,
                                const float *x_tgt, int B, int Nin, int N, float nu0, float nu1, int use_dt,
                                float dt, float decay, int has_min, float wmin, int has_max, float wmax,
                                int assume_clamped, snn_stream_t stream) {
    if (!W || !s_src || !x_src || !s_tgt || !x_tgt || B <= 0 || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (B > 256) return SNN_ERR_UNSUPPORTED;
    StdpArgs a{W, s_src, x_src, s_tgt, x_tgt, B, Nin, N, nu0, nu1, use_dt, dt, decay, has_min, wmin, has_max,
               wmax, (assume_clamped && decay == 1.0f) ? 1 : 0, 0, 0.f, nullptr};
    return launch_plasticity(a, (hipStream_t)stream);
}

__global__ __launch_bounds__(256) void k_mstdp_traces(float *__restrict__ p_plus, float *__restrict__ p_minus,
                                                      uint8_t *__restrict__ s_src_prev,
                                                      uint8_t *__restrict__ s_tgt_prev,
                                                      const uint8_t *__restrict__ s_src,
                                                      const uint8_t *__restrict__ s_tgt, long n_src, long n_tgt,
                                                      float a_plus, float a_minus, float d_plus, float d_minus) {
    const long n = n_src + n_tgt;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        if (k < n_src) {
            const uint8_t sv = s_src[k];
            const float p = p_plus[k] * d_plus;                  // learning.py:1564
            p_plus[k] = p + a_plus * (float)sv;                  // :1565
            s_src_prev[k] = sv;
        } else {
            const long q = k - n_src;
            const uint8_t sv = s_tgt[q];
            const float p = p_minus[q] * d_minus;                // :1566
            p_minus[q] = p + a_minus * (float)sv;                // :1567
            s_tgt_prev[q] = sv;
        }
    }
}

// f4: PostPre on a Conv2dConnection (learning.py:457-497).  Phase 1: one thread per (sample, weight element) sums its
// products over the output positions l in ascending order; phase 2: one thread per weight element reduces the samples in
// ATen's sum(dim=0) order, applies the learning rates, decay and clamp.
__global__ __launch_bounds__(256) void k_conv_pp_partial(const uint8_t *__restrict__ s_src, const float *__restrict__ x_src,
                                                         const uint8_t *__restrict__ s_tgt, const float *__restrict__ x_tgt,
                                                         float *__restrict__ part, int B, int Cin, int H, int Wd, int Cout, int KH,
                                                         int KW, int stride, int pad, int OH, int OW) {
    const long K = (long)Cin * KH * KW, E = (long)Cout * K;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)B * E) return;
    const int b = (int)(id / E);
    const long e = id - (long)b * E;
    const int co = (int)(e / K), k = (int)(e - (long)co * K);
    const int ci = k / (KH * KW), ky = (k / KW) % KH, kx = k % KW;
    const int L = OH * OW;
    float a = 0.f, p = 0.f;
    for (int l = 0; l < L; ++l) {
        const int oy = l / OW, ox = l - oy * OW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
        const size_t si = (((size_t)b * Cin + ci) * H + (in ? iy : 0)) * Wd + (in ? ix : 0);
        const size_t ti = ((size_t)b * Cout + co) * L + l;
        a += x_tgt[ti] * (in ? (float)s_src[si] : 0.0f);
        p += (float)s_tgt[ti] * (in ? x_src[si] : 0.0f);
    }
    part[id] = a;
    part[(size_t)B * E + id] = p;
}

// The same partial sums from packed spike rows (snn_conv_events.hpp): a workgroup = one (sample, input channel, slice of 256
// weight elements); it packs the sample's source rows of that channel and ALL its target rows into LDS words (one per image
// row), then every thread walks the set bits its element reads.  Bit-identical to k_conv_pp_partial (checked body against body
// on the host: tests/test_conv_events_host.py); rows wider than 32 pixels never get here, multi-valued spike bytes take the
// dense body.  The default since round 4 (first run on an MI355X: parity tests green, 8 879 vs 4 361 timesteps/s on the
// Conv2d 5x5x32 PostPre graph at B = 16, profiles/r04_conv_postpre_dense_vs_events.jsonl); SNN_CONV_PP_EVENTS=0 selects the dense body.
__global__ __launch_bounds__(256) void k_conv_pp_partial_ev(const uint8_t *__restrict__ s_src, const float *__restrict__ x_src,
                                                            const uint8_t *__restrict__ s_tgt, const float *__restrict__ x_tgt,
                                                            float *__restrict__ part, int B, snn::ConvGeom g) {
    // Round 6: a workgroup = (sample, input channel, chunk of CO output channels) and stages what its elements read -- the chunk's target
    // traces, the channel's source traces, the packed spike rows -- in LDS: the walk over the events used to chase one global load per
    // event (40 us per call at the conv_mnist.py shapes; every workgroup also packed ALL target rows of its sample).
    constexpr int CO = 8;
    extern __shared__ uint32_t ev_lds[];                 // srow[H] | trow[CO * OH] | multi | xt[CO * OH * OW] | xs[H * Wd]
    const int L = g.OH * g.OW;
    uint32_t *srow = ev_lds, *trow = ev_lds + g.H;
    int *multi = (int *)(trow + CO * g.OH);
    float *xt = (float *)(multi + 1), *xs = xt + CO * L;
    const int tid = threadIdx.x, b = blockIdx.x / g.Cin, ci = blockIdx.x - b * g.Cin;
    const int co0 = blockIdx.y * CO, nco = min(CO, g.Cout - co0);
    if (tid == 0) *multi = 0;
    __syncthreads();
    const size_t soff = ((size_t)b * g.Cin + ci) * g.H * g.Wd, toff0 = ((size_t)b * g.Cout + co0) * L;
    int mine = 0;
    // (conv_pack_row's result from 4-byte loads where the row starts on a 4-byte boundary: a row of 24 / 28 bytes is 6 / 7 loads, not 24 / 28)
    auto pack = [&](const uint8_t *row, int n) -> uint32_t {
        if ((((uintptr_t)row) & 3) != 0) return snn::conv_pack_row(row, n, &mine);
        uint32_t m = 0;
        int x = 0;
        for (; x + 4 <= n; x += 4) {
            const uint32_t v = *(const uint32_t *)(row + x);
            if (v & 0xFEFEFEFEu) mine = 1;
            const uint32_t nz = (v | ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;      // byte k non-zero -> bit 8k+7
            m |= (((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xFu) << x;
        }
        for (; x < n; ++x) { const uint8_t v = row[x]; m |= (uint32_t)(v != 0) << x; if (v > 1) mine = 1; }
        return m;
    };
    for (int r = tid; r < g.H; r += 256) srow[r] = pack(s_src + soff + (size_t)r * g.Wd, g.Wd);
    for (int r = tid; r < nco * g.OH; r += 256) trow[r] = pack(s_tgt + toff0 + (size_t)r * g.OW, g.OW);
    for (int k = tid; k < nco * L; k += 256) xt[k] = x_tgt[toff0 + k];
    for (int k = tid; k < g.H * g.Wd; k += 256) xs[k] = x_src[soff + k];
    if (mine) atomicOr(multi, 1);
    __syncthreads();
    const bool dense = *multi != 0;
    const int KK = g.KH * g.KW;
    const long K = (long)g.Cin * KK, E = (long)g.Cout * K;
    if (tid >= nco * KK) return;
    const int cl = tid / KK, kk = tid - cl * KK, ky = kk / g.KW, kx = kk - ky * g.KW, co = co0 + cl;
    float a, p;
    if (!dense) snn::conv_pp_events(g, ky, kx, srow, trow + cl * g.OH, xs, xt + cl * L, &a, &p);
    else snn::conv_pp_dense(g, ky, kx, s_src + soff, xs, s_tgt + toff0 + (size_t)cl * L, xt + cl * L, &a, &p);
    const long id = (long)b * E + (long)co * K + (long)ci * KK + kk;
    part[id] = a;
    part[(size_t)B * E + id] = p;
}

__global__ __launch_bounds__(256) void k_conv_pp_apply(float *__restrict__ W, const float *__restrict__ part, int B, long E, float nu0,
                                                       float nu1, float decay, int has_min, float wmin, int has_max, float wmax) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const bool tail = e >= (E / 32) * 32;
    float w = W[e];
    // (the loads of sixteen samples' partial sums are issued together, then added in ATen's order: one round trip per sixteen terms)
    auto ordered = [&](const float *base) {
        OuterSum acc; acc.init(tail);
        for (int b0 = 0; b0 < B; b0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = base[(size_t)min(b0 + u, B - 1) * E + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (b0 + u < B) acc.add(b0 + u, v[u], B);
        }
        return acc.finish(B);
    };
    if (nu0 != 0.f) w = w - nu0 * ordered(part);
    if (nu1 != 0.f) w = w + nu1 * ordered(part + (size_t)B * E);
    w = w * decay;
    if (has_min && w < wmin) w = wmin;
    if (has_max && w > wmax) w = wmax;
    W[e] = w;
}

extern "C" int snn_conv2d_postpre(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                                  int B, int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad, float nu0, float nu1,
                                  float decay, int has_min, float wmin, int has_max, float wmax, float *ws, snn_stream_t stream) {
    if (!W || !s_src || !x_src || !s_tgt || !x_tgt || !ws || B <= 0 || Cin <= 0 || H <= 0 || Wd <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || pad < 0) return SNN_ERR_INVALID;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return SNN_ERR_INVALID;
    const long E = (long)Cout * Cin * KH * KW, n = (long)B * E;
    static const bool events = [] { const char *v = getenv("SNN_CONV_PP_EVENTS"); return !(v && v[0] == '0'); }();
    const size_t ev_lds_bytes = ((size_t)H + (size_t)8 * OH + 1) * sizeof(uint32_t) + ((size_t)8 * OH * OW + (size_t)H * Wd) * sizeof(float);
    if (events && Wd <= 32 && OW <= 32 && ev_lds_bytes <= 60 * 1024 && 8 * KH * KW <= 256) {
        const snn::ConvGeom g{Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW};
        hipLaunchKernelGGL(k_conv_pp_partial_ev, dim3((unsigned)(B * Cin), (unsigned)((Cout + 7) / 8)), dim3(256), ev_lds_bytes,
                           (hipStream_t)stream, s_src, x_src, s_tgt, x_tgt, ws, B, g);
    } else {
        hipLaunchKernelGGL(k_conv_pp_partial, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s_src, x_src, s_tgt, x_tgt, ws,
                           B, Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW);
    }
    hipLaunchKernelGGL(k_conv_pp_apply, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, ws, B, E, nu0, nu1, decay,
                       has_min, wmin, has_max, wmax);
    return snn_check_launch();
}

// MSTDP on a Conv2dConnection (learning.py:1942-2015, batch 1).  State: E = eligibility [Cout, K], P = P^+ in input space
// [Cin, H, W] (the reference's unfolded copy goes through the same elementwise operations), Q = P^- [Cout, L].
// (1) w += nu0 * sum_over_output_channels(reward * E) -- the reference's torch.sum(update, dim=0) on a weight-shaped
//     eligibility, broadcast back over the channels --, decay, clamp; one thread per kernel tap k, ATen order over co.
__global__ __launch_bounds__(256) void k_conv_mstdp_apply(float *__restrict__ W, const float *__restrict__ E, int Cout, long K, float reward,
                                                          float nu0, float wdecay, int has_min, float wmin, int has_max, float wmax) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    OuterSum acc; acc.init(k >= (K / 32) * 32);
    for (int co = 0; co < Cout; ++co) acc.add(co, reward * E[(long)co * K + k], Cout);
    const float S = acc.finish(Cout);
    for (int co = 0; co < Cout; ++co) {
        float w = W[(long)co * K + k] + nu0 * S;
        w = w * wdecay;
        if (has_min && w < wmin) w = wmin;
        if (has_max && w > wmax) w = wmax;
        W[(long)co * K + k] = w;
    }
}

// (2) P = P * decay_plus + a_plus * s_src;  Q = Q * decay_minus + a_minus * s_tgt   (learning.py:1998-2001)
__global__ __launch_bounds__(256) void k_conv_mstdp_traces(float *__restrict__ P, float *__restrict__ Q, const uint8_t *__restrict__ s_src,
                                                           const uint8_t *__restrict__ s_tgt, long nP, long nQ, float a_plus,
                                                           float a_minus, float decay_plus, float decay_minus) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nP) { const float p = P[i] * decay_plus; P[i] = p + a_plus * (float)s_src[i]; }
    else if (i < nP + nQ) { const long q = i - nP; const float v = Q[q] * decay_minus; Q[q] = v + a_minus * (float)s_tgt[q]; }
}

// (3) E[co,k] = sum_l s_tgt[co,l] * unfold(P)[k,l] + sum_l Q[co,l] * unfold(s_src)[k,l], each sum ascending in l (the
//     canonical order pinned for the reference's two torch.bmm calls, :2004-2007)
__global__ __launch_bounds__(256) void k_conv_mstdp_elig(float *__restrict__ E, const float *__restrict__ P, const float *__restrict__ Q,
                                                         const uint8_t *__restrict__ s_src, const uint8_t *__restrict__ s_tgt, int Cin,
                                                         int H, int Wd, int Cout, int KH, int KW, int stride, int pad, int OH, int OW) {
    const long K = (long)Cin * KH * KW;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)Cout * K) return;
    const int co = (int)(e / K), k = (int)(e - (long)co * K);
    const int ci = k / (KH * KW), ky = (k / KW) % KH, kx = k % KW;
    const int L = OH * OW;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < L; ++l) {
        const int oy = l / OW, ox = l - oy * OW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
        const size_t si = ((size_t)ci * H + (in ? iy : 0)) * Wd + (in ? ix : 0);
        a += (float)s_tgt[(size_t)co * L + l] * (in ? P[si] : 0.0f);
        b += Q[(size_t)co * L + l] * (in ? (float)s_src[si] : 0.0f);
    }
    E[e] = a + b;
}

extern "C" int snn_conv2d_mstdp_step(float *W, float *elig, float *p_plus, float *p_minus, const uint8_t *s_src, const uint8_t *s_tgt,
                                     int Cin, int H, int Wd, int Cout, int KH, int KW, int stride, int pad, float reward, float nu0,
                                     float a_plus, float a_minus, float decay_plus, float decay_minus, float wdecay, int has_min,
                                     float wmin, int has_max, float wmax, snn_stream_t stream) {
    if (!W || !elig || !p_plus || !p_minus || !s_src || !s_tgt || Cin <= 0 || H <= 0 || Wd <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 ||
        stride <= 0 || pad < 0) return SNN_ERR_INVALID;
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (Wd + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return SNN_ERR_INVALID;
    const long K = (long)Cin * KH * KW, nP = (long)Cin * H * Wd, nQ = (long)Cout * OH * OW;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_conv_mstdp_apply, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, W, elig, Cout, K, reward, nu0, wdecay,
                       has_min, wmin, has_max, wmax);
    hipLaunchKernelGGL(k_conv_mstdp_traces, dim3((unsigned)((nP + nQ + 255) / 256)), dim3(256), 0, st, p_plus, p_minus, s_src, s_tgt, nP, nQ,
                       a_plus, a_minus, decay_plus, decay_minus);
    hipLaunchKernelGGL(k_conv_mstdp_elig, dim3((unsigned)((Cout * K + 255) / 256)), dim3(256), 0, st, elig, p_plus, p_minus, s_src, s_tgt,
                       Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW);
    return snn_check_launch();
}

extern "C" int snn_stdp_hebbian(float *W, const uint8_t *s_src, const float *x_src, const uint8_t *s_tgt, const float *x_tgt,
                                int B, int Nin, int N, float nu0, float nu1, int weight_dependent, float decay, int has_min,
                                float wmin, int has_max, float wmax, snn_stream_t stream) {
    if (!W || !s_src || !x_src || !s_tgt || !x_tgt || B <= 0 || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (weight_dependent && !(has_min && has_max)) return SNN_ERR_INVALID;      // learning.py:600-602: finite wmin and wmax
    if (B > 256) return SNN_ERR_UNSUPPORTED;
    StdpArgs a{W, s_src, x_src, s_tgt, x_tgt, B, Nin, N, nu0, nu1, 0, 1.f, decay, has_min, wmin, has_max, wmax, 0,
               weight_dependent ? 3 : 2, 0.f, nullptr};
    return launch_plasticity(a, (hipStream_t)stream);
}

// MSTDPET (learning.py:2187-2248, batch 1): the eligibility TRACE is genuinely dense state; the point eligibility is the
// rank-2 expression of the previous step's factors and is formed on the fly.
__global__ __launch_bounds__(256) void k_mstdpet(float *__restrict__ W, float *__restrict__ e_trace, const float *__restrict__ p_plus,
                                                 const float *__restrict__ p_minus, const uint8_t *__restrict__ s_src_prev,
                                                 const uint8_t *__restrict__ s_tgt_prev, int Nin, int N, float scale, float decay_e,
                                                 float tc_e, float wdecay, int has_min, float wmin, int has_max, float wmax) {
    const long E = (long)Nin * N;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x) {
        const int i = (int)(e / N), j = (int)(e - (long)i * N);
        const float el = p_plus[i] * (float)s_tgt_prev[j] + (float)s_src_prev[i] * p_minus[j];    // :2241-2243
        float et = e_trace[e] * decay_e;                // :2223
        et = et + el / tc_e;                            // :2224
        e_trace[e] = et;
        float w = W[e] + scale * et;                    // :2226-2230
        w = w * wdecay;
        if (has_min && w < wmin) w = wmin;
        if (has_max && w > wmax) w = wmax;
        W[e] = w;
    }
}

extern "C" int snn_mstdpet_step(float *W, float *e_trace, float *p_plus, float *p_minus, uint8_t *s_src_prev, uint8_t *s_tgt_prev,
                                const uint8_t *s_src, const uint8_t *s_tgt, int Nin, int N, float reward, float nu0, float dt,
                                float a_plus, float a_minus, float decay_plus, float decay_minus, float decay_e, float tc_e,
                                float wdecay, int has_min, float wmin, int has_max, float wmax, snn_stream_t stream) {
    if (!W || !e_trace || !p_plus || !p_minus || !s_src_prev || !s_tgt_prev || !s_src || !s_tgt || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    const long E = (long)Nin * N;
    const unsigned grid = (unsigned)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    const float scale = (nu0 * dt) * reward;            // ((nu[0] * dt) * reward) * eligibility_trace
    hipLaunchKernelGGL(k_mstdpet, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, e_trace, p_plus, p_minus, s_src_prev, s_tgt_prev,
                       Nin, N, scale, decay_e, tc_e, wdecay, has_min, wmin, has_max, wmax);
    int rc = snn_check_launch();
    if (rc) return rc;
    const long n = (long)Nin + N;
    hipLaunchKernelGGL(k_mstdp_traces, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p_plus, p_minus, s_src_prev,
                       s_tgt_prev, s_src, s_tgt, (long)Nin, (long)N, a_plus, a_minus, decay_plus, decay_minus);
    return snn_check_launch();
}

extern "C" int snn_mstdp_step(float *W, float *p_plus, float *p_minus, uint8_t *s_src_prev, uint8_t *s_tgt_prev,
                              const uint8_t *s_src, const uint8_t *s_tgt, int B, int Nin, int N, float reward,
                              const float *reward_vec, float nu0, float a_plus, float a_minus, float decay_plus,
                              float decay_minus, float wdecay, int has_min, float wmin, int has_max, float wmax,
                              snn_stream_t stream) {
    if (!W || !p_plus || !p_minus || !s_src_prev || !s_tgt_prev || !s_src || !s_tgt || B <= 0 || Nin <= 0 || N <= 0)
        return SNN_ERR_INVALID;
    if (B > 256) return SNN_ERR_UNSUPPORTED;
    // (1) W += nu0 * sum_b reward * elig_prev[b]; decay; clamp -- uses the OLD p_plus/p_minus.
    StdpArgs a{W, s_src_prev, p_plus, s_tgt_prev, p_minus, B, Nin, N, nu0, 0.f, 0, 1.f, wdecay, has_min, wmin,
               has_max, wmax, 0, 1, reward, reward_vec};
    int rc = launch_plasticity(a, (hipStream_t)stream);
    if (rc) return rc;
    // (2) trace updates + remember this step's spikes as the factors of the next eligibility.
    const long n = (long)B * (Nin + N);
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_mstdp_traces, dim3(grid), dim3(256), 0, (hipStream_t)stream, p_plus, p_minus, s_src_prev,
                       s_tgt_prev, s_src, s_tgt, (long)B * Nin, (long)B * N, a_plus, a_minus, decay_plus,
                       decay_minus);
    return snn_check_launch();
}

// =============================================================================================
// a11: normalize.  k_colsum: column sums in ATen's sum(dim=0) order; k_scale: elementwise W *= norm * (1/colsum).
// =============================================================================================
// The ATen order leaves a lot of parallelism without changing a single rounding:
//  * multi_row_sum columns (j < 32*floor(N/32)): the sums of the 16-row blocks are independent of each other (each a
//    plain sequential sum of its 16 terms); only folding them through the upper cascade levels is sequential.
//  * row_sum columns: four interleaved lanes (row mod 4), each lane again a cascade over 16-position blocks.
// Workgroup = 16 consecutive columns (never straddling the class boundary, 16 | 32) x 64 block-threads: every
// thread sums one block per round (coalesced 64-byte row segments), one thread per column (or per lane) folds the
// round's block sums in block order.
constexpr int kCsCols = 16, kCsBlk = 64;
__global__ __launch_bounds__(kCsCols * kCsBlk) void k_colsum(const float *__restrict__ W, int Nin, int N, float norm,
                                                             int use_abs, float *__restrict__ scale) {
    __shared__ float bs[kCsBlk][kCsCols + 1];          // block sums of one round: [block thread][column]
    __shared__ float lanes[4][kCsCols + 1];
    const int tid = threadIdx.x, cl = tid % kCsCols, bt = tid / kCsCols;
    const int j = blockIdx.x * kCsCols + cl;
    const bool valid = j < N;
    const int jc = valid ? j : N - 1;
    const bool tail = blockIdx.x * kCsCols >= (N / 32) * 32;      // uniform per workgroup
    if (!tail) {
        const int nfull = Nin >> 4;
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;            // upper cascade levels (bt == 0 threads)
        for (int base = 0; base < nfull; base += kCsBlk) {
            const int blk = base + bt;
            if (blk < nfull) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) { const float w = W[(size_t)(blk * 16 + k) * N + jc]; v[k] = use_abs ? fabsf(w) : w; }
                float a0 = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) a0 += v[k];
                bs[bt][cl] = a0;
            }
            __syncthreads();
            if (bt == 0)
                for (int b2 = base; b2 < min(nfull, base + kCsBlk); ++b2) {
                    a1 += bs[b2 - base][cl];
                    const int m = b2 + 1;              // boundary after this block
                    if ((m & 15) == 0) { a2 += a1; a1 = 0.f; if ((m & 255) == 0) { a3 += a2; a2 = 0.f; } }
                }
            __syncthreads();
        }
        if (bt == 0 && valid) {
            float a0 = 0.f;                            // rows past the last complete block
            for (int i = nfull * 16; i < Nin; ++i) { const float w = W[(size_t)i * N + jc]; a0 += use_abs ? fabsf(w) : w; }
            float cs = ((a0 + a1) + a2) + a3;
            if (cs == 0.f) cs = 1.0f;                   // topology_features.py:265
            scale[j] = (1.0f / cs) * norm;              // torch: python_scalar / tensor == reciprocal * scalar
        }
        return;
    }
    // row_sum columns: thread = (column cl, lane s4, block group bg); lane s4 owns the rows 4p + s4, p < n4
    const int s4 = bt & 3, bg = bt >> 2;               // 16 block groups
    const int n4 = Nin >> 2, nf4 = n4 >> 4;
    float a1 = 0.f, a2 = 0.f, a3 = 0.f;                // (bg == 0 threads)
    for (int base = 0; base < nf4; base += kCsBlk / 4) {
        const int blk = base + bg;
        if (blk < nf4) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float w = W[(size_t)(4 * (blk * 16 + k) + s4) * N + jc]; v[k] = use_abs ? fabsf(w) : w; }
            float a0 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a0 += v[k];
            bs[bt][cl] = a0;
        }
        __syncthreads();
        if (bg == 0)
            for (int b2 = base; b2 < min(nf4, base + kCsBlk / 4); ++b2) {
                a1 += bs[(b2 - base) * 4 + s4][cl];
                const int m = b2 + 1;
                if ((m & 15) == 0) { a2 += a1; a1 = 0.f; if ((m & 255) == 0) { a3 += a2; a2 = 0.f; } }
            }
        __syncthreads();
    }
    if (bg == 0) {
        float a0 = 0.f;                                // positions past the lane's last complete block
        for (int p_ = nf4 * 16; p_ < n4; ++p_) { const float w = W[(size_t)(4 * p_ + s4) * N + jc]; a0 += use_abs ? fabsf(w) : w; }
        float lane = ((a0 + a1) + a2) + a3;
        if (s4 == 0)                                   // the Nin % 4 leftover rows join lane 0 after its cascade
            for (int i = n4 * 4; i < Nin; ++i) { const float w = W[(size_t)i * N + jc]; lane += use_abs ? fabsf(w) : w; }
        lanes[s4][cl] = lane;
    }
    __syncthreads();
    if (bt == 0 && valid) {
        float cs = ((lanes[0][cl] + lanes[1][cl]) + lanes[2][cl]) + lanes[3][cl];
        if (cs == 0.f) cs = 1.0f;
        scale[j] = (1.0f / cs) * norm;
    }
}

__global__ __launch_bounds__(256) void k_scale_cols(float *__restrict__ W, long E, int N,
                                                    const float *__restrict__ scale) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (long)gridDim.x * blockDim.x)
        W[e] = W[e] * scale[e % N];
}

extern "C" int snn_normalize(float *W, int Nin, int N, float norm, int use_abs, float *colsum_ws,
                             snn_stream_t stream) {
    if (!W || !colsum_ws || Nin <= 0 || N <= 0) return SNN_ERR_INVALID;
    if (Nin > kMaxTerms) return SNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_colsum, dim3((N + kCsCols - 1) / kCsCols), dim3(kCsCols * kCsBlk), 0, (hipStream_t)stream, W, Nin, N,
                       norm, use_abs, colsum_ws);
    int rc = snn_check_launch();
    if (rc) return rc;
    const long E = (long)Nin * N;
    const unsigned grid = (unsigned)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_scale_cols, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, E, N, colsum_ws);
    return snn_check_launch();
}

// a11 for Conv2dConnection: normalize(), bindsnet/network/topology.py:824-837.  Every [KH*KW] filter of the [Cout*Cin, KH*KW] view
// is scaled to sum `norm`: w[f] *= norm / w[f].sum(0), where the sum is ATen's vectorised INNER sum (snn_order.hpp inner_sum8) and
// torch evaluates float / tensor as reciprocal(sum) * norm.  No zero guard (the reference has none).  One thread per filter: the
// weights of a convolution are a few hundred floats.  The body is __host__ __device__ and checked on the host against the reference
// fixture (tests/test_order_rng_host.py); first run on an MI355X in round 4 (reference fixture bit for bit, profiles/r04_experimental_suite_mi355x.log).
__global__ __launch_bounds__(64) void k_normalize_filters(float *__restrict__ W, int F, int K, float norm) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    float *w = W + (size_t)f * K;
    const float sum = inner_sum8(w, K);
    const float rc = 1.0f / sum;
    const float scale = rc * norm;
    for (int k = 0; k < K; ++k) w[k] = w[k] * scale;
}

extern "C" int snn_normalize_conv2d(float *W, int n_filters, int taps, float norm, snn_stream_t stream) {
    if (!W || n_filters <= 0 || taps <= 0) return SNN_ERR_INVALID;
    if (taps > kMaxTerms) return SNN_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_normalize_filters, dim3((unsigned)((n_filters + 63) / 64)), dim3(64), 0, (hipStream_t)stream, W, n_filters, taps, norm);
    return snn_check_launch();
}

// =============================================================================================
// Network.reset_state_variables(): every state tensor of every layer filled in ONE launch
// (bindsnet/network/network.py:467-481 -> nodes.py:109-120, 531-538, 1113-1120: s, x, refrac_count <- 0, v <- rest).
// =============================================================================================
namespace {
struct FillArgs { snn_fill_segment seg[SNN_MAX_FILL_SEGMENTS]; unsigned first_block[SNN_MAX_FILL_SEGMENTS + 1]; int n; };

__global__ __launch_bounds__(256) void k_fill_segments(const FillArgs a) {
    int k = 0;
    while (k + 1 < a.n && blockIdx.x >= a.first_block[k + 1]) ++k;          // (n <= 32: a short uniform scan)
    const snn_fill_segment sg = a.seg[k];
    const size_t off = ((size_t)(blockIdx.x - a.first_block[k]) * 256 + threadIdx.x) * 16;
    if (off >= sg.bytes) return;
    unsigned char *p = (unsigned char *)sg.ptr + off;
    if (off + 16 <= sg.bytes && (((uintptr_t)p) & 15) == 0) {
        *(uint4 *)p = make_uint4(sg.pattern, sg.pattern, sg.pattern, sg.pattern);
    } else {
        const size_t n = sg.bytes - off < 16 ? sg.bytes - off : 16;
        for (size_t b = 0; b < n; ++b) p[b] = (unsigned char)(sg.pattern >> (8 * ((off + b) & 3)));
    }
}
}  // namespace

extern "C" int snn_fill_segments(const snn_fill_segment *h_segs, int n, snn_stream_t stream) {
    if (n < 0 || (n > 0 && !h_segs)) return SNN_ERR_INVALID;
    if (n > SNN_MAX_FILL_SEGMENTS) return SNN_ERR_UNSUPPORTED;
    FillArgs a;
    a.n = 0;
    unsigned blocks = 0;
    for (int k = 0; k < n; ++k) {
        if (!h_segs[k].bytes) continue;
        if (!h_segs[k].ptr) return SNN_ERR_INVALID;
        // a non-zero pattern is a 32-bit value: the buffer must then be a whole number of aligned words
        if (h_segs[k].pattern && ((h_segs[k].bytes & 3) || (((uintptr_t)h_segs[k].ptr) & 3))) return SNN_ERR_INVALID;
        a.seg[a.n] = h_segs[k];
        a.first_block[a.n] = blocks;
        blocks += (unsigned)((h_segs[k].bytes + 4095) / 4096);
        ++a.n;
    }
    if (!a.n) return SNN_OK;
    a.first_block[a.n] = blocks;
    hipLaunchKernelGGL(k_fill_segments, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return snn_check_launch();
}
