// snn_convlif.hip -- fused plan "convlif-fused": Input -> Conv2dConnection (no update rule) -> LIFNodes,
// i.e. the loop body of bindsnet/network/network.py:380-461 for the graph of BASELINE cfg4
// (Conv2dConnection.compute, bindsnet/network/topology.py:799-815 = F.conv2d; LIFNodes.forward,
// bindsnet/network/nodes.py:500-529).
//
// Without a learning rule nothing couples two output neurons, and the only thing an output neuron needs
// from outside is the input spike image of the previous step.  So the WHOLE run is one launch:
//   workgroup  <->  (sample b, chunk of 8 output channels, tile of 256 output pixels)
//   thread     <->  one output pixel of the tile; the membrane state (v, refrac, trace) of its 8 neurons
//                   lives in registers for all T steps
// Per step the workgroup stages the sample's [Cin,H,W] spike image in LDS (double buffered: one barrier per
// step), every thread walks its KHxKW window in the reference's tap order (cin, kh, kw; zero taps skipped:
// x + 0.0f == x), steps its 8 LIF neurons and writes the spike / voltage rasters.  No state or current
// traffic to HBM inside the loop: the only per-step HBM traffic is the input image and the monitors.
// Results are bit-identical to the generic plan (k_conv2d + k_lif per step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include "../../include/snnhip.h"
#include "snn_common.hpp"
#include "snn_order.hpp"
#include "snn_conv_events.hpp"

using namespace snn;

void snn_set_plan_name(const char *name);
bool snn_prof_active();

namespace {

constexpr int NTC = 256;     // threads per workgroup = output pixels per tile
constexpr int CC = 8;        // output channels per workgroup

struct ConvCtx {
    int B, T, Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW, ntile, nchunk;
    const uint8_t *in;        // [T,B,Cin,H,W]
    const uint8_t *sX0;       // [B,Cin,H,W] input layer's spikes at entry
    float *xX; int x_traces; float x_decay, x_scale; int x_additive;   // input trace (only its final value matters)
    const float *W, *bias;
    float *v, *refrac, *x; uint8_t *s;     // [B,Cout,OH,OW]
    snn_lif_params p;
    uint8_t *ras; float *rasV;             // nullable [T,B,Cout,OH,OW]
};

// Input trace after the run (nodes.py:96-103); no rule reads it in between.
__global__ __launch_bounds__(256) void k_conv_xtrace(const ConvCtx c) {
    const long n = (long)c.B * c.Cin * c.H * c.Wd;
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    float x = c.xX[k];
    int t = 0;
    for (; t + 8 <= c.T; t += 8) {
        uint8_t sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = c.in[(size_t)(t + u) * n + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) x = trace_next(x, sv[u], c.x_decay, c.x_scale, c.x_additive);
    }
    for (; t < c.T; ++t) x = trace_next(x, c.in[(size_t)t * n + k], c.x_decay, c.x_scale, c.x_additive);
    c.xX[k] = x;
}

// KH_ x KW_ > 0: window size known at compile time (taps fully unrolled: the window's LDS reads are issued
// together and their addresses fold into immediates); 0: any size.
template <int KH_, int KW_>
__global__ __launch_bounds__(NTC) void k_convlif_run(const ConvCtx c) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int taps = c.Cin * c.KH * c.KW, img = c.Cin * c.H * c.Wd;
    const int imgw = (img + 3) / 4;                       // image staged as 32-bit words
    float *wl = (float *)smem;                            // [CC][taps] this chunk's filters
    uint32_t *im = (uint32_t *)(wl + CC * taps);          // [2][imgw]
    const int tid = threadIdx.x;
    int g = blockIdx.x;
    const int tile = g % c.ntile; g /= c.ntile;
    const int chunk = g % c.nchunk; const int b = g / c.nchunk;
    const int c0 = chunk * CC;
    const int pix = tile * NTC + tid, npix = c.OH * c.OW;
    const bool valid = pix < npix;
    const int oy = valid ? pix / c.OW : 0, ox = valid ? pix - oy * c.OW : 0;
    for (int k = tid; k < CC * taps; k += NTC) { const int cc = k / taps; wl[k] = (c0 + cc < c.Cout) ? c.W[(size_t)(c0 + cc) * taps + (k - cc * taps)] : 0.f; }
    float v[CC], rf[CC], xt[CC], bs[CC]; bool last[CC];
    const size_t nB = (size_t)c.Cout * npix;              // neurons per sample
#pragma unroll
    for (int u = 0; u < CC; ++u) {
        v[u] = rf[u] = xt[u] = 0.f; last[u] = false;
        bs[u] = (c.bias && c0 + u < c.Cout) ? c.bias[c0 + u] : 0.f;
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * npix + pix;
            v[u] = c.v[k]; rf[u] = c.refrac[k]; last[u] = c.s[k] != 0;
            if (c.p.traces) xt[u] = c.x[k];
        }
    }
    // stage the image of iteration 0 (the input layer's spikes at entry)
    {
        const uint8_t *src = c.sX0 + (size_t)b * img;
        for (int k = tid; k < imgw; k += NTC) {
            uint32_t w = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (4 * k + q < img) w |= (uint32_t)src[4 * k + q] << (8 * q);
            im[k] = w;
        }
    }
    __syncthreads();
    for (int t = 0; t < c.T; ++t) {
        const uint32_t *cur = im + (t & 1) * imgw;
        uint32_t *nxt = im + ((t + 1) & 1) * imgw;
        // prefetch the next iteration's image (= the input of THIS step) while this one is consumed
        uint32_t pre[4] = {0u, 0u, 0u, 0u};                // imgw <= 4 * NTC (host check)
        if (t + 1 < c.T) {
            const uint8_t *src = c.in + ((size_t)t * c.B + b) * img;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = tid + r * NTC;
                if (k < imgw) {
                    if (4 * k + 3 < img && (img & 3) == 0) pre[r] = *(const uint32_t *)(src + 4 * k);
                    else { uint32_t w = 0; for (int q = 0; q < 4; ++q) if (4 * k + q < img) w |= (uint32_t)src[4 * k + q] << (8 * q); pre[r] = w; }
                }
            }
        }
        if (valid) {
            float acc[CC];
#pragma unroll
            for (int u = 0; u < CC; ++u) acc[u] = 0.f;
            const uint8_t *ib = (const uint8_t *)cur;
            if constexpr (KH_ > 0) {
                const int y0 = oy * c.stride - c.pad, x0 = ox * c.stride - c.pad;
                for (int ci = 0; ci < c.Cin; ++ci) {
                    // which taps of the window carry a spike: spikes are sparse, so the accumulation below runs
                    // only over the set bits (ascending = the reference's tap order; a zero tap adds nothing)
                    const uint8_t *ic = ib + ci * c.H * c.Wd;
                    uint32_t m = 0;
                    if (c.pad == 0) {
#pragma unroll
                        for (int k = 0; k < KH_ * KW_; ++k)
                            m |= (uint32_t)(ic[(y0 + k / KW_) * c.Wd + x0 + k % KW_] != 0) << k;
                    } else {
#pragma unroll
                        for (int k = 0; k < KH_ * KW_; ++k) {
                            const int iy = y0 + k / KW_, ix = x0 + k % KW_;
                            const bool in = iy >= 0 && iy < c.H && ix >= 0 && ix < c.Wd;
                            m |= (uint32_t)(in && ic[(in ? iy : 0) * c.Wd + (in ? ix : 0)] != 0) << k;
                        }
                    }
                    const float *wq = wl + ci * (KH_ * KW_);
                    while (m) {
                        const int k = __ffs(m) - 1; m &= m - 1;
                        const float fs = (float)ic[(y0 + k / KW_) * c.Wd + x0 + k % KW_];
#pragma unroll
                        for (int u = 0; u < CC; ++u) acc[u] += fs * wq[u * taps + k];
                    }
                }
            } else {
                for (int ky = 0; ky < c.KH; ++ky) {       // reference order: taps row-major, input channels innermost
                    const int iy = oy * c.stride - c.pad + ky;
                    for (int kx = 0; kx < c.KW; ++kx)
                        for (int ci = 0; ci < c.Cin; ++ci) {
                            const int tap = (ci * c.KH + ky) * c.KW + kx;
                            const int ix = ox * c.stride - c.pad + kx;
                            if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.Wd) continue;
                            const uint8_t sv = ib[(ci * c.H + iy) * c.Wd + ix];
                            if (!sv) continue;
                            const float fs = (float)sv;
#pragma unroll
                            for (int u = 0; u < CC; ++u) acc[u] += fs * wl[u * taps + tap];
                        }
                    }
            }
#pragma unroll
            for (int u = 0; u < CC; ++u) {
                if (c0 + u >= c.Cout) continue;
                float r = acc[u];
                if (c.bias) r = r + bs[u];
                float cur_in = 0.0f + r;                   // zeros + conv (network.py:225-248)
                if (rf[u] > 0.f) cur_in = 0.f;             // nodes.py:511
                const bool sp = lif_update(v[u], rf[u], cur_in, c.p);
                last[u] = sp;
                if (c.p.traces) xt[u] = trace_next(xt[u], sp, c.p.trace_decay, c.p.trace_scale, c.p.traces_additive);
                const size_t k = ((size_t)t * c.B + b) * nB + (size_t)(c0 + u) * npix + pix;
                if (c.ras) c.ras[k] = sp;
                if (c.rasV) c.rasV[k] = v[u];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int k = tid + r * NTC; if (k < imgw) nxt[k] = pre[r]; }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < CC; ++u)
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * npix + pix;
            c.v[k] = v[u]; c.refrac[k] = rf[u]; c.s[k] = last[u];
            if (c.p.traces) c.x[k] = xt[u];
        }
}


// ======================================================================================================================================
// Plan "convpp-fused" (round 6): Input -> Conv2dConnection with PostPre (bindsnet/learning/learning.py:457-497) -> LIFNodes, the graph
// examples/mnist/conv_mnist.py trains, as ONE cooperative launch for the whole run.  The generic plan runs five launches per timestep
// (input step, convolution, LIF step, per-sample partial sums, batch reduction + apply).
//
//   workgroup  <->  (sample b, chunk of CC output channels), ALL output pixels: the partial sums of a weight element run over the output
//                   positions in ascending order, one chain per (sample, element) -- a workgroup that holds every position of its channels
//                   has that chain to itself
//   thread     <->  one output pixel; v / refrac / trace of its CC neurons in registers for all T steps
//   per step:  (1) convolution with the chunk's CURRENT filters (LDS) + LIF step, as in k_convlif_run; the new target spikes and traces go to
//                  LDS, the step's input image (= the "after" spikes PostPre pairs with) is staged meanwhile and the sample's input trace
//                  advanced (LDS);
//              (2) the chunk's CC * Cin * KH * KW elements: their two partial sums over the positions (snn_conv_events.hpp: the bodies
//                  k_conv_pp_partial_ev runs, on the same operands) -> the exchange area [parity][2][B][E] in the workspace;
//              (3) ONE hand-off per step among the B workgroups of a chunk (a counter per chunk: release behind the stores, acquire in front
//                  of the loads); every workgroup then reduces the B samples' partial sums of ITS chunk in ATen's order and applies rates,
//                  decay and clamp to its own copy of the filters (k_conv_pp_apply's statements) -- B identical copies instead of a second
//                  hand-off to spread the result.
// Bit-identical to the generic plan: same bodies, same order.  The exchange area has two halves by step parity: a workgroup may be writing
// step t+1 while a slower one still reads step t; to write step t+2 it has to pass the hand-off of step t+1, which needs that reader.
struct ConvPPCtx {
    ConvCtx c;
    float nu0, nu1, wdecay, wmin, wmax;
    int has_min, has_max, learning;
    float *Wout;              // the connection's weights (written back by the b == 0 workgroups)
    float *part;              // [2][2][B][E]
    unsigned *cnt;            // [nchunk] arrivals, zeroed before the launch
    int *status;              // nullable
    long long *dbg;           // developer aid (SNN_CONVPP_TIMING=1): [8] phase totals of workgroup 0 in 10 ns ticks
};

constexpr int kPPPoll = 4000000;      // polls of the chunk's counter before a workgroup gives up (a co-resident grid arrives within microseconds)

template <int KH_, int KW_>
__global__ __launch_bounds__(1024) void k_convpp_run(const ConvPPCtx a) {
    const ConvCtx &c = a.c;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NT = (int)blockDim.x;
    const int KK = c.KH * c.KW, taps = c.Cin * KK, img = c.Cin * c.H * c.Wd, L = c.OH * c.OW;
    const int imgw = (img + 3) / 4;
    float *wl = (float *)smem;                            // [CC][taps] this chunk's filters: the workgroup's own copy for the whole run
    uint32_t *im = (uint32_t *)(wl + CC * taps);          // [2][imgw] input images (bytes), by step parity
    float *xs = (float *)(im + 2 * imgw);                 // [img] the sample's input trace
    float *xt = xs + img;                                 // [CC][L] target traces after this step
    uint32_t *srow = (uint32_t *)(xt + CC * L);           // [Cin * H] packed source rows
    uint32_t *trow = srow + c.Cin * c.H;                  // [CC * OH] packed target rows
    int *multi = (int *)(trow + CC * c.OH);               // [4]: [0] a spike byte that is neither 0 nor 1 this step, [1] give up
    uint8_t *sbt = (uint8_t *)(multi + 4);                // [CC][L] target spikes of this step (bytes)
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x % c.nchunk, b = blockIdx.x / c.nchunk;
    const int c0 = chunk * CC, nco = min(CC, c.Cout - c0);
    const int pix = tid, npix = L;
    const bool valid = pix < npix;
    const int oy = valid ? pix / c.OW : 0, ox = valid ? pix - oy * c.OW : 0;
    const long K = taps, E = (long)c.Cout * K;
    for (int k = tid; k < CC * taps; k += NT) { const int cc = k / taps; wl[k] = (c0 + cc < c.Cout) ? c.W[(size_t)(c0 + cc) * taps + (k - cc * taps)] : 0.f; }
    for (int k = tid; k < img; k += NT) xs[k] = c.x_traces ? c.xX[(size_t)b * img + k] : 0.f;
    if (tid < 4) multi[tid] = 0;
    float v[CC], rf[CC], xtr[CC], bs[CC]; bool last[CC];
    const size_t nB = (size_t)c.Cout * npix;
#pragma unroll
    for (int u = 0; u < CC; ++u) {
        v[u] = rf[u] = xtr[u] = 0.f; last[u] = false;
        bs[u] = (c.bias && c0 + u < c.Cout) ? c.bias[c0 + u] : 0.f;
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * npix + pix;
            v[u] = c.v[k]; rf[u] = c.refrac[k]; last[u] = c.s[k] != 0;
            if (c.p.traces) xtr[u] = c.x[k];
        }
    }
    {
        const uint8_t *src = c.sX0 + (size_t)b * img;
        for (int k = tid; k < imgw; k += NT) {
            uint32_t w = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (4 * k + q < img) w |= (uint32_t)src[4 * k + q] << (8 * q);
            im[k] = w;
        }
    }
    __syncthreads();
    const ConvGeom g{c.Cin, c.H, c.Wd, c.Cout, c.KH, c.KW, c.stride, c.pad, c.OH, c.OW};
    // bit x of the result = (row[x] != 0); *flag is set when a byte is neither 0 nor 1 (k_conv_pp_partial_ev's pack)
    auto pack = [&](const uint8_t *row, int n, int &flag) -> uint32_t {
        if ((((uintptr_t)row) & 3) != 0) return conv_pack_row(row, n, &flag);
        uint32_t m = 0;
        int x = 0;
        for (; x + 4 <= n; x += 4) {
            const uint32_t w = *(const uint32_t *)(row + x);
            if (w & 0xFEFEFEFEu) flag = 1;
            const uint32_t nz = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
            m |= (((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xFu) << x;
        }
        for (; x < n; ++x) { const uint8_t w = row[x]; m |= (uint32_t)(w != 0) << x; if (w > 1) flag = 1; }
        return m;
    };
    bool dead = false;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
    const bool timing = a.dbg && blockIdx.x == 0 && tid == 0;
#define PPMARK(k) do { if (timing) { const long long now_ = (long long)wall_clock64(); ph[k] += now_ - tk; tk = now_; } } while (0)
    if (timing) tk = (long long)wall_clock64();
    for (int t = 0; t < c.T; ++t) {
        const uint32_t *cur = im + (t & 1) * imgw;
        uint32_t *nxt = im + ((t + 1) & 1) * imgw;
        // this step's input image: the convolution of the NEXT step reads it, PostPre of THIS step pairs with it
        uint32_t pre[4] = {0u, 0u, 0u, 0u};                // imgw <= 4 * NT (host check)
        {
            const uint8_t *src = c.in + ((size_t)t * c.B + b) * img;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = tid + r * NT;
                if (k < imgw) {
                    if (4 * k + 3 < img && (img & 3) == 0) pre[r] = *(const uint32_t *)(src + 4 * k);
                    else { uint32_t w = 0; for (int q = 0; q < 4; ++q) if (4 * k + q < img) w |= (uint32_t)src[4 * k + q] << (8 * q); pre[r] = w; }
                }
            }
        }
        if (valid) {
            float acc[CC];
#pragma unroll
            for (int u = 0; u < CC; ++u) acc[u] = 0.f;
            const uint8_t *ib = (const uint8_t *)cur;
            if constexpr (KH_ > 0) {
                const int y0 = oy * c.stride - c.pad, x0 = ox * c.stride - c.pad;
                for (int ci = 0; ci < c.Cin; ++ci) {
                    const uint8_t *ic = ib + ci * c.H * c.Wd;
                    uint32_t m = 0;
                    if (c.pad == 0) {
#pragma unroll
                        for (int k = 0; k < KH_ * KW_; ++k)
                            m |= (uint32_t)(ic[(y0 + k / KW_) * c.Wd + x0 + k % KW_] != 0) << k;
                    } else {
#pragma unroll
                        for (int k = 0; k < KH_ * KW_; ++k) {
                            const int iy = y0 + k / KW_, ix = x0 + k % KW_;
                            const bool in = iy >= 0 && iy < c.H && ix >= 0 && ix < c.Wd;
                            m |= (uint32_t)(in && ic[(in ? iy : 0) * c.Wd + (in ? ix : 0)] != 0) << k;
                        }
                    }
                    const float *wq = wl + ci * (KH_ * KW_);
                    while (m) {
                        const int k = __ffs(m) - 1; m &= m - 1;
                        const float fs = (float)ic[(y0 + k / KW_) * c.Wd + x0 + k % KW_];
#pragma unroll
                        for (int u = 0; u < CC; ++u) acc[u] += fs * wq[u * taps + k];
                    }
                }
            } else {
                for (int ky = 0; ky < c.KH; ++ky) {       // reference order: taps row-major, input channels innermost
                    const int iy = oy * c.stride - c.pad + ky;
                    for (int kx = 0; kx < c.KW; ++kx)
                        for (int ci = 0; ci < c.Cin; ++ci) {
                            const int tap = (ci * c.KH + ky) * c.KW + kx;
                            const int ix = ox * c.stride - c.pad + kx;
                            if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.Wd) continue;
                            const uint8_t sv = ib[(ci * c.H + iy) * c.Wd + ix];
                            if (!sv) continue;
                            const float fs = (float)sv;
#pragma unroll
                            for (int u = 0; u < CC; ++u) acc[u] += fs * wl[u * taps + tap];
                        }
                    }
            }
#pragma unroll
            for (int u = 0; u < CC; ++u) {
                if (c0 + u >= c.Cout) continue;
                float r = acc[u];
                if (c.bias) r = r + bs[u];
                float cur_in = 0.0f + r;                   // zeros + conv (network.py:225-248)
                if (rf[u] > 0.f) cur_in = 0.f;             // nodes.py:511
                const bool sp = lif_update(v[u], rf[u], cur_in, c.p);
                last[u] = sp;
                if (c.p.traces) xtr[u] = trace_next(xtr[u], sp, c.p.trace_decay, c.p.trace_scale, c.p.traces_additive);
                const size_t k = ((size_t)t * c.B + b) * nB + (size_t)(c0 + u) * npix + pix;
                if (c.ras) c.ras[k] = sp;
                if (c.rasV) c.rasV[k] = v[u];
                sbt[u * L + pix] = sp ? 1 : 0;
                xt[u * L + pix] = xtr[u];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int k = tid + r * NT; if (k < imgw) nxt[k] = pre[r]; }
        __syncthreads();                                   // image t, target spikes and traces of step t are in LDS
        PPMARK(0);
        // ---- the input trace after this step (nodes.py:96-103), the packed rows
        {
            const uint8_t *nb = (const uint8_t *)nxt;
            if (c.x_traces) for (int k = tid; k < img; k += NT) xs[k] = trace_next(xs[k], nb[k], c.x_decay, c.x_scale, c.x_additive);
            if (!a.learning) continue;                     // (no rule: the convolution of the next step is all that reads this step's results)
            int flag = 0;
            for (int r = tid; r < c.Cin * c.H; r += NT) srow[r] = pack(nb + (size_t)r * c.Wd, c.Wd, flag);
            for (int r = tid; r < nco * c.OH; r += NT) { const int cl = r / c.OH, y = r - cl * c.OH; trow[r] = pack(sbt + cl * L + y * c.OW, c.OW, flag); }
            if (flag) atomicOr(&multi[0], 1);
        }
        __syncthreads();
        PPMARK(1);
        // ---- partial sums of the chunk's elements for this sample -> exchange area
        float *pa = a.part + (size_t)(t & 1) * 2 * c.B * E;
        const int nel = nco * taps;
        if (tid < nel) {
            const int cl = tid / taps, kq = tid - cl * taps, ci = kq / KK, kk = kq - ci * KK, ky = kk / c.KW, kx = kk - ky * c.KW;
            float sa, sp_;
            const uint8_t *nb = (const uint8_t *)nxt;
            if (multi[0] == 0) conv_pp_events(g, ky, kx, srow + ci * c.H, trow + cl * c.OH, xs + ci * c.H * c.Wd, xt + cl * L, &sa, &sp_);
            else conv_pp_dense(g, ky, kx, nb + ci * c.H * c.Wd, xs + ci * c.H * c.Wd, sbt + cl * L, xt + cl * L, &sa, &sp_);
            const long id = (long)b * E + (long)(c0 + cl) * K + kq;
            pa[id] = sa;
            pa[(size_t)c.B * E + id] = sp_;
        }
        __syncthreads();                                   // (orders every thread's stores in front of thread 0's release; multi[0] was read)
        PPMARK(2);
        if (tid == 0) {
            multi[0] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&a.cnt[chunk], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            PPMARK(3);
            const unsigned want = (unsigned)(t + 1) * (unsigned)c.B;
            int spins = 0;
            while (__hip_atomic_load(&a.cnt[chunk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > kPPPoll) { multi[1] = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        PPMARK(4);
        if (multi[1]) { dead = true; break; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // ---- batch reduction in ATen's sum(dim=0) order + apply (k_conv_pp_apply's statements) on the own copy of the filters
        if (tid < nel) {
            const int cl = tid / taps, kq = tid - cl * taps;
            const long e = (long)(c0 + cl) * K + kq;
            const bool tail = e >= (E / 32) * 32;
            float w = wl[cl * taps + kq];
            auto ordered = [&](const float *base) {
                OuterSum accs; accs.init(tail);
                for (int b0 = 0; b0 < c.B; b0 += 16) {
                    float vv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) vv[u] = base[(size_t)min(b0 + u, c.B - 1) * E + e];
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (b0 + u < c.B) accs.add(b0 + u, vv[u], c.B);
                }
                return accs.finish(c.B);
            };
            if (a.nu0 != 0.f) w = w - a.nu0 * ordered(pa);
            if (a.nu1 != 0.f) w = w + a.nu1 * ordered(pa + (size_t)c.B * E);
            w = w * a.wdecay;
            if (a.has_min && w < a.wmin) w = a.wmin;
            if (a.has_max && w > a.wmax) w = a.wmax;
            wl[cl * taps + kq] = w;
        }
        __syncthreads();                                   // the filters of step t+1 are in place; the row tables are free
        PPMARK(5);
    }
    if (timing) for (int k = 0; k < 8; ++k) a.dbg[k] = ph[k];
    if (dead) { if (tid == 0 && a.status) atomicCAS(a.status, 0, SNN_ERR_TIMEOUT); return; }
#pragma unroll
    for (int u = 0; u < CC; ++u)
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * npix + pix;
            c.v[k] = v[u]; c.refrac[k] = rf[u]; c.s[k] = last[u];
            if (c.p.traces) c.x[k] = xtr[u];
        }
    if (a.learning && b == 0) for (int k = tid; k < nco * taps; k += NT) a.Wout[(size_t)c0 * taps + k] = wl[k];
    if (chunk == 0 && c.x_traces) for (int k = tid; k < img; k += NT) c.xX[(size_t)b * img + k] = xs[k];
}

size_t convpp_lds(const ConvCtx &c) {
    const int taps = c.Cin * c.KH * c.KW, img = c.Cin * c.H * c.Wd, L = c.OH * c.OW;
    return (size_t)CC * taps * 4 + (size_t)2 * ((img + 3) / 4) * 4 + (size_t)img * 4 + (size_t)CC * L * 4 + ((size_t)c.Cin * c.H + (size_t)CC * c.OH + 4) * 4 +
           (size_t)CC * L + 16;
}

// the graph and the shapes the plan takes; fills the geometry
bool convpp_match(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, ConvCtx &c) {
    if (nL != 2 || nC != 1) return false;
    if (L[0].kind != SNN_LAYER_INPUT || L[1].kind != SNN_LAYER_LIF) return false;
    const snn_conn_desc &d = C[0];
    if (d.kind != SNN_CONN_CONV2D || d.src != 0 || d.dst != 1 || d.rule != SNN_RULE_POSTPRE || d.has_norm || d.mask || d.raster_w) return false;
    if (R->T < 1 || R->one_step || R->B < 1 || R->B > 256) return false;
    if (L[0].clamp || L[0].unclamp || L[0].inject_v || L[0].ext_current || L[1].clamp || L[1].unclamp || L[1].inject_v || L[1].ext_current || L[1].thresh_vec) return false;
    memset(&c, 0, sizeof(c));
    c.B = R->B; c.T = R->T; c.Cin = d.cin; c.H = d.h; c.Wd = d.wd; c.Cout = d.cout; c.KH = d.kh; c.KW = d.kw;
    c.stride = d.stride; c.pad = d.pad;
    c.OH = (d.h + 2 * d.pad - d.kh) / d.stride + 1; c.OW = (d.wd + 2 * d.pad - d.kw) / d.stride + 1;
    if (c.OH <= 0 || c.OW <= 0 || c.Wd > 32 || c.OW > 32) return false;          // packed rows: one 32-bit word per image row
    const int img = c.Cin * c.H * c.Wd, taps = c.Cin * c.KH * c.KW, npix = c.OH * c.OW;
    if (L[0].n != img || L[1].n != c.Cout * npix) return false;
    if (npix > 1024 || CC * taps > 1024) return false;                           // a thread per output pixel / per element of the chunk
    const int nt = max((npix + 63) / 64 * 64, (CC * taps + 63) / 64 * 64);
    if ((img + 3) / 4 > 4 * nt) return false;
    if (convpp_lds(c) > 150 * 1024) return false;
    if ((double)R->T * R->B * L[1].n >= 9.0e15) return false;
    c.ntile = 1; c.nchunk = (c.Cout + CC - 1) / CC;
    if (!L[0].x || !L[1].x || !L[0].p.lif.traces || !L[1].p.lif.traces) return false;   // PostPre reads both traces
    return true;
}
size_t convpp_workspace(const ConvCtx &c) {
    const size_t E = (size_t)c.Cout * c.Cin * c.KH * c.KW;
    return (size_t)4 * c.B * E * sizeof(float) + 256 + (size_t)c.nchunk * sizeof(unsigned);
}

}  // namespace

int snn_try_fused_convlif(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R,
                          hipStream_t st, int *handled) {
    *handled = 0;
    if (nL != 2 || nC != 1) return SNN_OK;
    if (L[0].kind != SNN_LAYER_INPUT || L[1].kind != SNN_LAYER_LIF) return SNN_OK;
    const snn_conn_desc &d = C[0];
    if (d.kind != SNN_CONN_CONV2D || d.src != 0 || d.dst != 1 || d.rule != SNN_RULE_NONE || d.has_norm) return SNN_OK;
    if (R->T < 1) return SNN_OK;
    ConvCtx c;
    memset(&c, 0, sizeof(c));
    c.B = R->B; c.T = R->T; c.Cin = d.cin; c.H = d.h; c.Wd = d.wd; c.Cout = d.cout; c.KH = d.kh; c.KW = d.kw;
    c.stride = d.stride; c.pad = d.pad;
    c.OH = (d.h + 2 * d.pad - d.kh) / d.stride + 1; c.OW = (d.wd + 2 * d.pad - d.kw) / d.stride + 1;
    if (c.OH <= 0 || c.OW <= 0) return SNN_OK;
    const int img = c.Cin * c.H * c.Wd, taps = c.Cin * c.KH * c.KW;
    if (L[0].n != img || L[1].n != c.Cout * c.OH * c.OW) return SNN_OK;
    if ((img + 3) / 4 > 4 * NTC) return SNN_OK;                     // image words staged 4 per thread
    const size_t lds = (size_t)CC * taps * 4 + (size_t)2 * ((img + 3) / 4) * 4;
    if (lds > 60 * 1024) return SNN_OK;
    if ((double)R->T * R->B * L[1].n >= 9.0e15) return SNN_OK;
    c.ntile = (c.OH * c.OW + NTC - 1) / NTC; c.nchunk = (c.Cout + CC - 1) / CC;
    const long long grid = (long long)c.B * c.nchunk * c.ntile;
    if (grid > 2000000000ll) return SNN_OK;
    c.in = L[0].ext_spikes; c.sX0 = L[0].s;
    c.xX = L[0].x; c.x_traces = L[0].p.lif.traces && L[0].x; c.x_decay = L[0].p.lif.trace_decay;
    c.x_scale = L[0].p.lif.trace_scale; c.x_additive = L[0].p.lif.traces_additive;
    c.W = d.w; c.bias = d.bias;
    c.v = L[1].v; c.refrac = L[1].refrac; c.x = L[1].x; c.s = L[1].s; c.p = L[1].p.lif;
    c.ras = L[1].raster_s; c.rasV = L[1].raster_v;
    if (c.p.traces && !c.x) return SNN_OK;
    if (c.x_traces) {
        const long n = (long)c.B * img;
        hipLaunchKernelGGL(k_conv_xtrace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c);
    }
    // (the window-mask forms walk one input channel at a time; with several channels the reference's order interleaves
    //  them tap by tap, which the general form follows)
    if (c.Cin == 1 && c.KH == 5 && c.KW == 5) hipLaunchKernelGGL((k_convlif_run<5, 5>), dim3((unsigned)grid), dim3(NTC), lds, st, c);
    else if (c.Cin == 1 && c.KH == 3 && c.KW == 3) hipLaunchKernelGGL((k_convlif_run<3, 3>), dim3((unsigned)grid), dim3(NTC), lds, st, c);
    else hipLaunchKernelGGL((k_convlif_run<0, 0>), dim3((unsigned)grid), dim3(NTC), lds, st, c);
    const int rc = snn_check_launch();
    if (rc) return rc;
    snn_set_plan_name("convlif-fused");
    *handled = 1;
    return SNN_OK;
}


unsigned long long snn_convpp_workspace_bytes(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R) {
    ConvCtx c;
    if (!L || !C || !R || !convpp_match(L, nL, C, nC, R, c)) return 0;
    return convpp_workspace(c);
}

int snn_try_fused_convpp(const snn_layer_desc *L, int nL, const snn_conn_desc *C, int nC, const snn_run_desc *R, hipStream_t st, int *handled) {
    *handled = 0;
    ConvPPCtx a;
    memset(&a, 0, sizeof(a));
    ConvCtx &c = a.c;
    if (!convpp_match(L, nL, C, nC, R, c)) return SNN_OK;
    if (!R->workspace || R->workspace_bytes < convpp_workspace(c)) return SNN_OK;
    if (snn_prof_active()) return SNN_OK;                  // per-timestep event timing only exists for per-step plans
    static const bool off = [] { const char *v = getenv("SNN_CONVPP_FUSED"); return v && v[0] == '0'; }();
    if (off) return SNN_OK;
    const snn_conn_desc &d = C[0];
    const int taps = c.Cin * c.KH * c.KW, npix = c.OH * c.OW;
    c.in = L[0].ext_spikes; c.sX0 = L[0].s;
    c.xX = L[0].x; c.x_traces = 1; c.x_decay = L[0].p.lif.trace_decay; c.x_scale = L[0].p.lif.trace_scale; c.x_additive = L[0].p.lif.traces_additive;
    c.W = d.w; c.bias = d.bias;
    c.v = L[1].v; c.refrac = L[1].refrac; c.x = L[1].x; c.s = L[1].s; c.p = L[1].p.lif;
    c.ras = L[1].raster_s; c.rasV = L[1].raster_v;
    a.nu0 = d.nu0; a.nu1 = d.nu1; a.wdecay = d.wdecay; a.has_min = d.has_min; a.wmin = d.wmin; a.has_max = d.has_max; a.wmax = d.wmax;
    a.learning = R->learning ? 1 : 0;
    a.Wout = d.w;
    const size_t E = (size_t)c.Cout * taps;
    a.part = (float *)R->workspace;
    a.cnt = (unsigned *)((unsigned char *)R->workspace + (((size_t)4 * c.B * E * sizeof(float) + 255) & ~(size_t)255));
    a.status = R->status;
    static const bool want_timing = [] { const char *v = getenv("SNN_CONVPP_TIMING"); return v && v[0] == '1'; }();
    static long long *dbg_dev = nullptr;
    if (want_timing && !dbg_dev && hipMalloc(&dbg_dev, 8 * sizeof(long long)) != hipSuccess) dbg_dev = nullptr;
    a.dbg = want_timing ? dbg_dev : nullptr;
    const int nt = max((npix + 63) / 64 * 64, (CC * taps + 63) / 64 * 64);
    const size_t lds = convpp_lds(c);
    const long long grid = (long long)c.B * c.nchunk;
    void (*fn)(const ConvPPCtx) = (c.Cin == 1 && c.KH == 5 && c.KW == 5) ? k_convpp_run<5, 5> : ((c.Cin == 1 && c.KH == 3 && c.KW == 3) ? k_convpp_run<3, 3> : k_convpp_run<0, 0>);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return SNN_OK; }
    // every workgroup of the grid has to be resident at once (they wait for each other once per step): ask the occupancy calculator, and
    // launch cooperatively so that the runtime refuses what it cannot place
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, nt, lds) != hipSuccess) { (void)hipGetLastError(); return SNN_OK; }
    if (per_cu < 1 || grid > (long long)per_cu * prop.multiProcessorCount) return SNN_OK;
    if (a.learning && hipMemsetAsync(a.cnt, 0, (size_t)c.nchunk * sizeof(unsigned), st) != hipSuccess) return SNN_ERR_LAUNCH;
    void *args[] = {(void *)&a};
    const hipError_t e = hipLaunchCooperativeKernel((const void *)fn, dim3((unsigned)grid), dim3((unsigned)nt), args, (unsigned)lds, st);
    if (e != hipSuccess) { (void)hipGetLastError(); return SNN_OK; }         // refused: the generic plan runs
    const int rc = snn_check_launch();
    if (rc) return rc;
    if (a.dbg) {                                           // developer aid: synchronous
        long long h[8];
        if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[convpp, us per step, workgroup 0] conv + LIF %.2f | input trace + row packing %.2f | partial sums %.2f | release %.2f | wait for the chunk %.2f | batch reduction + apply %.2f\n",
                    h[0] / 100.0 / c.T, h[1] / 100.0 / c.T, h[2] / 100.0 / c.T, h[3] / 100.0 / c.T, h[4] / 100.0 / c.T, h[5] / 100.0 / c.T);
    }
    snn_set_plan_name("convpp-fused");
    *handled = 1;
    return SNN_OK;
}
