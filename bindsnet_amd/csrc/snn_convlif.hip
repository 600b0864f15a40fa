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
#include "../../include/snnhip.h"
#include "snn_common.hpp"

using namespace snn;

void snn_set_plan_name(const char *name);

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
