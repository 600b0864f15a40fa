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
//   workgroup  <->  (sample b, chunk of CCP output channels), ALL output pixels: the partial sums of a weight element run over the output
//                   positions in ascending order, one chain per (sample, element) -- a workgroup that holds every position of its channels
//                   has that chain to itself.  CCP = 8 / 4 / 2 is chosen so that the grid fills the chip: the LIF step is vector-issue work
//                   (a first cut with ONE workgroup per channel and all samples -- no exchange at all -- spent 16 us per step there)
//   thread     <->  one output pixel; v / refrac / trace of its CCP neurons in registers for all T steps
//   per step:  (1) the pixel's window from the PACKED rows of the image (a word per image row: KH reads instead of KH * KW byte reads), the
//                  set taps in ascending order with the chunk's CURRENT filters (LDS), LIF step; the new target spikes and traces go to LDS,
//                  the step's input image (= the "after" spikes PostPre pairs with) is staged meanwhile;
//              (2) the sample's input trace (LDS), packed rows of the step's image and of the new target spikes;
//              (3) the two partial sums of every element of the chunk over the positions: stride 1 and 0/1 spikes -- the step's spikes as
//                  ascending event lists (one per input channel, one per own output channel), a thread per SUM walking its list eight events
//                  at a time (window test, eight loads in flight, the additions in order); otherwise snn_conv_events.hpp's bodies (what
//                  k_conv_pp_partial_ev runs) -- published as TAGGED 8-byte granules {step + 1, value} in the workspace, [parity][a | p][B][E];
//              (4) every workgroup polls the granules of ITS chunk from all B samples (one hop: no release / acquire fences, which write
//                  back and invalidate whole caches: 0.8 + 3.3 + ~3 us per step in the first cut), reduces them in ATen's sum(dim=0) order
//                  and applies rates, decay and clamp to its own copy of the filters (k_conv_pp_apply's statements) -- B identical copies
//                  instead of a second hand-off to spread the result.
// Bit-identical to the generic plan: the same terms in the same order.  The exchange area has two halves by step parity: a workgroup may be
// publishing step t+1 while a slower one still polls step t; to publish step t+2 it has to get through the poll of step t+1, which needs
// that slower one's publish of t+1, i.e. its poll of step t finished.  The area is zeroed before every launch (tags are never 0).
struct ConvPPCtx {
    ConvCtx c;
    float nu0, nu1, wdecay, wmin, wmax;
    int has_min, has_max, learning;
    float *Wout;                  // the connection's weights (written back by the b == 0 workgroups)
    unsigned long long *gr;       // [2][2][B][E] granules
    unsigned *fin;                // [2] behind the granules: workgroups through with the run, "somebody gave up"; zeroed before the launch
    int stall_wg;                 // test hook (SNN_CONVPP_TEST_STALL): this workgroup returns at once
    int *status;                  // nullable
    long long *dbg;               // developer aid (SNN_CONVPP_TIMING=1): [8] phase totals of workgroup 0 in 10 ns ticks
};

__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void granule_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr int kPPPoll = 2000000;  // polls of a granule before a workgroup gives up (a co-resident grid publishes within microseconds)

template <int KH_, int KW_, int CCP>
__global__ __launch_bounds__(1024) void k_convpp_run(const ConvPPCtx a) {
    const ConvCtx &c = a.c;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NT = (int)blockDim.x;
    const int KK = c.KH * c.KW, taps = c.Cin * KK, img = c.Cin * c.H * c.Wd, L = c.OH * c.OW, B = c.B;
    const int imgw = (img + 3) / 4, nrS = c.Cin * c.H;
    float *wl = (float *)smem;                            // [CCP][taps] this chunk's filters: the workgroup's own copy for the whole run
    float *red = wl + CCP * taps;                         // [2][B][CCP * taps] the chunk's partial sums of all samples
    float *xs = red + 2 * B * CCP * taps;                 // [img] the sample's input trace
    float *xt = xs + img;                                 // [CCP][L] target traces after this step
    uint32_t *im = (uint32_t *)(xt + CCP * L);            // [2][imgw] input images (bytes), by step parity
    uint32_t *srow = im + 2 * imgw;                       // [2][Cin * H] packed source rows, by step parity
    uint32_t *trow = srow + 2 * nrS;                      // [CCP * OH] packed target rows
    int *multi = (int *)(trow + CCP * c.OH);              // [4]: [p] the image of parity p has a byte > 1; [2] give up
    uint8_t *sbt = (uint8_t *)(multi + 4);                // [CCP][L] target spikes of this step (bytes)
    int *nev = (int *)(sbt + ((CCP * L + 3) & ~3));       // [Cin + CCP] events per list
    uint16_t *evS = (uint16_t *)(nev + ((c.Cin + CCP + 1) & ~1));   // [Cin][H * Wd + 8] source spikes of the step's image as (iy << 8 | ix), ascending, padded
    uint16_t *evT = evS + c.Cin * (c.H * c.Wd + 8);       // [CCP][L + 8] target spikes of the step as (oy << 8 | ox), ascending, padded with 0xFFFF to a multiple of 8
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == a.stall_wg) return;
    const int chunk = blockIdx.x % c.nchunk, b = blockIdx.x / c.nchunk;
    const int c0 = chunk * CCP, nco = min(CCP, c.Cout - c0), nel = nco * taps;
    const int pix = tid;
    const bool valid = pix < L;
    const int oy = valid ? pix / c.OW : 0, ox = valid ? pix - oy * c.OW : 0;
    const long K = taps, E = (long)c.Cout * K;
    const ConvGeom g{c.Cin, c.H, c.Wd, c.Cout, c.KH, c.KW, c.stride, c.pad, c.OH, c.OW};
    for (int k = tid; k < CCP * taps; k += NT) { const int cc = k / taps; wl[k] = (c0 + cc < c.Cout) ? c.W[(size_t)(c0 + cc) * taps + (k - cc * taps)] : 0.f; }
    for (int k = tid; k < img; k += NT) xs[k] = c.xX[(size_t)b * img + k];
    if (tid < 4) multi[tid] = 0;
    float v[CCP], rf[CCP], xtr[CCP], bs[CCP]; bool last[CCP];
    const size_t nB = (size_t)c.Cout * L;
#pragma unroll
    for (int u = 0; u < CCP; ++u) {
        v[u] = rf[u] = xtr[u] = 0.f; last[u] = false;
        bs[u] = (c.bias && c0 + u < c.Cout) ? c.bias[c0 + u] : 0.f;
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * L + pix;
            v[u] = c.v[k]; rf[u] = c.refrac[k]; last[u] = c.s[k] != 0;
            xtr[u] = c.x[k];
        }
    }
    // bit x of the result = (row[x] != 0); flag is set when a byte is neither 0 nor 1 (k_conv_pp_partial_ev's pack)
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
    {   // the image at entry (the input layer's spikes before the run) and its packed rows
        const uint8_t *src = c.sX0 + (size_t)b * img;
        for (int k = tid; k < imgw; k += NT) {
            uint32_t w = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (4 * k + q < img) w |= (uint32_t)src[4 * k + q] << (8 * q);
            im[k] = w;
        }
        __syncthreads();
        int flag = 0;
        for (int r = tid; r < nrS; r += NT) srow[r] = pack((const uint8_t *)im + (size_t)r * c.Wd, c.Wd, flag);
        if (flag) atomicOr(&multi[0], 1);
    }
    __syncthreads();
    bool dead = false;
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
    const bool timing = a.dbg && blockIdx.x == 0 && tid == 0;
#define PPMARK(k) do { if (timing) { const long long now_ = (long long)wall_clock64(); ph[k] += now_ - tk; tk = now_; } } while (0)
    if (timing) tk = (long long)wall_clock64();
    for (int t = 0; t < c.T; ++t) {
        const int par = t & 1;
        const uint8_t *cur = (const uint8_t *)(im + par * imgw);
        uint32_t *nxt = im + (par ^ 1) * imgw;
        const uint32_t *srowc = srow + par * nrS;
        uint32_t *srown = srow + (par ^ 1) * nrS;
        const bool mcur = multi[par] != 0;
        // (the tables the packing pass below ORs into: last read in front of the previous step's barriers)
        const bool wordform = (c.Wd & 3) == 0 && (c.OW & 3) == 0;
        if (wordform) {
            for (int r = tid; r < nrS; r += NT) srown[r] = 0u;
            for (int r = tid; r < nco * c.OH; r += NT) trow[r] = 0u;
        }
        if (tid == 0) multi[par ^ 1] = 0;
        // this step's input image: the convolution of the NEXT step reads it, PostPre of THIS step pairs with it
        uint32_t pre[4] = {0u, 0u, 0u, 0u};                // imgw <= 4 * NT (host check)
        {
            const uint8_t *src = c.in + ((size_t)t * B + b) * img;
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
            float acc[CCP];
#pragma unroll
            for (int u = 0; u < CCP; ++u) acc[u] = 0.f;
            const int y0 = oy * c.stride - c.pad, x0 = ox * c.stride - c.pad;
            if constexpr (KH_ > 0) {                       // one input channel, KH_ * KW_ <= 32: the window as a bit mask from KH_ row words
                uint32_t m = 0;
#pragma unroll
                for (int ky = 0; ky < KH_; ++ky) {
                    const int iy = y0 + ky;
                    const bool inr = iy >= 0 && iy < c.H;
                    const uint32_t w = srowc[inr ? iy : 0];
                    const uint32_t bits = inr ? ((x0 >= 0 ? (w >> x0) : (w << (-x0))) & ((1u << KW_) - 1u)) : 0u;
                    m |= bits << (ky * KW_);
                }
                while (m) {                                // ascending taps = the generic kernel's order; a zero tap adds nothing
                    const int k = __ffs(m) - 1; m &= m - 1;
                    const float fs = mcur ? (float)cur[(y0 + k / KW_) * c.Wd + x0 + k % KW_] : 1.0f;
#pragma unroll
                    for (int u = 0; u < CCP; ++u) acc[u] += fs * wl[u * taps + k];
                }
            } else {
                for (int ky = 0; ky < c.KH; ++ky) {       // the generic kernel's order: taps row-major, input channels innermost
                    const int iy = y0 + ky;
                    for (int kx = 0; kx < c.KW; ++kx)
                        for (int ci = 0; ci < c.Cin; ++ci) {
                            const int tap = (ci * c.KH + ky) * c.KW + kx;
                            const int ix = x0 + kx;
                            if (iy < 0 || iy >= c.H || ix < 0 || ix >= c.Wd) continue;
                            const uint8_t sv = cur[(ci * c.H + iy) * c.Wd + ix];
                            if (!sv) continue;
                            const float fs = (float)sv;
#pragma unroll
                            for (int u = 0; u < CCP; ++u) acc[u] += fs * wl[u * taps + tap];
                        }
                }
            }
#pragma unroll
            for (int u = 0; u < CCP; ++u) {
                if (c0 + u >= c.Cout) continue;
                float r = acc[u];
                if (c.bias) r = r + bs[u];
                float cur_in = 0.0f + r;                   // zeros + conv (network.py:225-248)
                if (rf[u] > 0.f) cur_in = 0.f;             // nodes.py:511
                const bool sp = lif_update(v[u], rf[u], cur_in, c.p);
                last[u] = sp;
                xtr[u] = trace_next(xtr[u], sp, c.p.trace_decay, c.p.trace_scale, c.p.traces_additive);
                const size_t k = ((size_t)t * B + b) * nB + (size_t)(c0 + u) * L + pix;
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
            int flag = 0, flagt = 0;
            if (wordform) {
                // a thread per 4-byte word of the image: its four traces, its four bits of the row's word (rows are whole words: Wd % 4 == 0)
                for (int k = tid; k < (img >> 2); k += NT) {
                    const uint32_t w = nxt[k];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xs[4 * k + q] = trace_next(xs[4 * k + q], (uint8_t)(w >> (8 * q)), c.x_decay, c.x_scale, c.x_additive);
                    if (w & 0xFEFEFEFEu) flag = 1;
                    const uint32_t nz = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
                    const uint32_t nib = ((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xFu;
                    if (nib) { const int r = (4 * k) / c.Wd, x = 4 * k - r * c.Wd; atomicOr(&srown[r], nib << x); }
                }
                if (a.learning) {
                    const uint32_t *sw = (const uint32_t *)sbt;
                    for (int k = tid; k < ((nco * L) >> 2); k += NT) {
                        const uint32_t w = sw[k];
                        if (!w) continue;
                        const uint32_t nz = (w | ((w & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
                        const uint32_t nib = ((nz >> 7) | (nz >> 14) | (nz >> 21) | (nz >> 28)) & 0xFu;
                        const int r = (4 * k) / c.OW, x = 4 * k - r * c.OW;          // (L = OH * OW: r = cl * OH + oy)
                        atomicOr(&trow[r], nib << x);
                    }
                }
            } else {
                for (int k = tid; k < img; k += NT) xs[k] = trace_next(xs[k], nb[k], c.x_decay, c.x_scale, c.x_additive);
                for (int r = tid; r < nrS; r += NT) srown[r] = pack(nb + (size_t)r * c.Wd, c.Wd, flag);
                if (a.learning) for (int r = tid; r < nco * c.OH; r += NT) { const int cl = r / c.OH, y = r - cl * c.OH; trow[r] = pack(sbt + cl * L + y * c.OW, c.OW, flagt); }
            }
            if (flag) atomicOr(&multi[par ^ 1], 1);        // (zeroed at the top of the step, in front of the barrier behind the convolution)
        }
        __syncthreads();
        PPMARK(1);
        if (!a.learning) continue;                         // (no rule: the convolution of the next step is all that reads this step's results)
        // ---- the partial sums of the chunk's elements for this sample, published as tagged granules
        unsigned long long *ga = a.gr + (size_t)par * 2 * B * E, *gp = ga + (size_t)B * E;
        const unsigned long long tag = (unsigned long long)(uint32_t)(t + 1) << 32;
        const bool fast = c.stride == 1 && multi[par ^ 1] == 0;
        if (fast) {
            // the step's spikes as ascending lists -- one per input channel from the packed source rows, one per own output channel from the
            // packed target rows (a wave per list: row counts, prefix over the lanes, every row writes its events).  Every element's sum walks
            // the SAME list and keeps the events inside its window: (iy, ix) ascending is (oy, ox) ascending for each of them.
            {
                const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
                for (int li = wave; li < c.Cin + nco; li += nwaves) {
                    const bool sl = li < c.Cin;
                    const uint32_t *rows = sl ? srown + li * c.H : trow + (li - c.Cin) * c.OH;
                    uint16_t *dst = sl ? evS + li * (c.H * c.Wd + 8) : evT + (li - c.Cin) * (L + 8);
                    const int nr = sl ? c.H : c.OH;
                    int off = 0;
                    for (int r0 = 0; r0 < nr; r0 += 64) {
                        uint32_t word = (r0 + lane < nr) ? rows[r0 + lane] : 0u;
                        const int cnt = __popc(word);
                        int incl = cnt;
#pragma unroll
                        for (int dlt = 1; dlt < 64; dlt <<= 1) { const int o = __shfl_up(incl, dlt); if (lane >= dlt) incl += o; }
                        int pos = off + incl - cnt;
                        while (word) { const int x = __ffs(word) - 1; word &= word - 1; dst[pos++] = (uint16_t)(((r0 + lane) << 8) | x); }
                        off += __shfl(incl, 63);
                    }
                    if (lane < 8) dst[off + lane] = 0xFFFFu;          // (row 255: outside every window -- the sums walk whole groups of eight)
                    if (lane == 0) nev[li] = off;
                }
            }
            __syncthreads();
            for (int id = tid; id < 2 * nel; id += NT) {
                const bool src = id < nel;
                const int e = src ? id : id - nel, cl = e / taps, kq = e - cl * taps, ci = kq / KK, kk = kq - ci * KK, ky = kk / c.KW, kx = kk - ky * c.KW;
                const int dy = ky - c.pad, sh = kx - c.pad;
                // src: events (iy, ix) of input channel ci; the term sits at x_tgt[cl][(iy - dy) * OW + ix - sh] when (iy - dy, ix - sh) is an output position
                // tgt: events (oy, ox) of output channel cl; the term sits at x_src[ci][(oy + dy) * Wd + ox + sh] when (oy + dy, ox + sh) is an input position
                const uint16_t *ev = src ? evS + ci * (c.H * c.Wd + 8) : evT + cl * (L + 8);
                const int n = nev[src ? ci : c.Cin + cl];
                const float *vals = src ? xt + cl * L : xs + ci * c.H * c.Wd;
                const int ddy = src ? -dy : dy, ddx = src ? -sh : sh, rowlen = src ? c.OW : c.Wd, nrow = src ? c.OH : c.H;
                const float acc = conv_pp_list_sum(ev, n, vals, ddy, ddx, rowlen, nrow);      // (snn_conv_events.hpp: also run on the host, tests/test_conv_events_host.py)
                granule_store((src ? ga : gp) + (size_t)b * E + (size_t)(c0 + cl) * K + kq, tag | (unsigned long long)__float_as_uint(acc));
            }
        } else {
            const uint8_t *nb = (const uint8_t *)nxt;
            for (int e = tid; e < nel; e += NT) {
                const int cl = e / taps, kq = e - cl * taps, ci = kq / KK, kk = kq - ci * KK, ky = kk / c.KW, kx = kk - ky * c.KW;
                float sa, sp_;
                if (multi[par ^ 1] == 0) conv_pp_events(g, ky, kx, srown + ci * c.H, trow + cl * c.OH, xs + ci * c.H * c.Wd, xt + cl * L, &sa, &sp_);
                else conv_pp_dense(g, ky, kx, nb + ci * c.H * c.Wd, xs + ci * c.H * c.Wd, sbt + cl * L, xt + cl * L, &sa, &sp_);
                const size_t id = (size_t)b * E + (size_t)(c0 + cl) * K + kq;
                granule_store(ga + id, tag | (unsigned long long)__float_as_uint(sa));
                granule_store(gp + id, tag | (unsigned long long)__float_as_uint(sp_));
            }
        }
        PPMARK(2);
        // ---- the chunk's partial sums of ALL samples: poll the granules (own ones included), values -> LDS
        {
            const int want = 2 * B * nel;
            for (int id0 = 0; id0 < want; id0 += 4 * NT) {
                unsigned long long x[4];
                const unsigned long long *src[4];
                uint32_t need = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int id = id0 + tid + u * NT;
                    src[u] = ga;
                    if (id < want) {
                        const int which = id / (B * nel), r0 = id - which * (B * nel), bb = r0 / nel, e = r0 - bb * nel;
                        src[u] = (which ? gp : ga) + (size_t)bb * E + (size_t)c0 * K + e;      // (the chunk's elements are contiguous: (c0 + cl) * K + kq = c0 * K + e)
                        need |= 1u << u;
                    }
                }
                for (int spins = 0; need; ++spins) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) if ((need >> u) & 1u) x[u] = granule_load(src[u]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (((need >> u) & 1u) && (x[u] >> 32) == (tag >> 32)) {
                        need &= ~(1u << u);
                        red[id0 + tid + u * NT] = __uint_as_float((uint32_t)x[u]);             // red[which][bb][e]: the same flat index
                    }
                    if (need && spins > kPPPoll) { multi[2] = 1; break; }
                    if (need && (spins & 7) == 7) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        __syncthreads();
        PPMARK(3);
        if (multi[2]) { dead = true; break; }
        // ---- batch reduction in ATen's sum(dim=0) order + apply (k_conv_pp_apply's statements) on the own copy of the filters
        if (tid < nel) {
            const long e = (long)c0 * K + tid;
            const bool tail = e >= (E / 32) * 32;
            float w = wl[tid];
            auto ordered = [&](const float *base) {           // (sixteen samples' values are read together, then added in ATen's order)
                if (tail) {
                    OuterSum accs; accs.init(true);
                    for (int bb = 0; bb < B; ++bb) accs.add(bb, base[bb * nel + tid], B);
                    return accs.finish(B);
                }
                // OuterSum's cascade (snn_order.hpp Cascade) over the dense positions 0 .. B-1: the sixteen positions of a group share their
                // block (b0 >> 4 <= B >> 4), so the block test runs once per group instead of once per term
                Cascade cs; cs.init();
                for (int b0 = 0; b0 < B; b0 += 16) {
                    float vv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) vv[u] = base[min(b0 + u, B - 1) * nel + tid];
                    if ((b0 >> 4) != cs.cb) cs.advance(b0 >> 4);
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (b0 + u < B) cs.a0 += vv[u];
                }
                return cs.finish(B >> 4);
            };
            if (a.nu0 != 0.f) w = w - a.nu0 * ordered(red);
            if (a.nu1 != 0.f) w = w + a.nu1 * ordered(red + B * nel);
            w = w * a.wdecay;
            if (a.has_min && w < a.wmin) w = a.wmin;
            if (a.has_max && w > a.wmax) w = a.wmax;
            wl[tid] = w;
        }
        __syncthreads();                                   // the filters of step t+1 are in place; red and the row tables are free
        PPMARK(4);
    }
    if (timing) for (int k = 0; k < 8; ++k) a.dbg[k] = ph[k];
    // ---- COMMIT: nobody writes a state tensor, the filters or the input trace back unless EVERY workgroup of the grid got through its T steps (the
    //      chunks do not wait for each other during the run: without this a chunk that gave up would leave the others' state advanced).  One
    //      counter hop per run; a workgroup that gave up says so and still counts, so that nobody waits for it.
    if (tid == 0) {
        if (dead) __hip_atomic_store(&a.fin[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.fin[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&a.fin[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x &&
               !__hip_atomic_load(&a.fin[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (++spins > kPPPoll) { __hip_atomic_store(&a.fin[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(4);
        }
        multi[3] = (int)__hip_atomic_load(&a.fin[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (multi[3]) { if (tid == 0 && a.status) atomicCAS(a.status, 0, SNN_ERR_TIMEOUT); return; }
#pragma unroll
    for (int u = 0; u < CCP; ++u)
        if (valid && c0 + u < c.Cout) {
            const size_t k = (size_t)b * nB + (size_t)(c0 + u) * L + pix;
            c.v[k] = v[u]; c.refrac[k] = rf[u]; c.s[k] = last[u];
            c.x[k] = xtr[u];
        }
    if (a.learning && b == 0) for (int k = tid; k < nel; k += NT) a.Wout[(size_t)c0 * taps + k] = wl[k];
    if (chunk == 0) for (int k = tid; k < img; k += NT) c.xX[(size_t)b * img + k] = xs[k];
}

size_t convpp_lds(const ConvCtx &c, int ccp) {
    const size_t taps = (size_t)c.Cin * c.KH * c.KW, img = (size_t)c.Cin * c.H * c.Wd, L = (size_t)c.OH * c.OW;
    return (ccp * taps + 2 * c.B * ccp * taps + img + ccp * L) * 4 + 2 * ((img + 3) / 4) * 4 + (2 * c.Cin * c.H + ccp * c.OH + 4) * 4 + ((ccp * L + 3) & ~(size_t)3) +
           ((c.Cin + ccp + 1) & ~(size_t)1) * 4 + (img + 8 * c.Cin + ccp * (L + 8)) * 2 + 32;
}
int convpp_threads(const ConvCtx &c, int ccp) {
    const int taps = c.Cin * c.KH * c.KW, npix = c.OH * c.OW;
    int nt = (npix + 63) / 64 * 64;
    if (nt < 256) nt = 256;
    (void)taps; (void)ccp;
    return nt;
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
    const int img = c.Cin * c.H * c.Wd, npix = c.OH * c.OW;
    if (L[0].n != img || L[1].n != c.Cout * npix) return false;
    if (npix > 1024 || max(c.H, c.OH) + max(c.KH, c.pad) > 255) return false;    // a thread per output pixel; event lists hold (row << 8 | column), row 255 = padding
    if ((img + 3) / 4 > 4 * convpp_threads(c, 2)) return false;
    if (convpp_lds(c, 2) > 150 * 1024) return false;
    if ((double)R->T * R->B * L[1].n >= 9.0e15) return false;
    if (!L[0].x || !L[1].x || !L[0].p.lif.traces || !L[1].p.lif.traces) return false;   // PostPre reads both traces
    return true;
}
size_t convpp_workspace(const ConvCtx &c) {
    const size_t E = (size_t)c.Cout * c.Cin * c.KH * c.KW;
    return (size_t)4 * c.B * E * sizeof(unsigned long long) + 256;
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

template <int CCP>
static void (*convpp_kernel(const ConvCtx &c))(const ConvPPCtx) {
    if (c.Cin == 1 && c.KH == 5 && c.KW == 5) return k_convpp_run<5, 5, CCP>;
    if (c.Cin == 1 && c.KH == 3 && c.KW == 3) return k_convpp_run<3, 3, CCP>;
    return k_convpp_run<0, 0, CCP>;
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
    const int taps = c.Cin * c.KH * c.KW;
    c.in = L[0].ext_spikes; c.sX0 = L[0].s;
    c.xX = L[0].x; c.x_traces = 1; c.x_decay = L[0].p.lif.trace_decay; c.x_scale = L[0].p.lif.trace_scale; c.x_additive = L[0].p.lif.traces_additive;
    c.W = d.w; c.bias = d.bias;
    c.v = L[1].v; c.refrac = L[1].refrac; c.x = L[1].x; c.s = L[1].s; c.p = L[1].p.lif;
    c.ras = L[1].raster_s; c.rasV = L[1].raster_v;
    a.nu0 = d.nu0; a.nu1 = d.nu1; a.wdecay = d.wdecay; a.has_min = d.has_min; a.wmin = d.wmin; a.has_max = d.has_max; a.wmax = d.wmax;
    a.learning = R->learning ? 1 : 0;
    a.Wout = d.w;
    const size_t E = (size_t)c.Cout * taps;
    a.gr = (unsigned long long *)R->workspace;
    a.fin = (unsigned *)((unsigned char *)R->workspace + (size_t)4 * c.B * E * sizeof(unsigned long long));
    a.stall_wg = getenv("SNN_CONVPP_TEST_STALL") ? atoi(getenv("SNN_CONVPP_TEST_STALL")) : -1;
    a.status = R->status;
    static const bool want_timing = [] { const char *v = getenv("SNN_CONVPP_TIMING"); return v && v[0] == '1'; }();
    static long long *dbg_dev = nullptr;
    if (want_timing && !dbg_dev && hipMalloc(&dbg_dev, 8 * sizeof(long long)) != hipSuccess) dbg_dev = nullptr;
    a.dbg = want_timing ? dbg_dev : nullptr;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return SNN_OK; }
    // channels per workgroup: the smallest chunk whose grid of B * ceil(Cout / CCP) workgroups is resident at once (they poll each other every
    // step) -- the LIF step is vector-issue work, so more, smaller workgroups win as long as they fit.  SNN_CONVPP_CC forces one.
    const int force_cc = getenv("SNN_CONVPP_CC") ? atoi(getenv("SNN_CONVPP_CC")) : 0;          // (test / measurement switch, read per run)
    const int tries[3] = {2, 4, 8};
    for (int ti = 0; ti < 3; ++ti) {
        const int ccp = tries[ti];
        if (force_cc && ccp != force_cc) continue;
        void (*fn)(const ConvPPCtx) = ccp == 2 ? convpp_kernel<2>(c) : (ccp == 4 ? convpp_kernel<4>(c) : convpp_kernel<8>(c));
        const int nt = convpp_threads(c, ccp);
        const size_t lds = convpp_lds(c, ccp);
        if (lds > 150 * 1024 || (taps * ccp > 32 * 1024)) continue;
        c.nchunk = (c.Cout + ccp - 1) / ccp; c.ntile = 1;
        const long long grid = (long long)c.B * c.nchunk;
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); continue; }
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, nt, lds) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (per_cu < 1 || grid > (long long)per_cu * prop.multiProcessorCount) continue;
        if (hipMemsetAsync(a.learning ? (void *)a.gr : (void *)a.fin, 0, (a.learning ? (size_t)4 * c.B * E * sizeof(unsigned long long) : 0) + 16, st) != hipSuccess) return SNN_ERR_LAUNCH;
        void *args[] = {(void *)&a};
        const hipError_t e = hipLaunchCooperativeKernel((const void *)fn, dim3((unsigned)grid), dim3((unsigned)nt), args, (unsigned)lds, st);
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }          // refused: a coarser chunk, then the generic plan
        const int rc = snn_check_launch();
        if (rc) return rc;
        if (a.dbg) {                                       // developer aid: synchronous
            long long h[8];
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h, a.dbg, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                fprintf(stderr, "[convpp, %d channels per workgroup, grid %lld x %d; us per step, workgroup 0] conv + LIF %.2f | input trace + row packing %.2f | partial sums + publish %.2f | poll %.2f | batch reduction + apply %.2f\n",
                        ccp, grid, nt, h[0] / 100.0 / c.T, h[1] / 100.0 / c.T, h[2] / 100.0 / c.T, h[3] / 100.0 / c.T, h[4] / 100.0 / c.T);
        }
        snn_set_plan_name("convpp-fused");
        *handled = 1;
        return SNN_OK;
    }
    return SNN_OK;
}
