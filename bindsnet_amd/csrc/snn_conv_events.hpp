// snn_conv_events.hpp -- event-driven partial sums of PostPre on a Conv2dConnection (bindsnet/learning/learning.py:457-497).
//
// k_conv_pp_partial (snn_ops.hip) gives every (sample, weight element) thread a serial loop over ALL OH*OW output positions:
// two loads and two multiply-adds per position, 576 positions at the conv_mnist.py shapes, while ~5 % of the source pixels
// and ~1 % of the target neurons carry a spike.  Here the spikes of a sample are first packed into one 32-bit word per
// image row (rows of up to 32 pixels), and an element walks only the SET bits: for the pre-synaptic sum the source spikes
// of the input row a tap reads, for the post-synaptic sum the target spikes of the output row -- in ascending output
// position, i.e. in exactly the order of the dense loop, whose skipped terms are +-0 (0/1 spikes, finite traces): the
// partial sums are bit-identical.  Rows wider than 32 pixels or multi-valued spike bytes take the dense body.
//
// The two bodies are __host__ __device__ so that tests/hostcheck/conv_events_host.hip can run them on the CPU, element by
// element, against each other (the build container has no GPU): tests/test_conv_events_host.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snn {

struct ConvGeom { int Cin, H, Wd, Cout, KH, KW, stride, pad, OH, OW; };

// bit x of the result = (row[x] != 0) for x < n <= 32; *multi is set when a byte is neither 0 nor 1
__host__ __device__ inline uint32_t conv_pack_row(const uint8_t *row, int n, int *multi) {
    uint32_t m = 0;
    for (int x = 0; x < n; ++x) {
        const uint8_t v = row[x];
        m |= (uint32_t)(v != 0) << x;
        if (v > 1) *multi = 1;
    }
    return m;
}

// The dense body: what k_conv_pp_partial computes for weight element (co, ci, ky, kx) of one sample.
//   a = sum_l x_tgt[co,l] * s_src[ci,iy,ix]     p = sum_l s_tgt[co,l] * x_src[ci,iy,ix]     l = oy*OW + ox ascending,
//   iy = oy*stride - pad + ky, ix = ox*stride - pad + kx, out-of-image taps contribute 0.
__host__ __device__ inline void conv_pp_dense(const ConvGeom &g, int ky, int kx, const uint8_t *s_src_c, const float *x_src_c,
                                              const uint8_t *s_tgt_c, const float *x_tgt_c, float *a_out, float *p_out) {
    float a = 0.f, p = 0.f;
    const int L = g.OH * g.OW;
    for (int l = 0; l < L; ++l) {
        const int oy = l / g.OW, ox = l - oy * g.OW;
        const int iy = oy * g.stride - g.pad + ky, ix = ox * g.stride - g.pad + kx;
        const bool in = iy >= 0 && iy < g.H && ix >= 0 && ix < g.Wd;
        const int si = (in ? iy : 0) * g.Wd + (in ? ix : 0);
        a += x_tgt_c[l] * (in ? (float)s_src_c[si] : 0.0f);
        p += (float)s_tgt_c[l] * (in ? x_src_c[si] : 0.0f);
    }
    *a_out = a;
    *p_out = p;
}

// The event-driven body: srow[H] = packed source rows of (sample, ci), trow[OH] = packed target rows of (sample, co).
__host__ __device__ inline void conv_pp_events(const ConvGeom &g, int ky, int kx, const uint32_t *srow, const uint32_t *trow,
                                               const float *x_src_c, const float *x_tgt_c, float *a_out, float *p_out) {
    float a = 0.f, p = 0.f;
    if (g.stride == 1) {
        // stride 1 (what conv_mnist.py builds): ox = ix + pad - kx, so the source spikes a tap can see are a contiguous bit range of the
        // row's word -- mask it once, then every set bit is a term (no division, no per-event range tests); the same terms in the same order
        const int sh = kx - g.pad;                         // ix = ox + sh
        for (int oy = 0; oy < g.OH; ++oy) {
            const int iy = oy - g.pad + ky;
            if (iy < 0 || iy >= g.H) continue;
            uint32_t m = srow[iy];
            // keep ix with 0 <= ix - sh < OW
            const int lo = sh > 0 ? sh : 0, hi = g.OW + sh < 32 ? g.OW + sh : 32;      // ix in [lo, hi)
            m = hi <= lo ? 0u : (m >> lo << lo) & (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u));
            const float *xt = x_tgt_c + oy * g.OW - sh;
            while (m) {
                const int ix = __builtin_ctz(m);
                m &= m - 1;
                a += xt[ix];                               // x_tgt[oy, ix - sh] * 1.0f
            }
            uint32_t q = trow[oy];
            // keep ox with 0 <= ox + sh < Wd
            const int qlo = sh < 0 ? -sh : 0, qhi = g.Wd - sh < 32 ? g.Wd - sh : 32;
            q = qhi <= qlo ? 0u : (q >> qlo << qlo) & (qhi >= 32 ? 0xFFFFFFFFu : ((1u << qhi) - 1u));
            const float *xs = x_src_c + iy * g.Wd + sh;
            while (q) {
                const int ox = __builtin_ctz(q);
                q &= q - 1;
                p += xs[ox];                               // 1.0f * x_src[iy, ox + sh]
            }
        }
        *a_out = a;
        *p_out = p;
        return;
    }
    for (int oy = 0; oy < g.OH; ++oy) {
        const int iy = oy * g.stride - g.pad + ky;
        if (iy < 0 || iy >= g.H) continue;                 // every tap of this output row lies outside the image
        uint32_t m = srow[iy];                             // source spikes of input row iy, ascending ix <=> ascending ox
        while (m) {
            const int ix = __builtin_ctz(m);
            m &= m - 1;
            const int t = ix + g.pad - kx;                 // = ox * stride
            if (t < 0) continue;
            const int ox = t / g.stride;
            if (ox * g.stride != t || ox >= g.OW) continue;
            a += x_tgt_c[oy * g.OW + ox];                  // x_tgt * 1.0f
        }
        uint32_t q = trow[oy];                             // target spikes of output row oy
        while (q) {
            const int ox = __builtin_ctz(q);
            q &= q - 1;
            const int ix = ox * g.stride - g.pad + kx;
            if (ix < 0 || ix >= g.Wd) continue;
            p += x_src_c[iy * g.Wd + ix];                  // 1.0f * x_src
        }
    }
    *a_out = a;
    *p_out = p;
}

}  // namespace snn
