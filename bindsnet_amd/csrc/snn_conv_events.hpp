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

// ---- the list form (round 6, k_convpp_run in snn_convlif.hip; stride 1, 0/1 spikes).  The spikes of one image (a sample's input channel, or one
// of its output channels) as an ASCENDING list of (row << 8 | column), padded with eight entries 0xFFFF (row 255: outside every window, so a sum
// can walk whole groups of eight).  Every weight element's sum walks the SAME list and keeps the events inside its window: (iy, ix) ascending is
// (oy, ox) ascending for each of them, i.e. the order of conv_pp_events / conv_pp_dense.  Serial form of the build (the kernel has a wave do it:
// row counts, prefix over the lanes); dst holds up to rows * 32 + 8 entries.  Returns the number of events.
__host__ __device__ inline int conv_event_list(const uint32_t *rows, int nr, uint16_t *dst) {
    int n = 0;
    for (int r = 0; r < nr; ++r) {
        uint32_t m = rows[r];
        while (m) { const int x = __builtin_ctz(m); m &= m - 1; dst[n++] = (uint16_t)((r << 8) | x); }
    }
    for (int u = 0; u < 8; ++u) dst[n + u] = 0xFFFFu;
    return n;
}

// One partial sum from a padded list: event (y, x) contributes vals[(y + ddy) * rowlen + x + ddx] when (y + ddy, x + ddx) lies inside
// [0, nrow) x [0, rowlen), in list order.  Pre-synaptic sum of element (ky, kx): the SOURCE list of the input channel, vals = x_tgt of the output
// channel, ddy = pad - ky, ddx = pad - kx, rowlen = OW, nrow = OH.  Post-synaptic sum: the TARGET list of the output channel, vals = x_src of the
// input channel, ddy = ky - pad, ddx = kx - pad, rowlen = Wd, nrow = H.  Eight events at a time: eight loads in flight, the additions in order.
__host__ __device__ inline float conv_pp_list_sum(const uint16_t *ev, int n, const float *vals, int ddy, int ddx, int rowlen, int nrow) {
    float acc = 0.f;
    for (int i0 = 0; i0 < n; i0 += 8) {
        int ad[8]; bool in[8]; float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = (int)ev[i0 + u];
            const int y = (q >> 8) + ddy, x = (q & 0xFF) + ddx;
            in[u] = (unsigned)y < (unsigned)nrow && (unsigned)x < (unsigned)rowlen;
            ad[u] = in[u] ? y * rowlen + x : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = vals[ad[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (in[u]) acc += vv[u];
    }
    return acc;
}

}  // namespace snn
