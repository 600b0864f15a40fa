// Host-side check of the kernels' ordered-sum accumulators (bindsnet_amd/csrc/snn_order.hpp) and of the device generator's
// arithmetic (csrc/snn_rng.hpp: mt19937 tempering / twist mix, the glibc log1p port, the Exp(1) draw): the structs and functions
// are __host__ __device__, so they are run HERE on the CPU exactly as the kernels' threads drive them -- only the NON-ZERO terms,
// in ascending index -- and compared with the oracle / torch by tests/test_order_rng_host.py.  Compiled by hipcc like the
// kernels (same front end, -ffp-contract=off); no device code is executed.  Test infrastructure only; not part of libsnnhip.
#include <stdint.h>
#include "../../bindsnet_amd/csrc/snn_order.hpp"
#include "../../bindsnet_amd/csrc/snn_rng.hpp"

using namespace snn;

// k_prop<OuterSum>'s thread (sample b, column j): out[b,j] = sum_i W[i,j] * s[b,i], silent sources skipped.
// kind: 0 OuterSum (class by column), 1 CascadeFlat (Nin < 4096), 2 SeqSum, 3 CascadeN, 4 RowSum4 for every column
extern "C" void hostcheck_prop(const float *W, const uint8_t *s, int B, int Nin, int N, int kind, float *out) {
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < N; ++j) {
            const uint8_t *srow = s + (size_t)b * Nin;
            OuterSum o; CascadeFlat cf; SeqSum q; CascadeN cn; RowSum4 r4;
            o.init(j >= (N / 32) * 32); cf.init(); q.init(false); cn.init(); r4.init();
            for (int i = 0; i < Nin; ++i) {
                if (!srow[i]) continue;
                const float term = W[(size_t)i * N + j] * (float)srow[i];
                switch (kind) {
                    case 0: o.add(i, term, Nin); break;
                    case 1: cf.add(i, term, Nin); break;
                    case 2: q.add(i, term, Nin); break;
                    case 3: cn.add(i, term, Nin); break;
                    default: r4.add(i, term, Nin); break;
                }
            }
            float r;
            switch (kind) {
                case 0: r = o.finish(Nin); break;
                case 1: r = cf.finish(Nin); break;
                case 2: r = q.finish(Nin); break;
                case 3: r = cn.finish(Nin); break;
                default: r = r4.finish(Nin); break;
            }
            out[(size_t)b * N + j] = 0.0f + r;
        }
}

// The batch reduction of the plasticity kernels: element e of E, sum over b of terms[b, e] (zero terms skipped) through OuterSum
// with the flat-index class (e >= 32 * floor(E / 32): row_sum).
extern "C" void hostcheck_batch_sum(const float *terms, int B, long E, float *out) {
    for (long e = 0; e < E; ++e) {
        OuterSum acc;
        acc.init(e >= (E / 32) * 32);
        for (int b = 0; b < B; ++b) {
            const float t = terms[(size_t)b * E + e];
            if (t != 0.0f) acc.add(b, t, B);
        }
        out[e] = acc.finish(B);
    }
}

// n Exp(1) draws of at::mt19937 + exponential_(1) from a 624-word state image and the index of its next output (624 = twist
// first), as k_rng_fill / the resident kernels produce them; the state is advanced in place, *pos updated.
extern "C" void hostcheck_exponential(uint32_t *mt, int *pos, long n, float *out) {
    uint32_t nxt[624];
    int p = *pos;
    auto word = [&]() -> uint32_t {
        if (p >= 624) {                                          // the twist of mt_twist_block, serially
            for (int i = 0; i < 227; ++i) nxt[i] = mt[i + 397] ^ mt_mix(mt[i], mt[i + 1]);
            for (int i = 227; i < 454; ++i) nxt[i] = nxt[i - 227] ^ mt_mix(mt[i], mt[i + 1]);
            for (int i = 454; i < 624; ++i) nxt[i] = nxt[i - 227] ^ mt_mix(mt[i], i == 623 ? nxt[0] : mt[i + 1]);
            for (int i = 0; i < 624; ++i) mt[i] = nxt[i];
            p = 0;
        }
        return mt_temper(mt[p++]);
    };
    for (long k = 0; k < n; ++k) {
        const uint32_t hi = word(), lo = word();
        out[k] = exp1_from_words(hi, lo);
    }
    *pos = p;
}

extern "C" double hostcheck_log1p(double x) { return log1p_glibc(x); }

// ---- csrc/snn_common.hpp: the elementwise neuron updates, driven like k_lif / k_dc_membrane drive them -------------------
#include "../../bindsnet_amd/csrc/snn_common.hpp"

// T steps of LIFNodes.forward over [B*N] neurons: k_lif's loop body (the current is zeroed where the refractory counter is
// positive BEFORE lif_update, nodes.py:511), trace, raster.
extern "C" void hostcheck_lif_sequence(float *v, float *rc, uint8_t *s, float *x, const float *I, long n, int T, const snn_lif_params *p,
                                       uint8_t *raster) {
    for (int t = 0; t < T; ++t)
        for (long k = 0; k < n; ++k) {
            float vv = v[k], r = rc[k], cur = I[(size_t)t * n + k];
            if (r > 0.f) cur = 0.f;
            const uint8_t sp = lif_update(vv, r, cur, *p);
            v[k] = vv; rc[k] = r; s[k] = sp;
            if (p->traces) x[k] = trace_next(x[k], sp, p->trace_decay, p->trace_scale, p->traces_additive);
            raster[(size_t)t * n + k] = sp;
        }
}

// T steps of DiehlAndCookNodes.forward WITHOUT one_spike: k_dc_membrane's loop (theta decay, threshold, dc_update per sample,
// crossing count, theta bump) + the trace of k_dc_arbitrate.
extern "C" void hostcheck_dc_sequence(float *v, float *rc, uint8_t *s, float *x, float *theta, const float *I, int B, int N, int T,
                                      const snn_dc_params *p, uint8_t *raster) {
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < N; ++j) {
            float th = theta[j];
            if (p->learning) th = th * p->theta_decay;
            const float thr = p->lif.thresh + th;
            int cnt = 0;
            for (int b = 0; b < B; ++b) {
                const size_t k = (size_t)b * N + j;
                float vv = v[k], r = rc[k];
                const uint8_t sp = dc_update(vv, r, I[(size_t)t * B * N + k], thr, p->lif);
                v[k] = vv; rc[k] = r; s[k] = sp;
                cnt += sp;
                if (p->lif.traces) x[k] = trace_next(x[k], sp, p->lif.trace_decay, p->lif.trace_scale, p->lif.traces_additive);
                raster[(size_t)t * B * N + k] = sp;
            }
            if (p->learning) th = th + p->theta_plus * (float)cnt;
            theta[j] = th;
        }
}

// ---- csrc/snn_order.hpp inner_sum8 + k_normalize_filters' thread body (Conv2dConnection.normalize) --------------------------------
extern "C" float hostcheck_inner_sum8(const float *x, int n) { return inner_sum8(x, n); }

extern "C" void hostcheck_normalize_filters(float *W, int F, int K, float norm) {
    for (int f = 0; f < F; ++f) {                 // = one thread of k_normalize_filters
        float *w = W + (size_t)f * K;
        const float sum = inner_sum8(w, K);
        const float rc = 1.0f / sum;
        const float scale = rc * norm;
        for (int k = 0; k < K; ++k) w[k] = w[k] * scale;
    }
}
